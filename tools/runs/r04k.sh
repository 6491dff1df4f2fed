#!/bin/bash
# round-4 GPU call K: lse kernel with batched loads, int32 gather index, Philox bounds: InfoNCE + trajectory + full-size tests,
# InfoNCE / loop times
export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_trajectories.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -k "infonce or trajectory or dense_stack or fullsize or config3 or config4 or pendulum" --durations=4 ) > $O/tests.log 2>&1; tail -n 8 $O/tests.log
( timeout 200 python tools/infonce_bench.py --dims 64 ) 2>&1 | grep -v "l1\|linf" | tee $O/infonce_bench.txt
python tools/config2_loop_trace.py 2048 2>/dev/null | tail -n 1 | tee $O/config2_loop.txt; python tools/config2_loop_trace.py 128 2>/dev/null | tail -n 1 | tee -a $O/config2_loop.txt
python tools/small_batch_bench.py 2>/dev/null | tail -n 1 | tee $O/small_batch.txt
