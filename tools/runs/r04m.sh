#!/bin/bash
# round-4 GPU call M: 16-row chunks of the fused output head / skinny wgrad for small batches: parity file + default-batch time
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp_and_cache.py tests/test_gpu_trajectories.py::test_keras_path_trajectory_160_steps_through_the_ramp -q --durations=3 ) > $O/tests.log 2>&1; tail -n 6 $O/tests.log
for i in 1 2; do python tools/small_batch_bench.py 2>/dev/null | tail -n 1; done | tee $O/small_batch.txt
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b8192', d['ms_per_step'])"
