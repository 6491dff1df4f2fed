#!/bin/bash
# round 6v: dib_small_step - encoder forward + integration network + encoder backward as one launch
O=gpurun_out/r06v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp_and_cache.py -q -m gpu -k "one_launch_step_equals or three_bucket_backward_hooks" 2>&1 | tail -n 4 > $O/tests_a.txt
cat $O/tests_a.txt
for c in 0 1; do echo "== small_step_one_launch=$c"; DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py 128 small_step_one_launch=$c 2>&1 | grep -v "amdgpu.ids\|inside"; done > $O/phase_timing.txt
cat $O/phase_timing.txt
for rep in 1 2; do for c in 0 1; do
  DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py small_step_one_launch=$c 2>&1 | tail -n 1
done; done > $O/default_pair_ab.txt
cat $O/default_pair_ab.txt
