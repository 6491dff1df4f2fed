#!/bin/bash
# round 6x: the encoder row-tile kernels' layers 2 / 3 and both dgrads on the slice primitives
O=gpurun_out/r06x; mkdir -p $O
for c in 8 0; do echo "== int_cluster=$c"; DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=$c 2>&1 | grep -v "amdgpu.ids"; done > $O/phase_timing.txt
cat $O/phase_timing.txt
for rep in 1 2 3; do DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py 2>&1 | tail -n 1; done > $O/default_pair.txt
cat $O/default_pair.txt
timeout 600 python tools/config2_cluster_ab.py 128 2048 2>&1 | grep -v amdgpu.ids > $O/config2.txt; cat $O/config2.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_concurrency.py tests/test_gpu_dp_and_cache.py -q -m gpu 2>&1 | tail -n 6 > $O/tests.txt; cat $O/tests.txt
