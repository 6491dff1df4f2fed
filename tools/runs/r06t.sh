#!/bin/bash
# round 6t: cluster mode incl. the paired grid - parity tests of the row-tile paths, InfoNCE loop A/B, default pair
O=gpurun_out/r06t; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_concurrency.py -q -m gpu 2>&1 | tail -n 6 > $O/tests.txt
cat $O/tests.txt
for rep in 1 2; do for c in 0 8; do
  DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py int_cluster=$c 2>&1 | tail -n 1
done; done > $O/default_pair_ab.txt
cat $O/default_pair_ab.txt
timeout 600 python tools/config2_cluster_ab.py 128 2048 2>&1 | grep -v amdgpu.ids > $O/config2_loop_ab.txt
cat $O/config2_loop_ab.txt
