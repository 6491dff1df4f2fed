#!/bin/bash
# round 6q: cluster mode at its shipped defaults - tests on the cluster_tiles dispatch path, phase timing, default-pair A/B
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "integration_cluster_equals or stale_gradient_slabs or test_forward_backward_parity or random_architectures or late_annealing or fit_trajectory or tuning_switchboard or small_batch_row_tile or workspace_needs" 2>&1 | tail -n 5 > $O/tests_a.txt
cat $O/tests_a.txt
for c in 0 4; do
  echo "== int_cluster=$c"; DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=$c 2>&1 | grep -v "amdgpu.ids\|encoder\|gaps"
done > $O/phase_timing.txt
cat $O/phase_timing.txt
for rep in 1 2 3; do for c in 0 4; do
  DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py int_cluster=$c 2>&1 | tail -n 1
done; done > $O/default_pair_ab.txt
cat $O/default_pair_ab.txt
timeout 400 python tools/int_cluster_sweep.py 2>&1 | grep -v amdgpu.ids > $O/int_cluster_sweep.txt
DIB_SWEEP_F=4 timeout 400 python tools/int_cluster_sweep.py 2>&1 | grep -v amdgpu.ids >> $O/int_cluster_sweep.txt
cat $O/int_cluster_sweep.txt
