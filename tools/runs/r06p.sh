#!/bin/bash
# round 6p: cluster mode of the row-tile integration kernel - parity subset, phase timing, default-pair A/B
O=gpurun_out/r06p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "integration_cluster_equals or workspace_needs_only or stale_gradient_slabs" 2>&1 | tail -n 8 > $O/tests_a.txt
cat $O/tests_a.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_forward_backward_parity or small_batch_row_tile or step_tail_equals" 2>&1 | tail -n 8 > $O/tests_b.txt
cat $O/tests_b.txt
for c in 0 2 4 8; do
  echo "== int_cluster=$c"; DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=$c 2>&1 | grep -v amdgpu.ids
done > $O/phase_timing.txt
cat $O/phase_timing.txt
for rep in 1 2; do for c in 0 2 4 8; do
  DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py int_cluster=$c 2>&1 | tail -n 1
done; done > $O/default_pair_ab.txt
cat $O/default_pair_ab.txt
