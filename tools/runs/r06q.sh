#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "integration_cluster_equals or stale_gradient_slabs" 2>&1 | tail -n 5 > $O/tests_a.txt
cat $O/tests_a.txt
for c in 2 4 8; do
  echo "== int_cluster=$c"; DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=$c 2>&1 | grep -v "amdgpu.ids\|encoder\|gaps"
done > $O/phase_timing.txt
cat $O/phase_timing.txt
for rep in 1 2; do for c in 0 4 8; do
  DIB_SMALL_EPOCHS=1000 timeout 300 python tools/small_batch_bench.py int_cluster=$c 2>&1 | tail -n 1
done; done > $O/default_pair_ab.txt
cat $O/default_pair_ab.txt
