#!/bin/bash
# round 6u: the cluster size adapts to the row-tile count (8 / 4 / 2 workgroups per tile within 256 workgroups)
O=gpurun_out/r06u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "integration_cluster_equals or companion or small_batch_row_tile or dense_stack" 2>&1 | tail -n 4 > $O/tests.txt
cat $O/tests.txt
timeout 500 python tools/int_cluster_sweep.py 2>&1 | grep -v amdgpu.ids > $O/int_cluster_sweep.txt
DIB_SWEEP_F=4 timeout 500 python tools/int_cluster_sweep.py 2>&1 | grep -v amdgpu.ids >> $O/int_cluster_sweep.txt
cat $O/int_cluster_sweep.txt
timeout 600 python tools/config2_cluster_ab.py 1024 2048 2>&1 | grep -v amdgpu.ids > $O/config2_loop_ab.txt
cat $O/config2_loop_ab.txt
