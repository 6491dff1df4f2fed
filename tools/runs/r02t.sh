#!/bin/bash
# fused encoder forward: paired output tiles (alternating accumulators) + pinned weight-fragment prefetch, A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r02t
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh BASE P1 BASE P1 2>&1 | tee gpurun_out/r02t/ab.txt
BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE P1 2>&1 | tee -a gpurun_out/r02t/ab.txt
cp exp/lib_P1.so $P; touch $P
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 3
cp /tmp/keep.so $P
