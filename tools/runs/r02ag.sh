#!/bin/bash
# fused forward: dynamic sub-tile tickets (LDS counter) vs the static wave -> rows map, A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ag
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh STATIC DYN STATIC DYN 2>&1 | tee gpurun_out/r02ag/ab.txt
BATCH=8192 TAG=b8192 bash tools/ab_bench.sh STATIC DYN 2>&1 | tee -a gpurun_out/r02ag/ab.txt
cp exp/lib_FTIMING.so $P; touch $P
timeout 200 python tools/fused_phase_timing.py --steps 20 2>&1 | tail -n 18 | tee gpurun_out/r02ag/phases.txt
cp exp/lib_DYN.so $P; touch $P
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "config3_full_batch or f50" 2>&1 | tail -n 3
cp /tmp/keep.so $P
