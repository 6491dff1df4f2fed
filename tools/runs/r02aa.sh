#!/bin/bash
# fused kernels: alternating s_setprio between the two waves of a SIMD (A/B on one box) + per-wave timeline
export TMPDIR=/tmp
mkdir -p gpurun_out/r02aa
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh BASE PRIO1 PRIO2 BASE PRIO1 2>&1 | tee gpurun_out/r02aa/ab.txt
BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE PRIO1 2>&1 | tee -a gpurun_out/r02aa/ab.txt
cp exp/lib_FTIMING.so $P; touch $P
timeout 200 python tools/fused_phase_timing.py --steps 30 2>&1 | tail -n 14 | tee gpurun_out/r02aa/phases.txt
cp /tmp/keep.so $P
