#!/bin/bash
# fused forward: real ping-pong (provably uniform barrier branches) and real alternating s_setprio, A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ae
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh BASE PP PRIO1 PRIO3 BASE PP 2>&1 | tee gpurun_out/r02ae/ab.txt
cp exp/lib_FTIMING.so $P; touch $P
timeout 200 python tools/fused_phase_timing.py --steps 20 2>&1 | tail -n 32 | tee gpurun_out/r02ae/phases.txt
cp exp/lib_PP.so $P; touch $P
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 3
cp /tmp/keep.so $P
