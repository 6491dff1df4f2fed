#!/bin/bash
# ping-pong forward: finer phase timers
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ad
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
cp exp/lib_FTIMING.so $P; touch $P
timeout 200 python tools/fused_phase_timing.py --steps 20 2>&1 | tail -n 32 | head -16 | tee gpurun_out/r02ad/phases.txt
cp /tmp/keep.so $P
