#!/bin/bash
# shader clock under load: s_memtime vs s_memrealtime inside the fused forward and the attention backward
export TMPDIR=/tmp
mkdir -p gpurun_out/r02v
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
cp exp/lib_FTIMING.so $P; touch $P; timeout 200 python tools/fused_phase_timing.py 2>&1 | tail -n 12 | tee gpurun_out/r02v/phases.txt
timeout 200 python tools/fused_phase_timing.py --batch 8192 2>&1 | tail -n 12 | tee -a gpurun_out/r02v/phases.txt
cp /tmp/keep.so $P
timeout 200 python tools/attn_phase_timing.py 2>&1 | tail -n 10 | tee -a gpurun_out/r02v/phases.txt
