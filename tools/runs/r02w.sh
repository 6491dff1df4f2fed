#!/bin/bash
# timeline of the fused forward: per-workgroup entry / staged / done and the gaps to the neighbouring kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/r02w
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
cp exp/lib_FTIMING.so $P; touch $P
for n in 30; do echo "== after $n steps"; timeout 200 python tools/fused_phase_timing.py --steps $n 2>&1 | tail -n 21; done | tee gpurun_out/r02w/phases.txt
cp /tmp/keep.so $P
