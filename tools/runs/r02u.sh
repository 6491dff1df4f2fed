#!/bin/bash
# where the fused forward's wave cycles go (s_memtime per phase), plain and paired-tile builds
export TMPDIR=/tmp
mkdir -p gpurun_out/r02u
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for v in FTIMING FTIMINGP; do cp exp/lib_$v.so $P; touch $P; echo "== $v"; timeout 200 python tools/fused_phase_timing.py 2>&1 | tail -n 11; done | tee gpurun_out/r02u/phases.txt
cp /tmp/keep.so $P
