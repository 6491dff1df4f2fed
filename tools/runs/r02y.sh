#!/bin/bash
# timing experiment: fused forward with individual output streams removed (bit 0 h1, 1 h2, 2 u, 3 mu|logvar; results wrong - timing only)
export TMPDIR=/tmp
mkdir -p gpurun_out/r02y
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for sk in 0 4 8 12 15; do DIB_SKIP_STASH=$sk TAG=skip$sk STEPS=10 bash tools/ab_bench.sh SKIP; done 2>&1 | tee gpurun_out/r02y/ab2.txt
cp /tmp/keep.so $P
