#!/bin/bash
# (1) write-drain microbenchmark (2) fused kernels with plain instead of non-temporal stash stores
export TMPDIR=/tmp
mkdir -p gpurun_out/r02z2
timeout 120 exp/write_drain 2>&1 | tee gpurun_out/r02z2/write_drain.txt
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh BASE PLAIN 2>&1 | tee gpurun_out/r02z2/ab.txt
cp /tmp/keep.so $P
