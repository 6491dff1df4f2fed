#!/bin/bash
# fused kernels: stash stores straight from the accumulator fragments (no LDS transpose), with and without ping-pong
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ac
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh PAIR DIRECT PPDIRECT PAIR DIRECT 2>&1 | tee gpurun_out/r02ac/ab.txt
cp exp/lib_DIRECT.so $P; touch $P
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 3
cp /tmp/keep.so $P
