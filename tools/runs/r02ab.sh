#!/bin/bash
# fused forward as a ping-pong of the two waves of a SIMD (matrix segment / other segment, 2 s_barrier per tile)
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ab
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
bash tools/ab_bench.sh BASE PP PPNP BASE PP 2>&1 | tee gpurun_out/r02ab/ab.txt
BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE PP 2>&1 | tee -a gpurun_out/r02ab/ab.txt
cp exp/lib_FTIMING.so $P; touch $P
timeout 200 python tools/fused_phase_timing.py --steps 30 2>&1 | tail -n 26 | tee gpurun_out/r02ab/phases.txt
cp exp/lib_PP.so $P; touch $P
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 3
cp /tmp/keep.so $P
