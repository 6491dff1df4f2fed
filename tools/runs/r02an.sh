#!/bin/bash
# fused forward: h2 stash + mask bits as fillers between the layer-3 MFMA steps (DIB_FUSED_DENSE), A/B + parity on the new build
export TMPDIR=/tmp
mkdir -p gpurun_out/r02an
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
STEPS=10 bash tools/ab_bench.sh BASE DENSE BASE DENSE 2>&1 | tee gpurun_out/r02an/ab.txt
cp exp/lib_DENSE.so $P; touch $P
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 2 | tee gpurun_out/r02an/tests.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "config3_full_batch" 2>&1 | tail -n 2 | tee -a gpurun_out/r02an/tests.txt
cp /tmp/keep.so $P
