#!/bin/bash
# r03a: GPU parity suite on the first round-3 library (u - mu backward) + same-box A/B of the fused-kernel diet variants
# (exp/lib_V00/V10/V01/V11.so from tools/build_variant.sh with -DDIB_H1_MASK / -DDIB_DW1_B64; the knobs were removed afterwards)
mkdir -p gpurun_out/r03a
(timeout 600 python -m pytest tests -m gpu -x -q -k "not set_transformer" > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.log)
REPS=2 STEPS=20 bash tools/ab_bench.sh V00 V10 V01 V11 2>&1 | tee gpurun_out/r03a/ab.log
BATCH=8192 TAG=b8192 bash tools/ab_bench.sh V00 V11 2>&1 | tee -a gpurun_out/r03a/ab.log
