#!/bin/bash
# r03r: prefetch-piece variants of the GEMM (-DDIB_GEMM_SPLIT_PREFETCH=1..4: 2 / 4 pieces, weight gradients only / every mode)
O=gpurun_out/r03r; mkdir -p $O
REPS=2 STEPS=20 bash tools/ab_bench.sh GBASE GS1 GS2 GS3 GS4 2>&1 | tee $O/ab.log
BATCH=8192 TAG=b8192 REPS=2 bash tools/ab_bench.sh GBASE GS1 GS2 GS3 GS4 2>&1 | tee -a $O/ab.log
for v in GS2 GS4; do (DIB_LIB_PATH=exp/lib_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or forward_backward_parity or split_batch" > $O/pytest_$v.log 2>&1; echo "rc=$?" >> $O/pytest_$v.log); tail -2 $O/pytest_$v.log; done
