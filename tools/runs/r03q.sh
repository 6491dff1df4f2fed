#!/bin/bash
# r03q: weight-gradient GEMMs with the second operand's prefetch issued in the middle of the MFMA phase (exp/lib_GSPLIT.so)
# vs the product (exp/lib_GBASE.so): parity of the GEMM modes + same-box A/B at B = 65536 and 8192
O=gpurun_out/r03q; mkdir -p $O
(DIB_LIB_PATH=exp/lib_GSPLIT.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or forward_backward_parity or split_batch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
REPS=3 STEPS=20 bash tools/ab_bench.sh GBASE GSPLIT 2>&1 | tee $O/ab.log
BATCH=8192 TAG=b8192 REPS=2 bash tools/ab_bench.sh GBASE GSPLIT 2>&1 | tee -a $O/ab.log
