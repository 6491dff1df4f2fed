#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp exp/lib_B64.so $P; touch $P
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $O/parity_b64.log 2>&1; tail -n 3 $O/parity_b64.log
( DIB_GEMM_MODE=bf16x6 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $O/parity_b64_bf16.log 2>&1; tail -n 3 $O/parity_b64_bf16.log
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q -x ) > $O/st_b64.log 2>&1; tail -n 2 $O/st_b64.log
for i in 1 2; do
echo "== B=65536"; bash tools/ab_bench.sh NOB64 B64 BK212
done
echo "== B=8192"; BATCH=8192 TAG=b8192 bash tools/ab_bench.sh NOB64 B64 BK212
