#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_set_transformer.py ) > $O/suite.log 2>&1
tail -n 4 $O/suite.log
for i in 1 2; do
echo "== B=65536 fork";  TAG=f bash tools/ab_bench.sh FORK
echo "== B=65536 no fork";  DIB_CONCURRENT_WGRAD=0 TAG=nf bash tools/ab_bench.sh FORK
done
echo "== B=8192 fork";  BATCH=8192 TAG=b8192f bash tools/ab_bench.sh FORK
echo "== B=8192 no fork";  DIB_CONCURRENT_WGRAD=0 BATCH=8192 TAG=b8192nf bash tools/ab_bench.sh FORK
echo "== B=16384 fork";  BATCH=16384 TAG=b16f bash tools/ab_bench.sh FORK
echo "== B=16384 no fork";  DIB_CONCURRENT_WGRAD=0 BATCH=16384 TAG=b16nf bash tools/ab_bench.sh FORK
