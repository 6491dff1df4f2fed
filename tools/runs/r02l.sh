#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_set_transformer.py ) > $O/suite.log 2>&1
tail -n 6 $O/suite.log
echo "== B=8192 fused head";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh HEAD
echo "== B=8192 unfused";  DIB_DISABLE_FUSED_HEAD=1 BATCH=8192 TAG=b8192u bash tools/ab_bench.sh HEAD
echo "== B=8192 fused head";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh HEAD
echo "== B=65536 fused";  bash tools/ab_bench.sh HEAD
echo "== B=65536 unfused";  DIB_DISABLE_FUSED_HEAD=1 TAG=u bash tools/ab_bench.sh HEAD
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k "not trajectory" ) > $O/fullsize.log 2>&1
tail -n 4 $O/fullsize.log
