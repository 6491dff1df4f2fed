#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp exp/lib_PE.so $P; touch $P
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py ) > $O/suite.log 2>&1
tail -n 12 $O/suite.log
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "not trajectory" ) > $O/fullsize.log 2>&1
tail -n 6 $O/fullsize.log
echo "== B=8192";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh PE
echo "== B=65536";  bash tools/ab_bench.sh PE
echo "== B=8192 again";  BATCH=8192 TAG=b8192b bash tools/ab_bench.sh PE
