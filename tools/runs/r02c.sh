#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp exp/lib_NEW2.so $P; touch $P
( time timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q --durations=10 ) > $O/st.log 2>&1
tail -n 40 $O/st.log
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s --durations=10 ) > $O/fullsize.log 2>&1
tail -n 6 $O/fullsize.log
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_set_transformer.py ) > $O/suite.log 2>&1
tail -n 4 $O/suite.log
echo "== A/B B=65536"; bash tools/ab_bench.sh BASE NEW2 SPLIT1024
echo "== A/B B=8192";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE NEW2 SPLIT1024
echo "== A/B B=16384";  BATCH=16384 TAG=b16384 bash tools/ab_bench.sh BASE NEW2
echo "== A/B B=32768";  BATCH=32768 TAG=b32768 bash tools/ab_bench.sh BASE NEW2
cp exp/lib_NEW2.so $P; touch $P
