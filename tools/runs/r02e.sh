#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q --durations=5 ) > $O/st.log 2>&1
tail -n 30 $O/st.log
for at in gemm flash; do for bp in "32 50" "4 512" "2 2048" "1 4096" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done
echo "== B=8192";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh NEW4
echo "== B=65536";  bash tools/ab_bench.sh NEW4
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_set_transformer.py ) > $O/suite.log 2>&1
tail -n 4 $O/suite.log
