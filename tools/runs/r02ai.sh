#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ai; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
echo "== product library (committed state)"
timeout 300 python -m pytest tests/test_gpu_set_transformer.py -q -x -k large_token 2>&1 | grep -E "^E|passed|failed" | head -12
cp exp/lib_NEW.so $P; touch $P
echo "== NEW"
timeout 300 python -m pytest tests/test_gpu_set_transformer.py -q -x -k large_token 2>&1 | grep -E "^E|passed|failed" | head -12
cp /tmp/keep.so $P
