#!/bin/bash
export TMPDIR=/tmp
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
for v in ATT2 ATT3; do cp exp/lib_$v.so $P; touch $P; echo "== $v"; for bp in "4 4096" "2 4096" "32 50" "8 1024"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 2>&1 | tail -n 1; done; done
cp exp/lib_ATT2.so $P; touch $P
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q ) 2>&1 | tail -n 3
DIB_ST_ATTENTION=flash timeout 300 python tools/set_transformer_bench.py --batch 4 --particles 4096 --steps 5 2>&1 | tail -n 1
