#!/bin/bash
# flash forward at 2 vs 3 waves per SIMD after the tile-staging fix (no stack object any more)
export TMPDIR=/tmp
mkdir -p gpurun_out/r02al
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for v in W3 W2 W3 W2; do cp exp/lib_$v.so $P; touch $P
  for bp in "4 4096" "8 1024" "32 50"; do set -- $bp; echo -n "$v "; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done
done | tee gpurun_out/r02al/attn_ab.txt
cp exp/lib_W2.so $P; touch $P
( timeout 300 python -m pytest tests/test_gpu_set_transformer.py -q -x ) 2>&1 | tail -n 2
cp /tmp/keep.so $P
