#!/bin/bash
# flash attention A/B: V1 (previous) vs V4 (by-value tile staging: no scratch in the backward) vs V5 (V4 with the forward's array tiles)
export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for v in V1 V4 V5; do
  cp exp/lib_$v.so $P; touch $P
  for bp in "4 4096" "8 1024"; do set -- $bp; echo -n "$v "; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done
done | tee $O/attn_ab.txt
cp exp/lib_V4.so $P; touch $P
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q -x ) > $O/st_tests.log 2>&1; tail -n 3 $O/st_tests.log
timeout 200 python tools/attn_phase_timing.py 2>&1 | tee $O/phase_timing.txt
cp /tmp/keep.so $P
