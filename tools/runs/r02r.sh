#!/bin/bash
# flash attention A/B: BASE (committed) vs V1 (new backward, dt-inner PV) vs V2 (+ split S chain in forward) vs V3 (V2 at 2 waves/SIMD)
export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for v in BASE V1 V2 V3; do
  cp exp/lib_$v.so $P; touch $P
  for bp in "4 4096" "8 1024"; do set -- $bp; echo -n "$v "; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done
done | tee $O/attn_ab.txt
cp exp/lib_V1.so $P; touch $P
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q -x ) > $O/st_tests.log 2>&1; tail -n 3 $O/st_tests.log
timeout 200 python tools/attn_phase_timing.py 2>&1 | tee $O/phase_timing.txt
timeout 200 python tools/attn_phase_timing.py --batch 8 --particles 1024 2>&1 | tee -a $O/phase_timing.txt
cp /tmp/keep.so $P
