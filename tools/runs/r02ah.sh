#!/bin/bash
# (1) attention backward: exponentials of query group g+1 under the dV MFMAs of group g, K fragments across the barrier
# (2) fused encoder backward: paired dh2 tiles + pinned W fragment prefetch.  OLD = both off, NEW = both on.
export TMPDIR=/tmp
O=gpurun_out/r02ah; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp $P /tmp/keep.so
for v in OLD NEW OLD NEW; do
  cp exp/lib_$v.so $P; touch $P
  for bp in "4 4096" "8 1024"; do set -- $bp; echo -n "$v "; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done
done | tee $O/attn_ab.txt
bash tools/ab_bench.sh OLD NEW OLD NEW 2>&1 | tee $O/ab.txt
cp exp/lib_NEW.so $P; touch $P
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_parity.py -q -x ) 2>&1 | tail -n 3
cp /tmp/keep.so $P
