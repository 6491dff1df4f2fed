#!/bin/bash
# round-4 GPU call D: InfoNCE with the parallel lse combine (parity + times), GEMM LDS-fragment prefetch A/B (FP1 vs BASE4),
# empirical sweep of the weight-gradient split count for F = 50 (DIB_WGRAD_NS override)
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "infonce" ) > $O/tests.log 2>&1; tail -n 3 $O/tests.log
( timeout 200 python tools/infonce_bench.py --dims 64 ) > $O/infonce_bench.txt 2>&1; grep -v "l1\|linf" $O/infonce_bench.txt
REPS=2 bash tools/ab_bench.sh BASE4 FP1 2>&1 | tee $O/gemm_frag_prefetch_ab.txt
REPS=1 BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE4 FP1 2>&1 | tee -a $O/gemm_frag_prefetch_ab.txt
run() {  # $1 = DIB_WGRAD_NS
  echo -n "F=50 DIB_WGRAD_NS=$1: "
  DIB_WGRAD_NS=$1 timeout 180 python bench.py --features 50 --steps 10 --warmup 3 --blocks 2 --no-cpu-baseline --no-extra 2> $O/ab.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d.get('roofline_by_kernel',{})
print(d['ms_per_step'], {k[4:].replace('_kernel',''): v['ms_per_step'] for k,v in r.items() if 'gemm<2' in k})"
}
# 50 tiles = encoder layer-2 (128x128 tile) AND layer-3 (128x64 tile) wgrads (both forced by the same key), 26 = integration layer 1
for ns in 32 30 28 25 20 16 10; do run "50:$ns,26:32"; done 2>&1 | tee $O/split_sweep_F50.txt
for ns in 32 29 24 19 16 13; do run "50:32,26:$ns"; done 2>&1 | tee -a $O/split_sweep_F50.txt
