#!/bin/bash
# r06j: config 3 under fewer weight-gradient slabs (dib_set_tuning "wgrad_max_splits": the tail's slab reduce reads slabs x 8.9 MB),
# same box, interleaved; then config 4 and B = 8192 for the best candidates
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06j; mkdir -p $O
cd $R
for rep in 1 2; do
  for ms in 32 16 8 24; do
    timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra --tuning wgrad_max_splits=$ms 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 max_splits $ms', d['ms_per_step'], d['timing']['blocks_ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline_by_kernel'].items() if k.startswith('dib_gemm_kernel<2')})"
  done
done | tee $O/config3_max_splits_ab.txt
for ms in 32 16 8; do
  timeout 300 python bench.py --features 50 --steps 20 --warmup 4 --no-cpu-baseline --no-extra --tuning wgrad_max_splits=$ms 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 max_splits $ms', d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
  timeout 300 python bench.py --batch 8192 --steps 50 --warmup 5 --no-cpu-baseline --no-extra --tuning wgrad_max_splits=$ms 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8192 max_splits $ms', d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
done | tee -a $O/config3_max_splits_ab.txt
