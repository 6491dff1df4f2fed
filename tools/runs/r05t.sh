#!/bin/bash
# r05t: 8-wave attention forward and backward switched separately: same-box A/B of the notebook-size set-transformer step
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05t; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_set_transformer.py -q -x -m gpu -k "8_waves" ) > $O/tests.txt 2>&1; tail -n 2 $O/tests.txt
for rep in 1 2 3; do for t in "attn_small_fwd_waves=4,attn_small_bwd_waves=4" "attn_small_fwd_waves=4,attn_small_bwd_waves=8" "attn_small_fwd_waves=8,attn_small_bwd_waves=8" "attn_small_fwd_waves=8,attn_small_bwd_waves=4"; do
  timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 80 --warmup 8 --tuning $t 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['tuning'], d['ms_per_step'])"
done; done | tee $O/st_ab.txt
