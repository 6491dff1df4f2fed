#!/bin/bash
# r05n: same-box A/B of the row-tile variants: HEAD (generalised contraction shares + column-owned tile loads) vs K0 (2 shares),
# TS (flat tile loop), K0TS (both = r05l's library)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05n; mkdir -p $O
for rep in 1 2; do
for v in HEAD K0 TS K0TS; do
  if [ $v = HEAD ]; then unset DIB_LIB_PATH; else export DIB_LIB_PATH=$R/exp/lib_$v.so; fi
  st=$(timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  bc=$(timeout 120 python tools/small_batch_bench.py 2>&1 | tail -n 1 | sed 's/.*-> \([0-9]*\) us.*/\1/')
  c2=$(timeout 200 python tools/config2_loop_trace.py 128 2>&1 | tail -n 1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$v rep $rep: set transformer 32x50 $st ms/step | boolean default $bc us/pair | infonce loop B=128 $c2 ms/step" | tee -a $O/ab.txt
done
done
unset DIB_LIB_PATH
