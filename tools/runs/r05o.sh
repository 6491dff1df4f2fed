#!/bin/bash
# r05o: same-box A/B of the set-transformer token-chain kernels against the layer-by-layer launches (32 x 50 and 8 x 200 and
# 2 x 2048 particles), whole GPU suite at HEAD
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05o; mkdir -p $O
for rep in 1 2; do
for shape in "32 50" "8 200" "2 2048"; do
  set -- $shape
  for c in 1 0; do
    ms=$(timeout 100 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 60 --warmup 8 --chain $c 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "rep $rep: $1 x $2 particles, chain=$c: $ms ms/step" | tee -a $O/chain_ab.txt
  done
done
done
( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/tests.txt 2>&1
tail -n 8 $O/tests.txt
