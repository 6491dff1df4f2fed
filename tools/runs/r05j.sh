#!/bin/bash
# r05j: whole GPU suite at HEAD (merged weight gradients, single-bucket small-batch DP, fused reduce + Adam in the set transformer)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05j; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/tests.txt 2>&1
tail -n 12 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 2>/dev/null | tail -n 1
