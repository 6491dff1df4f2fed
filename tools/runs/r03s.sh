#!/bin/bash
# r03s: the adopted prefetch-piece rule (exp/lib_G5.so = product) vs the round-2 prefetch (exp/lib_GBASE.so); attention forward
# with the V tile's prefetch after the S product (exp/lib_G5A.so); GEMM-heavy parity subset on the product
O=gpurun_out/r03s; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_set_transformer.py -m gpu -q -x -k "not config5_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
REPS=3 STEPS=20 bash tools/ab_bench.sh GBASE G5 2>&1 | tee $O/ab.log
BATCH=8192 TAG=b8192 REPS=2 bash tools/ab_bench.sh GBASE G5 2>&1 | tee -a $O/ab.log
for rep in 1 2; do for v in G5 G5A; do echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash 1 2>/dev/null)"; done; done | tee $O/attn_ab.txt
for v in GBASE G5; do echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python bench.py --config5-only --steps 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; done | tee $O/st_ab.txt
