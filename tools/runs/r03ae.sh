#!/bin/bash
# r03ae: from how few tokens the streaming skinny-K projections pay: the notebook's own batch (32 x 50 = 1600 tokens) and
# smaller, same box; the set-transformer parity file with EVERY eligible projection on the streaming kernel (threshold 1)
O=gpurun_out/r03ae; mkdir -p $O
(DIB_SKINNY_K_MIN_TOKENS=1 timeout 1500 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "not config5_size" > $O/pytest_st_all_skinny.log 2>&1; echo "rc=$?" >> $O/pytest_st_all_skinny.log); tail -3 $O/pytest_st_all_skinny.log
for rep in 1 2; do for v in 2048 1; do for a in "--batch 32 --particles 50 --steps 50 --warmup 5" "--batch 8 --particles 50 --steps 50 --warmup 5" "--batch 4 --particles 256 --steps 30 --warmup 5"; do echo "min_tokens=$v $a $(DIB_SKINNY_K_MIN_TOKENS=$v python tools/set_transformer_bench.py $a 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; done; done; done | tee $O/st_ab.txt
