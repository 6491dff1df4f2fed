#!/bin/bash
# r03g: cheap A/Bs - cache policy of the score stash (exp/lib_ATTN{0,1,2}.so: -DDIB_ATTN_STASH_NT=0/1/2; ATT0 = 3 = product),
# tile forcing of the set transformer's K = 32 / N = 32 projection GEMMs at config 5 (DIB_FORCE_TILE{0,1,2})
O=gpurun_out/r03g; mkdir -p $O
for rep in 1 2; do for v in ATT0 ATTN0 ATTN1 ATTN2; do echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash 1 2>/dev/null)"; done; done 2>&1 | tee $O/attn_nt_ab.txt
for t in "" "DIB_FORCE_TILE0=12" "DIB_FORCE_TILE0=11" "DIB_FORCE_TILE0=21" "DIB_FORCE_TILE1=11" "DIB_FORCE_TILE1=12" "DIB_FORCE_TILE2=22" "DIB_FORCE_TILE2=11"; do
  echo "[$t] $(env $t python bench.py --config5-only --steps 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline_by_kernel"]["dib_attn_fwd_kernel"]["avg_launch_ms"], d["roofline_by_kernel"]["dib_attn_bwd_kernel"]["avg_launch_ms"])')"
done 2>&1 | tee $O/config5_tile_forcing.txt
