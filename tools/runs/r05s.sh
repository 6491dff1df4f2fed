#!/bin/bash
# r05s: 8-wave single-workgroup attention (forward + backward) for <= 64 particles: bit-identity with the 4-wave kernels, the
# attention parity tests under 8 waves, same-box A/B of the notebook-size set-transformer step, kernel trace
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05s; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q -x -m gpu -k "8_waves or score_stash_equals or tuning" ) > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
for rep in 1 2; do for w in 4 8; do
  timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 60 --warmup 8 --tuning attn_small_waves=$w 2>/dev/null | tail -n 1
done; done | tee $O/st_ab.txt
timeout 100 python tools/set_transformer_bench.py --batch 8 --particles 64 --steps 40 --warmup 8 --tuning attn_small_waves=4 2>/dev/null | tail -n 1 | tee -a $O/st_ab.txt
timeout 100 python tools/set_transformer_bench.py --batch 8 --particles 64 --steps 40 --warmup 8 --tuning attn_small_waves=8 2>/dev/null | tail -n 1 | tee -a $O/st_ab.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 --tuning attn_small_waves=8 > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05s/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:14]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/35,1), round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
