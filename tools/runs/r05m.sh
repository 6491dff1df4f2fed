#!/bin/bash
# r05m: contraction shares generalised (8 waves = slots x shares), batched tile loads; poisoned-workspace test
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05m; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_parity.py -q -x -m gpu ) > $O/tests.txt 2>&1
tail -n 8 $O/tests.txt
timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 2>/dev/null | tail -n 1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05m/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernel ms per step", round(tot/1e6/35,3), "launches per step", round(calls/35,1))
for r in rows[:16]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/35,1), round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
timeout 120 python tools/small_batch_bench.py 2>&1 | tail -n 1
timeout 200 python tools/config2_loop_trace.py 128 2>&1 | tail -n 1
