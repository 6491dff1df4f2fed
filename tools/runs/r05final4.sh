#!/bin/bash
# r05final4: kernel trace of the reference's default run (B = 128, fit with the epoch's validation batches merged) at HEAD
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05final4; mkdir -p $O
cd /tmp
DIB_SMALL_EPOCHS=50 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/small_batch_bench.py > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05final4/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
pairs=(3+50)*8
print("kernel us per (train + validation) step pair", round(tot/1e3/pairs,1))
for r in rows[:12]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/pairs,3), round(float(r["AverageNs"])/1e3,2))
PY
tail -n 2 $O/kt.log
