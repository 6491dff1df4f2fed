#!/bin/bash
# round-4 GPU call I: kernel trace of the config-2 InfoNCE loop step at B = 2048 and B = 128 (where does the step go?)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04i; mkdir -p $O
cd /tmp
for b in 2048 128; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$b -o kt -- python $R/tools/config2_loop_trace.py $b > $O/kt_$b.log 2>&1
  find $O/kt_$b -mindepth 2 -type f -exec mv {} $O/kt_$b/ \; 2>/dev/null
done
cd $R
python - <<'PY'
import csv,glob
for b in (2048,128):
    f=glob.glob(f"gpurun_out/r04i/kt_{b}/kt_kernel_stats.csv")
    if not f: print(b,"no stats"); continue
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    steps=16+4+64+8   # warm-up fit (16 train + 4 val) + timed fit (64 train + 8 val)
    print(b, "kernel us per step (all steps incl. warm-up)", round(tot/1e3/steps,1))
    for r in rows[:26]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
