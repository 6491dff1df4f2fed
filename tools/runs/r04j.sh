#!/bin/bash
# round-4 GPU call J: narrow-layout weight-gradient slabs (>= 128 rows when the largest wgrad would stay under one workgroup per
# CU), copy-free InfoNCE loop step, per-workgroup loss partials: whole GPU suite, config-2 loop times + kernel trace, smoke
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04j; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -n 10 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
python tools/config2_loop_trace.py 2048 2>/dev/null | tail -n 1; python tools/config2_loop_trace.py 128 2>/dev/null | tail -n 1
( timeout 200 python tools/infonce_bench.py --dims 64 ) 2>&1 | grep -v "l1\|linf"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2048 -o kt -- python $R/tools/config2_loop_trace.py 2048 > $O/kt_2048.log 2>&1
find $O/kt_2048 -mindepth 2 -type f -exec mv {} $O/kt_2048/ \; 2>/dev/null
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04j/kt_2048/kt_kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernel us per step", round(tot/1e3/92,1))
    for r in rows[:16]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config3', d['ms_per_step'], d['roofline']['frac'])"
