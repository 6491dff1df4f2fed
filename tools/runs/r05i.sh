#!/bin/bash
# r05i: double-buffered row-tile kernels (ring prefetch reverted), merged weight gradients, head inputs prefetched: whole GPU
# suite, phase timing, default-batch trace, InfoNCE loop
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05i; mkdir -p $O
( time timeout 1800 python -m pytest tests -q -x -m gpu ) > $O/tests.txt 2>&1
tail -n 8 $O/tests.txt
DIB_LIB_PATH=$R/exp/lib_STIMING.so timeout 300 python tools/small_phase_timing.py 128 > $O/phase_b128.txt 2>&1; cat $O/phase_b128.txt | tail -n 6
cd /tmp
DIB_SMALL_EPOCHS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/small_batch_bench.py > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05i/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
pairs=(3+50)*8
print("kernel us per (train + validation) step pair", round(tot/1e3/pairs,1))
for r in rows[:10]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/pairs,2), round(float(r["AverageNs"])/1e3,2), r["MinNs"], r["MaxNs"])
PY
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
timeout 120 python tools/small_batch_bench.py 2>&1 | tail -n 1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o kt -- python $R/tools/config2_loop_trace.py 128 > $O/kt2.log 2>&1
find $O/kt2 -mindepth 2 -type f -exec mv {} $O/kt2/ \; 2>/dev/null
cd $R
rm -f $O/kt2/*kernel_trace.csv $O/kt2/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05i/kt2/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:22]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
timeout 200 python tools/config2_loop_trace.py 128 2>&1 | tail -n 1
timeout 200 python tools/config2_loop_trace.py 2048 2>&1 | tail -n 1
