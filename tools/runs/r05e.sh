#!/bin/bash
# r05e: small-batch row-tile kernels: equivalence tests, the whole GPU suite, default-batch + InfoNCE-loop kernel traces
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "small_batch_row_tile or step_tail" ) > $O/tests_small.txt 2>&1
tail -n 12 $O/tests_small.txt
( time timeout 1800 python -m pytest tests -q -x -m gpu ) > $O/tests.txt 2>&1
tail -n 8 $O/tests.txt
cd /tmp
DIB_SMALL_EPOCHS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/small_batch_bench.py > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
tail -n 1 $O/kt.log
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05e/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
pairs=(3+50)*8
print("kernel us per (train + validation) step pair", round(tot/1e3/pairs,1))
for r in rows[:30]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/pairs,2), round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
timeout 120 python tools/small_batch_bench.py 2>&1 | tail -n 1
timeout 300 python tools/infonce_bench.py 2>&1 | tail -n 12
