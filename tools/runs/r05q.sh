#!/bin/bash
# r05q: instruction-fetch probe (tools/icache_probe.hip), the compact one-launch InfoNCE kernel (A/B by tuning key), fit with the
# epoch's validation batches evaluated together (bench.keras_path_default_batch reports both)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05q; mkdir -p $O
timeout 120 exp/icache_probe 2>&1 | tee $O/icache_probe.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "infonce" ) > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 300 python tools/config2_loop_ab.py 128 2>&1 | grep '^{' | tee $O/loop_ab.txt
timeout 200 python -c "
import json, bench
print(json.dumps(bench.keras_path_default_batch('cuda:0')))
print(json.dumps(bench.keras_path_default_batch('cuda:0')))" 2>&1 | grep '^{' | tee $O/keras.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config2_loop_trace.py 128 > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05q/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:8]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
