#!/bin/bash
# r05p: the custom loop at its default batch in 9 launches: output encoder on the row-tile kernels (dib_mlp_small_*), riding in
# the X model's integration grids (dib_integration_fwd_and_mlp_fwd / dib_backward_and_mlp_bwd), one-launch InfoNCE
# (dib_infonce_small_kernel): equivalence + oracle tests, loop trajectories, same-box A/B by tuning key, kernel trace
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05p; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "infonce or dense or companion or tuning or small_batch or custom" ) > $O/tests.txt 2>&1
tail -n 6 $O/tests.txt
( time timeout 600 python -m pytest tests/test_gpu_trajectories.py -q -x -m gpu -k "infonce" ) > $O/tests_traj.txt 2>&1
tail -n 6 $O/tests_traj.txt
timeout 300 python tools/config2_loop_ab.py 128 1024 2>&1 | grep '^{' | tee $O/loop_ab.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config2_loop_trace.py 128 > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05p/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:16]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
