#!/bin/bash
# r05h: ring prefetch in the row-tile kernels, merged weight-gradient launch (dib_backward), overlapping edge tile of the
# weight-gradient GEMM: tests, phase timing, default-batch trace, config-4 step
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05h; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_dp_and_cache.py tests/test_gpu_fullsize.py -q -x -m gpu ) > $O/tests.txt 2>&1
tail -n 8 $O/tests.txt
DIB_LIB_PATH=$R/exp/lib_STIMING.so timeout 300 python tools/small_phase_timing.py 128 > $O/phase_b128.txt 2>&1; cat $O/phase_b128.txt | tail -n 6
cd /tmp
DIB_SMALL_EPOCHS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/small_batch_bench.py > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05h/kt/kt_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
pairs=(3+50)*8
print("kernel us per (train + validation) step pair", round(tot/1e3/pairs,1))
for r in rows[:10]: print("  ", r["Name"][:80].ljust(80), round(int(r["Calls"])/pairs,2), round(float(r["AverageNs"])/1e3,2), r["MinNs"], r["MaxNs"])
PY
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
timeout 120 python tools/small_batch_bench.py 2>&1 | tail -n 1
timeout 200 python tools/config2_loop_trace.py 128 2>&1 | tail -n 1
timeout 200 python tools/config2_loop_trace.py 2048 2>&1 | tail -n 1
timeout 300 python bench.py --features 50 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>$O/b50.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['ms_per_step'], d['timing']['blocks_ms_per_step'], {k:(v['avg_launch_ms'],v['frac']) for k,v in d['roofline_by_kernel'].items()})"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>$O/b64.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['ms_per_step'], d['timing']['blocks_ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --steps 20 --warmup 3 --batch 8192 --no-cpu-baseline --no-extra 2>$O/b8192.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B8192', d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
