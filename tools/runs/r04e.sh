#!/bin/bash
# round-4 GPU call E: single-workgroup attention for neighbourhoods of <= 64 particles (parity zoo + direct test + step time at
# the notebook's size), final weight-gradient split rule (config 3 and 4), InfoNCE loop time (config 2)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04e; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q -x --deselect tests/test_gpu_set_transformer.py::test_config5_size_4096_particles_flash_all_gradients \
    --deselect tests/test_gpu_set_transformer.py::test_config5_full_depth_six_blocks_at_4096_particles ) > $O/tests_st.log 2>&1; tail -n 12 $O/tests_st.log
for i in 1 2; do timeout 120 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 2>&1 | tail -n 1; done | tee $O/st_notebook_size.txt
timeout 120 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 --graphs 1 2>&1 | tail -n 1 | tee -a $O/st_notebook_size.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_st -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 20 > $O/kt_st.log 2>&1
cd $R
find $O/kt_st -mindepth 2 -type f -exec mv {} $O/kt_st/ \; 2>/dev/null
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04e/kt_st/kt_kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel us/step", tot/1e3/22)
    for r in rows[:14]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
for F in 64 50; do
  for pol in 0 1 0 1; do
    echo -n "F=$F policy=$pol: "
    DIB_SPLIT_POLICY=$pol timeout 180 python bench.py --features $F --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2> $O/ab.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], d['timing']['blocks_ms_per_step'], {k[4:].replace('_kernel',''): v['ms_per_step'] for k,v in d.get('roofline_by_kernel',{}).items()})"
  done
done 2>&1 | tee $O/split_policy_final_ab.txt
python - <<'PY' 2>&1 | tee gpurun_out/r04e/config2_loop.txt
import sys, json
sys.path.insert(0, ".")
import bench
print(json.dumps([bench.config2_infonce_loop("cuda:0", 128), bench.config2_infonce_loop("cuda:0", 2048)]))
PY
