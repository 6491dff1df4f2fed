#!/bin/bash
# round-4 GPU call F: software-pipelined single-workgroup attention (parity + kernel times at the notebook's size)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04f; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q -x --deselect tests/test_gpu_set_transformer.py::test_config5_size_4096_particles_flash_all_gradients \
    --deselect tests/test_gpu_set_transformer.py::test_config5_full_depth_six_blocks_at_4096_particles ) > $O/tests_st.log 2>&1; tail -n 4 $O/tests_st.log
for i in 1 2; do timeout 120 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 2>&1 | tail -n 1; done | tee $O/st_notebook_size.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_st -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 20 > $O/kt_st.log 2>&1
cd $R
find $O/kt_st -mindepth 2 -type f -exec mv {} $O/kt_st/ \; 2>/dev/null
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04f/kt_st/kt_kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel us/step", tot/1e3/22)
    for r in rows[:8]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
