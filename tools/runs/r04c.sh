#!/bin/bash
# round-4 GPU call C: whole GPU suite minus the two 4096-particle tests (new InfoNCE kernels, CU-granular split policy,
# epoch-synchronised trajectory tests), kernel traces of the InfoNCE sequence and of the notebook-size set-transformer step,
# split policy A/B with the CU-granular cost model
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04c; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_set_transformer.py::test_config5_size_4096_particles_flash_all_gradients \
    --deselect tests/test_gpu_set_transformer.py::test_config5_full_depth_six_blocks_at_4096_particles --durations=6 ) > $O/tests.log 2>&1
tail -n 14 $O/tests.log
( timeout 120 python -m pytest tests/test_gpu_trajectories.py -q -s 2>&1 | grep -i "epoch-synchronised" ) > $O/trajectory_errors.txt; cat $O/trajectory_errors.txt
( timeout 200 python tools/infonce_bench.py --dims 64 ) > $O/infonce_bench.txt 2>&1; cat $O/infonce_bench.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_infonce -o kt -- python $R/tools/infonce_bench.py --dims 64 --batches 2048 > $O/kt_infonce.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_st -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 20 > $O/kt_st.log 2>&1
cd $R
for d in $O/kt_infonce $O/kt_st; do find $d -mindepth 2 -type f -exec mv {} $d/ \; 2>/dev/null; done
python - <<'PY'
import csv,glob
for d in ("kt_infonce","kt_st"):
    f=glob.glob(f"gpurun_out/r04c/{d}/kt_kernel_stats.csv")
    if not f: print(d,"no stats"); continue
    rows=list(csv.DictReader(open(f[0])))
    print(d)
    for r in rows[:28]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
for F in 64 50; do
  for cfg in "0 128" "1 128" "0 128" "1 128"; do set -- $cfg
    echo -n "F=$F policy=$1 overhead=$2: "
    DIB_SPLIT_POLICY=$1 DIB_SPLIT_OVERHEAD=$2 timeout 180 python bench.py --features $F --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2> $O/ab.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], d['timing']['blocks_ms_per_step'], {k[4:].replace('_kernel',''): v['ms_per_step'] for k,v in d.get('roofline_by_kernel',{}).items()})"
  done
done 2>&1 | tee $O/split_policy_ab.txt
