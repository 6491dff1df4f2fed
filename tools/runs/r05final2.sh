#!/bin/bash
# r05final2: HEAD after the loop / validation launch merges (output encoder on the row-tile kernels riding in the X model's
# integration grids, one-launch InfoNCE, fit's merged validation batches): whole `-m gpu` suite, smoke(), default bench line,
# B = 8192 line, kernel trace of the config-2 loop step at B = 128
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05final2; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -n 12 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05final2/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"])
e=d["extra"]
for k in ("fit_surface","config4_F50","config5_set_transformer","set_transformer_notebook_size","keras_path_default_batch","config2_infonce_loop"):
    v=e.get(k,{})
    print(k, json.dumps({a:b for a,b in v.items() if a not in ("roofline_by_kernel","workload","roofline")})[:800])
print(d["cpu_baseline"])
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_b8192.json') if l.startswith('{')][-1]); print('b8192', d['ms_per_step'], d['value'])"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config2_loop_trace.py 128 > $O/kt.log 2>&1
find $O/kt -mindepth 2 -type f -exec mv {} $O/kt/ \; 2>/dev/null
cd $R
rm -f $O/kt/*kernel_trace.csv $O/kt/*agent_info.csv
ls $O
