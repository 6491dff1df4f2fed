#!/bin/bash
# r05final (part B): whole `-m gpu` suite at HEAD, smoke(), the default bench line (with cpu_baseline + extras), the B = 8192 line,
# config-5 rocprofv3 passes
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05final; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -n 14 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05final/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"])
e=d["extra"]
for k in ("fit_surface","config4_F50","config5_set_transformer","set_transformer_notebook_size","keras_path_default_batch","config2_infonce_loop"):
    v=e.get(k,{})
    print(k, json.dumps({a:b for a,b in v.items() if a not in ("roofline_by_kernel","workload","roofline")})[:700])
print(d["cpu_baseline"])
PY
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192_$i.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_b8192_$i.json') if l.startswith('{')][-1]); print('b8192', d['ms_per_step'], d['value'])"; done
CONFIG5=1 bash tools/collect_profiles.sh gpurun_out/r05final/c5 > $O/collect_c5.log 2>&1
ls $O
