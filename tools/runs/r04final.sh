#!/bin/bash
# r04final: validation + measurement of the round-4 library: whole `-m gpu` suite, smoke(), the default bench line, the B = 8192
# line, rocprofv3 stats + FETCH / WRITE / SQ PMC passes of config 3, config 4 (--features 50) and config 5, InfoNCE kernel times
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -n 14 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04final/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"])
e=d["extra"]
for k in ("fit_surface","config4_F50","config5_set_transformer","set_transformer_notebook_size","keras_path_default_batch","config2_infonce_loop"):
    v=e.get(k,{})
    print(k, json.dumps({a:b for a,b in v.items() if a not in ("roofline_by_kernel","workload","roofline")})[:600])
print(d["cpu_baseline"])
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_b8192.json') if l.startswith('{')][-1]); print('b8192', d['ms_per_step'], d['value'])"
( timeout 200 python tools/infonce_bench.py --dims 8 64 ) > $O/infonce_bench.txt 2>&1; grep '"D": 64' $O/infonce_bench.txt
for i in 1 2; do timeout 120 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 2>&1 | tail -n 1; done | tee $O/st_notebook_size.txt
bash tools/collect_profiles.sh gpurun_out/r04final/c3 > $O/collect_c3.log 2>&1
bash tools/collect_profiles.sh gpurun_out/r04final/c4 --features 50 > $O/collect_c4.log 2>&1
CONFIG5=1 bash tools/collect_profiles.sh gpurun_out/r04final/c5 > $O/collect_c5.log 2>&1
ls $O
