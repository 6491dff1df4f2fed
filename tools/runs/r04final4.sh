#!/bin/bash
# r04final4: HEAD at the end of round 4 (library unchanged since r04final3; + the paper-circuit science-level test, the PCIe-inclusive
# fit figures in bench.py): whole `-m gpu` suite, smoke(), default bench line, B = 8192 line
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04final4; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -n 10 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04final4/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"], d["timing"]["blocks_ms_per_step"])
print({k:(v["ms_per_step"],v["frac"]) for k,v in d["roofline_by_kernel"].items()})
e=d["extra"]
for k in ("fit_surface","config4_F50","config5_set_transformer","set_transformer_notebook_size","keras_path_default_batch","config2_infonce_loop"):
    v=e.get(k,{})
    print(k, json.dumps({a:b for a,b in v.items() if a not in ("roofline_by_kernel","workload","roofline")})[:500])
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_b8192.json') if l.startswith('{')][-1]); print('b8192', d['ms_per_step'], d['value'])"
