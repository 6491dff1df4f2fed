#!/bin/bash
# r05final3: HEAD at the end of round 5 (8-wave attention backward, 4-loads-per-trip tile loader on top of r05final2): the GPU
# test files r05u did not run (trajectories, DP / cache, bench launcher), smoke(), default bench line, B = 8192 line
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05final3; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_trajectories.py tests/test_gpu_dp_and_cache.py -m gpu -q > $O/pytest_rest.log 2>&1; echo "rc=$?" >> $O/pytest_rest.log); tail -n 4 $O/pytest_rest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05final3/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"])
e=d["extra"]
for k in ("config4_F50","config5_set_transformer","set_transformer_notebook_size","keras_path_default_batch","config2_infonce_loop"):
    v=e.get(k,{})
    print(k, json.dumps({a:b for a,b in v.items() if a not in ("roofline_by_kernel","workload","roofline")})[:600])
print(d["cpu_baseline"]["value"])
PY
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_b8192.json') if l.startswith('{')][-1]); print('b8192', d['ms_per_step'], d['value'])"
