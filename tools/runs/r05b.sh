#!/bin/bash
# r05b: the one-launch step tail (csrc/dib_tail.h) + dib_set_tuning: whole GPU suite, default bench line
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05b; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -x -m gpu ) > $O/tests.txt 2>&1
tail -n 15 $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05b/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["step_roofline"]["frac"])
for k,v in d["extra"].items(): print(k, json.dumps(v)[:600])
PY
tail -n 3 $O/bench.err
