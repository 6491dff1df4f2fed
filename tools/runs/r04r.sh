#!/bin/bash
# r04r: default bench line with cpu_baseline.reference_default_size (the CPU restatement at train.py's own default size, same box)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04r; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04r/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"], d["timing"]["blocks_ms_per_step"])
print(json.dumps(d["cpu_baseline"]))
print(json.dumps(d["extra"]["keras_path_default_batch"]))
PY
tail -n 3 $O/bench.err
