#!/bin/bash
# r04o: default bench line with the PCIe-inclusive figures of the fit surface (whole model.fit call from host numpy arrays,
# dataset upload rate) - tier rule: the boundary hands over host buffers, so the inclusive rate is stated (never the headline)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04o; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04o/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"], d["timing"]["blocks_ms_per_step"])
print(json.dumps({a:b for a,b in d["extra"]["fit_surface"].items() if a!="workload"}))
PY
tail -n 5 $O/bench.err
