#!/bin/bash
# r05a: round-5 parity-gap tests on the GPU (tf.data shuffle-buffer stream in the InfoNCE loop, 1e-3 free-running bar, --ib,
# subset-information ceiling), the effect of the shuffle buffer on config 2 at fixed beta, default bench line of HEAD
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trajectories.py "tests/test_gpu_parity.py" -q -x -s -m gpu \
  -k "trajectory or ib_flag or paper_circuit or infonce_training_loop or train_script" > $O/tests.txt 2>&1
tail -n 12 $O/tests.txt
timeout 300 python tools/stream_effect.py exp/pendulum100 4 1e-3 > $O/stream_effect.txt 2>&1; cat $O/stream_effect.txt | tail -n 12
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05a/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["step_roofline"]["frac"])
for k,v in d["extra"].items(): print(k, json.dumps(v)[:400])
PY
tail -n 3 $O/bench.err
