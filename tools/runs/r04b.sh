#!/bin/bash
# round-4 GPU call B: InfoNCE similarity + gradients on the MFMAs (parity + kernel times), trajectory tests with the
# displacement-relative parameter bound, weight-gradient split policy A/B (config 3 and config 4), fit_surface after the
# permutation prefetch
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_trajectories.py tests/test_gpu_parity.py -q -x -k "infonce or trajectory or dense_stack" -s --durations=5 ) > $O/tests.log 2>&1
tail -n 12 $O/tests.log
( timeout 200 python tools/infonce_bench.py --dims 64 ) > $O/infonce_bench.txt 2>&1; cat $O/infonce_bench.txt
for F in 64 50; do
  for cfg in "0 128" "1 0" "1 128" "1 512" "0 128" "1 128"; do set -- $cfg
    echo -n "F=$F policy=$1 overhead=$2: "
    DIB_SPLIT_POLICY=$1 DIB_SPLIT_OVERHEAD=$2 timeout 180 python bench.py --features $F --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2> $O/ab.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], d['timing']['blocks_ms_per_step'], {k[4:].replace('_kernel',''): v['ms_per_step'] for k,v in d.get('roofline_by_kernel',{}).items()}, d.get('other_timed_kernels_ms_per_step'))"
  done
done 2>&1 | tee $O/split_policy_ab.txt
( timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04b/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["frac"])
e=d["extra"]
print(json.dumps(e.get("fit_surface")))
print({k:v for k,v in e.get("config4_F50").items() if k!="roofline_by_kernel"})
print(json.dumps(e.get("config2_infonce_loop")))
PY
