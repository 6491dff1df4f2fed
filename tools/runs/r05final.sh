#!/bin/bash
# r05final (part A): the data-parallel step + its extras under ONE RCCL rank (the code path of the driver's SCALE run),
# rocprofv3 stats + FETCH / WRITE / SQ PMC passes of config 3 and config 4 at HEAD
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05final; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 \
  --no-cpu-baseline --force-dp-extras --extra-timeout 400 > $O/bench_dp1.json 2> $O/bench_dp1.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05final/bench_dp1.json") if l.startswith("{")][-1])
print("dp1", d["ms_per_step"], d["config"]["parallelism"], json.dumps(d.get("extra", {}).get("dp_breakdown"))[:900])
print({k: (str(v)[:200]) for k, v in d.get("extra", {}).items() if k != "dp_breakdown"})
PY
tail -n 3 $O/bench_dp1.err
bash tools/collect_profiles.sh gpurun_out/r05final/c3 > $O/collect_c3.log 2>&1
bash tools/collect_profiles.sh gpurun_out/r05final/c4 --features 50 > $O/collect_c4.log 2>&1
ls $O
