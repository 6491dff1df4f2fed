#!/bin/bash
# A/B harness: swap prebuilt library variants in and run the bench (GPU box only)
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
for v in "$@"; do
  cp exp/lib_$v.so $P; touch $P
  [ -n "$EPSPREC" ] && timeout 60 python exp/epsprec.py 2>&1 | tail -1
  timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_$v.json"))
print("$v", d["ms_per_step"], {k.split("<")[0][4:]+k[k.find("<"):] if "<" in k else k[4:]: v["ms_per_step"] for k,v in d["roofline_by_kernel"].items()})
PY
done
