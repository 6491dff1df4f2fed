#!/bin/bash
# A/B harness for kernel experiments (GPU box only): build library variants on the CPU side, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDIB_EXP_X <pkg>/csrc/dib_api.hip -o exp/lib_X.so
# (exp/*.so is git-ignored but travels with gpurun), then on the box:  bash tools/ab_bench.sh BASE X Y
# Each variant is swapped in as the product library and timed with bench.py; one line per variant with the per-kernel
# ms/step.  EPSPREC=1 additionally prints the noise generator's error against the fp64 oracle.
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
for v in "$@"; do
  cp exp/lib_$v.so $P; touch $P
  [ -n "$EPSPREC" ] && timeout 60 python tools/eps_precision.py 2>&1 | tail -1
  timeout 120 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BATCH:+--batch $BATCH} > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_$v.json"))
print("$v", d["ms_per_step"], {k.split("<")[0][4:]+k[k.find("<"):] if "<" in k else k[4:]: v["ms_per_step"] for k,v in d["roofline_by_kernel"].items()})
PY
done
