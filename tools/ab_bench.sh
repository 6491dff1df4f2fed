#!/bin/bash
# A/B harness for kernel experiments (GPU box only): build library variants on the CPU side, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDIB_EXP_X <pkg>/csrc/dib_api.hip -o exp/lib_X.so
# (exp/*.so is git-ignored but travels with gpurun), then on the box:  bash tools/ab_bench.sh BASE X Y
# Each variant is swapped in as the product library and timed with bench.py; one line per variant with the per-kernel
# ms/step.  BATCH=8192 selects the per-GPU batch of 8-GPU strong scaling; TAG labels the output files; extra environment
# (DIB_FORCE_TILE0=..., DIB_L3_HALVE=0, ...) passes through to the library's tile-rule knobs.
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
for v in "$@"; do
  cp exp/lib_$v.so $P; touch $P
  o=gpurun_out/exp_${v}${TAG:+_$TAG}
  timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-extra ${BATCH:+--batch $BATCH} > $o.json 2> $o.err
  python - <<PY
import json
d=json.load(open("$o.json"))
print("$v ${TAG}", d["ms_per_step"], d["timing"]["blocks_ms_per_step"], {k.split("<")[0][4:]+k[k.find("<"):] if "<" in k else k[4:]: v["ms_per_step"] for k,v in d.get("roofline_by_kernel",{}).items()})
PY
done
