#!/bin/bash
# A/B harness for kernel experiments (GPU box only).  Build library variants on the CPU side with tools/build_variant.sh
# (exp/lib_TAG.so: git-ignored, travels with gpurun), then on the box:  bash tools/ab_bench.sh BASE X Y
# Each variant is selected through DIB_LIB_PATH (dib_amd/_lib.py) - the product libdib_hip.so is never touched - and timed
# with bench.py; one line per variant with the per-kernel ms/step.  BATCH=8192 selects the per-GPU batch of 8-GPU strong
# scaling; TAG labels the output files; REPS repeats the whole list (interleaved, for box drift); extra environment
# (DIB_FORCE_TILE0=..., DIB_L3_HALVE=0, ...) passes through to the library's tile-rule knobs.
for rep in $(seq 1 ${REPS:-1}); do
for v in "$@"; do
  o=gpurun_out/exp_${v}${TAG:+_$TAG}
  DIB_LIB_PATH=exp/lib_$v.so timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-extra ${BATCH:+--batch $BATCH} > $o.json 2> $o.err
  python - <<PY
import json
d=json.load(open("$o.json"))
print("$v ${TAG}", d["ms_per_step"], d["timing"]["blocks_ms_per_step"], {k.split("<")[0][4:]+k[k.find("<"):] if "<" in k else k[4:]: v["ms_per_step"] for k,v in d.get("roofline_by_kernel",{}).items()})
PY
done
done
