#!/bin/bash
# last bench line of round 2 on the final library (flash attention forward at 2 waves/SIMD)
export TMPDIR=/tmp
O=gpurun_out/r02final3; mkdir -p $O
timeout 130 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'], d['roofline']['frac']); print(json.dumps(d['extra'])[:900]); print(d['cpu_baseline']['value'])"
