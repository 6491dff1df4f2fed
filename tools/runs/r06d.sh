#!/bin/bash
# r06d: deferred weight gradients incl. the particle encoder's and the head's (64 launches), grouped-GEMM attention on per-block
# buffers; set-transformer tests; per-launch timeline of one notebook-size step
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_building_blocks.py tests/test_gpu_dp_and_cache.py -m gpu -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -n 8 $O/tests.txt
for rep in 1 2; do
  for args in "--defer 0" "--defer 1"; do
    python tools/set_transformer_bench.py --steps 200 --warmup 20 $args 2>/dev/null | tail -n 1
  done
done | tee $O/set_transformer_ab.txt
cd /tmp && rocprofv3 --kernel-trace -f csv -d /tmp/st_tr -- python $R/tools/set_transformer_bench.py --steps 12 --warmup 4 > $O/bench_under_trace.txt 2>&1
cd $R; f=$(ls /tmp/st_tr/*/*kernel_trace.csv | head -n 1); n=$(python -c "import json;print(int(json.loads(open('$O/bench_under_trace.txt').read().strip().splitlines()[-1])['library_launches_per_step']))")
echo "library launches per step: $n"
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
names=[r["Kernel_Name"] for r in rows]
# all launches of the last steps incl. torch kernels: find the period by the step tail kernel
idx=[i for i,n in enumerate(names) if "dib_reduce_adam" in n or "dib_step_tail" in n]
print("period (all kernels):", [b-a for a,b in zip(idx[-6:],idx[-5:])])
open("$O/period.txt","w").write(str(idx[-2]-idx[-3]))
PY
python tools/st_step_timeline.py $f $(cat $O/period.txt) 2 | tee $O/set_transformer_step_timeline.txt | tail -n 80
