#!/bin/bash
# r06h: set transformer - the projections' input gradient inside the <= 64-particle attention backward (dib_attention_bwd_proj),
# weight fragments of the forward projections requested up front; tests, A/B, per-launch timeline
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_dp_and_cache.py tests/test_gpu_building_blocks.py -m gpu -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -n 8 $O/tests.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "dense_stack or companion or infonce_training_loop" > $O/tests_b.txt 2>&1; tail -n 3 $O/tests_b.txt
for rep in 1 2; do
  for args in "--attn-proj 0 --attn-bwd-proj 0" "--attn-proj 1 --attn-bwd-proj 0" "--attn-proj 0 --attn-bwd-proj 1" "--attn-proj 1 --attn-bwd-proj 1"; do
    python tools/set_transformer_bench.py --steps 200 --warmup 20 $args 2>/dev/null | tail -n 1
  done
done | tee $O/set_transformer_ab.txt
for args in "--batch 8 --particles 200" "--batch 2 --particles 2048 --steps 30"; do
  python tools/set_transformer_bench.py --steps 100 --warmup 10 $args 2>/dev/null | tail -n 1
done | tee -a $O/set_transformer_ab.txt
cd /tmp && rocprofv3 --kernel-trace -f csv -d /tmp/st_tr -- python $R/tools/set_transformer_bench.py --steps 12 --warmup 4 > $O/bench_under_trace.txt 2>&1
cd $R; f=$(ls /tmp/st_tr/*/*kernel_trace.csv | head -n 1)
python - <<PY
import csv
names=[r["Kernel_Name"] for r in csv.DictReader(open("$f"))]
idx=[i for i,n in enumerate(names) if "dib_reduce_adam" in n or "dib_step_tail" in n]
open("$O/period.txt","w").write(str(idx[-2]-idx[-3]))
PY
python tools/st_step_timeline.py $f $(cat $O/period.txt) 2 | tee $O/set_transformer_step_timeline.txt | tail -n 75
