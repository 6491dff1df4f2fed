#!/bin/bash
# round-2 FINAL (second) validation + measurement run on the final library (new flash attention, paired fused forward):
# whole GPU suite, bench (N=1), B=8192 bench, set-transformer table, attention kernel microbench, rocprofv3 kernel stats + PMC
export TMPDIR=/tmp
O=gpurun_out/r02final2; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 ) > $O/gpu_tests.log 2>&1
tail -n 12 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'], d['roofline']['frac']); print(json.dumps(d['extra'])[:1800]); print(d['cpu_baseline'])"
timeout 300 python bench.py --batch 8192 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err
python -c "
import json; d=json.load(open('$O/bench_b8192.json')); print('B=8192', d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
for at in flash gemm; do for bp in "32 50" "4 512" "2 2048" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done > $O/st_bench.txt
cut -c1-230 $O/st_bench.txt
for bp in "4 4096" "8 1024" "32 50"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done > $O/attn_bench.txt
cat $O/attn_bench.txt
bash tools/collect_profiles.sh $O/prof > /dev/null 2>&1
ls $O
