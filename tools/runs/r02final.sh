#!/bin/bash
# round-2 FINAL validation + measurement run: whole GPU suite, bench (N=1), B=8192 bench + kernel trace, set-transformer table,
# attention kernel microbench, rocprofv3 kernel stats + PMC passes of the headline
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -q --durations=6 ) > $O/gpu_tests.log 2>&1
tail -n 12 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'], d['roofline']['frac']); print(json.dumps(d['extra'])[:1500])"
timeout 300 python bench.py --batch 8192 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err
python -c "
import json; d=json.load(open('$O/bench_b8192.json')); print('B=8192', d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
for at in flash gemm; do for bp in "32 50" "4 512" "2 2048" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done > $O/st_bench.txt
cat $O/st_bench.txt
for bp in "4 4096" "8 1024" "32 50"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 2>&1 | tail -n 1; done > $O/attn_bench.txt
cat $O/attn_bench.txt
R=$(pwd)
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_b8192 -o kt -- python $R/bench.py --batch 8192 --steps 30 --blocks 1 --no-cpu-baseline --no-extra --no-kernel-timing > $R/$O/prof_b8192.log 2>&1
cd $R; find $O/prof_b8192 -mindepth 2 -type f -exec mv {} $O/prof_b8192/ \;
bash tools/collect_profiles.sh $O/prof > /dev/null 2>&1
ls $O
