#!/bin/bash
# round-2 validation + measurement run on the final library: whole GPU suite, bench (N=1), B=8192, set-transformer table,
# rocprofv3 kernel stats + PMC passes
export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 ) > $O/gpu_tests.log 2>&1
tail -n 16 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'], d['roofline']['frac'], d['extra'].keys())"
timeout 300 python bench.py --batch 8192 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err
python -c "
import json; d=json.load(open('$O/bench_b8192.json')); print('B=8192', d['value'], d['ms_per_step'], d['timing']['blocks_ms_per_step'])"
for at in flash gemm; do for bp in "32 50" "4 512" "2 2048" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done > $O/st_bench.txt
cat $O/st_bench.txt
bash tools/collect_profiles.sh gpurun_out/r02z/prof > /dev/null 2>&1
ls $O
