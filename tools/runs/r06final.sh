#!/bin/bash
# r06final: the default bench line, the B = 8192 line, rocprofv3 stats + FETCH / WRITE / SQ PMC passes of config 3 at HEAD,
# kernel stats of the notebook-size set-transformer step
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06final; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 300 python bench.py --batch 8192 --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err; tail -c 200 $O/bench_b8192.json
bash tools/collect_profiles.sh gpurun_out/r06final/c3 > $O/collect_c3.log 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d /tmp/st_prof -- python $R/tools/set_transformer_bench.py --steps 30 --warmup 5 > /dev/null 2>&1
cd $R; f=$(ls /tmp/st_prof/*/*kernel_stats.csv | head -n 1); cp $f $O/set_transformer_notebook_size_kernel_stats.csv
ls $O $O/c3* | head -40
