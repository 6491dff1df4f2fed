#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q --durations=3 ) > $O/st.log 2>&1
tail -n 25 $O/st.log
for at in flash gemm; do for bp in "32 50" "4 512" "2 2048" "2 4096" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done
R=$(pwd)
cd /tmp && DIB_ST_ATTENTION=flash timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_flash -o kt -- python $R/tools/set_transformer_bench.py --batch 2 --particles 4096 --steps 3 > $R/$O/prof_flash.log 2>&1
cd $R; find $O/prof_flash -mindepth 2 -type f -exec mv {} $O/prof_flash/ \;
head -n 8 $O/prof_flash/kt_kernel_stats.csv | cut -c1-160
