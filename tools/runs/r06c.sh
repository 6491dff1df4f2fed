#!/bin/bash
# r06c: q / k / v projections fused into the set transformer's chain launches: tests, then the A/B at the notebook's size
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_building_blocks.py -m gpu -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -n 12 $O/tests.txt
for rep in 1 2; do
  for args in "--defer 0" "--defer 1 --fuse-qkv 0 --defer-target 1536" "--defer 1 --fuse-qkv 1 --defer-target 1536" "--defer 1 --fuse-qkv 1 --defer-target 2048" "--defer 1 --fuse-qkv 1 --defer-target 3072"; do
    python tools/set_transformer_bench.py --steps 200 --warmup 20 $args 2>/dev/null | tail -n 1
  done
done | tee $O/set_transformer_fuse_ab.txt
for args in "--batch 8 --particles 200 --fuse-qkv 0" "--batch 8 --particles 200 --fuse-qkv 1" "--batch 2 --particles 2048 --steps 30 --fuse-qkv 0" "--batch 2 --particles 2048 --steps 30 --fuse-qkv 1"; do
  python tools/set_transformer_bench.py --steps 100 --warmup 10 $args 2>/dev/null | tail -n 1
done | tee -a $O/set_transformer_fuse_ab.txt
cd /tmp && rocprofv3 --kernel-trace --stats -T -f csv -d /tmp/st_prof -- python $R/tools/set_transformer_bench.py --steps 30 --warmup 5 > /dev/null 2>&1
cd $R; f=$(ls /tmp/st_prof/*/*kernel_stats.csv | head -n 1); cp $f $O/set_transformer_notebook_size_kernel_stats.csv; head -n 12 $f | cut -c1-150
