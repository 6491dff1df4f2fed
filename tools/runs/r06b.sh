#!/bin/bash
# r06b: the tests added after r06a (building blocks, new zoo entries, deferred set-transformer weight gradients, InfoNCE D = 100 / 160),
# then the set-transformer A/B at the notebook's size: per-block weight-gradient launches vs the deferred grouped ones, and a sweep
# of the deferred launches' workgroup target
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_building_blocks.py tests/test_gpu_set_transformer.py -m gpu -q -p no:cacheprovider -x > $O/new_tests_a.txt 2>&1; tail -n 6 $O/new_tests_a.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "head_wide or infonce_edge_shapes or fused_128_leaky or e8_leaky or e16_linear or tuning_switchboard" > $O/new_tests_b.txt 2>&1; tail -n 6 $O/new_tests_b.txt
for rep in 1 2; do
  for args in "--defer 0" "--defer 1" "--defer 1 --defer-target 384" "--defer 1 --defer-target 512" "--defer 1 --defer-target 1024" "--defer 1 --defer-target 1536"; do
    python tools/set_transformer_bench.py --steps 200 --warmup 20 $args 2>/dev/null | tail -n 1
  done
done | tee $O/set_transformer_defer_ab.txt
python tools/set_transformer_bench.py --batch 8 --particles 200 --steps 100 --warmup 10 --defer 0 2>/dev/null | tail -n 1 | tee -a $O/set_transformer_defer_ab.txt
python tools/set_transformer_bench.py --batch 8 --particles 200 --steps 100 --warmup 10 --defer 1 2>/dev/null | tail -n 1 | tee -a $O/set_transformer_defer_ab.txt
python tools/set_transformer_bench.py --batch 2 --particles 2048 --steps 30 --warmup 5 --defer 0 2>/dev/null | tail -n 1 | tee -a $O/set_transformer_defer_ab.txt
python tools/set_transformer_bench.py --batch 2 --particles 2048 --steps 30 --warmup 5 --defer 1 2>/dev/null | tail -n 1 | tee -a $O/set_transformer_defer_ab.txt
# per-kernel table of the notebook-size step with the deferred launches
cd /tmp && rocprofv3 --kernel-trace --stats -T -f csv -d /tmp/st_prof -- python $R/tools/set_transformer_bench.py --steps 30 --warmup 5 > /dev/null 2>&1
cd $R; f=$(ls /tmp/st_prof/*/*kernel_stats.csv | head -n 1); cp $f $O/set_transformer_notebook_size_kernel_stats.csv; head -n 30 $f | cut -c1-150
