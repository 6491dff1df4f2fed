#!/bin/bash
# GPU box: hipGraph replay of the set-transformer step (test + eager-vs-replay timing + kernel traces), then the four
# rocprofv3 passes of the headline encoder-bank step (tools/collect_profiles.sh).
O=gpurun_out/r03e; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "graph_replay or train_steps or fit_loop" > $O/pytest_graph.log 2>&1; echo "rc=$?" >> $O/pytest_graph.log); tail -4 $O/pytest_graph.log
for g in 0 1 0 1; do python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 --graphs $g; done 2>&1 | grep -v amdgpu.ids | tee $O/st_notebook_size.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for g in 0 1; do timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_g$g -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 20 --warmup 3 --graphs $g > $R/$O/kt_g$g.log 2>&1; done
cd $R
for d in $O/kt_g0 $O/kt_g1; do find $d -mindepth 2 -type f -exec mv {} $d/ \; 2>/dev/null; done
bash tools/collect_profiles.sh $O/p > $O/collect.log 2>&1; tail -2 $O/collect.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench_headline.err
