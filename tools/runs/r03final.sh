#!/bin/bash
# r03final: validation + measurement of the final round-3 library: the whole `-m gpu` suite, smoke(), the default bench line,
# rocprofv3 stats + PMC passes of the config-3 step and of the config-5 step (stash mode), step tables of the secondary paths
O=gpurun_out/${RUN_TAG:-r03final}; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -5 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json
bash tools/collect_profiles.sh $O/p3 > $O/collect3.log 2>&1
CONFIG5=1 bash tools/collect_profiles.sh $O/p5 > $O/collect5.log 2>&1
for a in "--batch 32 --particles 50 --steps 50 --warmup 5" "--batch 4 --particles 512 --steps 20" "--batch 2 --particles 2048 --steps 10" "--batch 4 --particles 4096 --features 16 --steps 4"; do python tools/set_transformer_bench.py $a; done 2>&1 | grep -v amdgpu.ids | tee $O/st_table.txt
python tools/secondary_paths_bench.py infonce mi 2>/dev/null | tee $O/secondary.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err
