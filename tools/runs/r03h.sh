#!/bin/bash
# r03h: validation + measurement of the round-3 library: the whole `-m gpu` suite, smoke(), the default bench line, the four
# rocprofv3 passes of the config-5 step in score-stash mode (CONFIG5=1), the notebook-size table
O=gpurun_out/r03h; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -5 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
CONFIG5=1 bash tools/collect_profiles.sh $O/p5 > $O/collect5.log 2>&1
for a in "--batch 32 --particles 50 --steps 50 --warmup 5" "--batch 4 --particles 512 --steps 20" "--batch 2 --particles 2048 --steps 10" "--batch 4 --particles 4096 --features 16 --steps 4"; do python tools/set_transformer_bench.py $a; done 2>&1 | grep -v amdgpu.ids | tee $O/st_table.txt
python tools/attn_bench.py --batch 4 --particles 4096 --stash 1 2>/dev/null | tee $O/attn.txt; python tools/attn_bench.py --batch 4 --particles 4096 --stash 0 2>/dev/null | tee -a $O/attn.txt
