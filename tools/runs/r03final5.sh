#!/bin/bash
# r03final5: last validation of the round on HEAD (library with dib_gemm_skinny_k): whole `-m gpu` suite, smoke(), the default
# bench line, the B = 8192 line
O=gpurun_out/r03final5; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log); tail -4 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['extra']['config5_set_transformer'].get('ms_per_step'), d['extra']['config4_F50']['value'], d['extra']['set_transformer_notebook_size'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_b8192.json')); print('b8192', d['ms_per_step'], d['value'])"
