#!/bin/bash
# r03ah: last call of the round on HEAD (attention backward with full-line dQ partial stores): the 2 x 4096-particle parity
# test, smoke(), the config-5 line, the headline line without extras
O=gpurun_out/r03ah; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "config5_size or stash" > $O/pytest_st_big.log 2>&1; echo "rc=$?" >> $O/pytest_st_big.log); tail -3 $O/pytest_st_big.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --config5-only --steps 4 > $O/config5.json 2> $O/config5.err; python -c "
import json; d=json.load(open('$O/config5.json')); print('config5', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])"
python bench.py --no-extra --no-cpu-baseline > $O/bench_noextra.json 2>> $O/config5.err; python -c "
import json; d=json.load(open('$O/bench_noextra.json')); print('headline', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
