#!/bin/bash
# r03ac: bench.py after the headline-before-extras restructure (default line with all N = 1 extras), smoke, and the DP / cache
# GPU tests (1-rank RCCL paths of both models) on HEAD
O=gpurun_out/r03ac; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
(time python bench.py > $O/bench.json 2> $O/bench.err) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], list(d['extra'].keys()), d['extra']['config5_set_transformer'].get('ms_per_step'), d['cpu_baseline']['value'])"
(timeout 900 python -m pytest tests/test_gpu_dp_and_cache.py -m gpu -q -x > $O/pytest_dp.log 2>&1; echo "rc=$?" >> $O/pytest_dp.log); tail -3 $O/pytest_dp.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo "rc=$?"); tail -c 600 $O/bench_rccl1.json
