#!/bin/bash
# r06final5: the round's last bench line at the final dispatch rule (cluster size by row-tile count, paired grids within the CU budget)
O=gpurun_out/r06final5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_concurrency.py tests/test_gpu_dp_and_cache.py -q -m gpu 2>&1 | tail -n 4 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 300 python bench.py --batch 8192 --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err; tail -c 200 $O/bench_b8192.json
