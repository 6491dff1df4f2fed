#!/bin/bash
# r03i: the new GPU tests (staged backward hooks, graph release, plan LRU), InfoNCE kernel times at working sizes
O=gpurun_out/r03i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dp_and_cache.py tests/test_gpu_set_transformer.py -m gpu -q -x -k "three_bucket or released_step or lru or workspace_cache or graph_replay" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log); tail -6 $O/pytest_new.log
python tools/infonce_bench.py 2>/dev/null | tee $O/infonce_bench.txt
