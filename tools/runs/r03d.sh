#!/bin/bash
# r03d: the whole GPU suite + the default bench line on the library with score-stash attention and three gradient buckets
mkdir -p gpurun_out/r03d
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03d/pytest_all.log 2>&1
python bench.py > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
