#!/bin/bash
# round-2 GPU call A: new full-size parity tests, the whole GPU suite, bench at B=65536 and B=8192, B=8192 kernel trace
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -s --durations=10 ) > $O/fullsize.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_dp_and_cache.py -q --durations=10 ) > $O/dpcache.log 2>&1
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_dp_and_cache.py ) > $O/suite.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --batch 8192 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_b8192 -o b8192 -- python /root/repo/bench.py --batch 8192 --steps 30 --blocks 1 --no-cpu-baseline --no-extra --no-kernel-timing > /root/repo/$O/prof_b8192.log 2>&1
cd /root/repo
ls -R $O/prof_b8192 | head -30
tail -3 $O/fullsize.log $O/dpcache.log $O/suite.log
cat $O/bench.json | head -c 3000
echo; cat $O/bench_b8192.json | head -c 1500
