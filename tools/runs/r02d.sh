#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp exp/lib_NEW3.so $P; touch $P
( time timeout 900 python -m pytest tests/test_gpu_set_transformer.py -q --durations=5 ) > $O/st.log 2>&1
tail -n 25 $O/st.log
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k trajectory ) > $O/traj.log 2>&1
tail -n 12 $O/traj.log
for bp in "32 50" "4 512" "2 2048" "1 4096"; do set -- $bp; timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done
echo "== A/B B=8192";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh NEW3 BK64
cp exp/lib_NEW3.so $P; touch $P
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_b8192 -o kt -- python /root/repo/bench.py --batch 8192 --steps 30 --blocks 1 --no-cpu-baseline --no-extra --no-kernel-timing > /root/repo/$O/prof_b8192.log 2>&1
cd /root/repo; find $O/prof_b8192 -mindepth 2 -type f -exec mv {} $O/prof_b8192/ \;
head -n 30 $O/prof_b8192/kt_kernel_stats.csv
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_st -o kt -- python /root/repo/tools/set_transformer_bench.py --steps 5 > /root/repo/$O/prof_st.log 2>&1
cd /root/repo; find $O/prof_st -mindepth 2 -type f -exec mv {} $O/prof_st/ \;
head -n 25 $O/prof_st/kt_kernel_stats.csv
