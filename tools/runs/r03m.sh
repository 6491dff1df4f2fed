#!/bin/bash
# r03m: secondary paths at their working sizes: wall time + rocprofv3 kernel stats of each
O=gpurun_out/r03m; mkdir -p $O
python tools/secondary_paths_bench.py infonce mi 2>$O/bench.err | tee $O/secondary.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for w in infonce mi; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$w -o kt -- python $R/tools/secondary_paths_bench.py $w > $R/$O/kt_$w.log 2>&1; done
cd $R; for d in $O/kt_infonce $O/kt_mi; do find $d -mindepth 2 -type f -exec mv {} $d/ \; 2>/dev/null; rm -f $d/kt_kernel_trace.csv; done
head -12 $O/kt_infonce/kt_kernel_stats.csv | cut -c1-150; head -8 $O/kt_mi/kt_kernel_stats.csv | cut -c1-150
