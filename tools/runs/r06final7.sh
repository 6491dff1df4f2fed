#!/bin/bash
# r06final7: the round's last validation + measurement call at the final library: the WHOLE GPU suite under rocprofv3 (coverage
# record), smoke() on every dispatch path, the default bench line, the B = 8192 line
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06final7; mkdir -p $O /tmp/cov
cd $R
DIB_COVERAGE_LOG=/tmp/cov/tests.tsv timeout 3000 rocprofv3 --kernel-trace --marker-trace -M -f csv -d /tmp/cov/trace -- \
  python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite_profiled.txt 2>&1
echo "suite rc $?"; grep -v "rocprofv3\|^W2026\|^E2026" $O/gpu_suite_profiled.txt | tail -n 12
python tools/kernel_coverage.py build /tmp/cov/trace /tmp/cov/tests.tsv $O/r06_suite_kernel_coverage.txt 2>&1 | tail -n 3
grep "^MISSING\|^# " $O/r06_suite_kernel_coverage.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 300 python bench.py --batch 8192 --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_b8192.json 2> $O/bench_b8192.err; tail -c 200 $O/bench_b8192.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_default -o kt -- python $R/tools/small_batch_bench.py > $O/default_batch_prof.log 2>&1; cp /tmp/prof_default/*kernel_stats.csv $O/default_batch_kernel_stats.csv 2>/dev/null || find /tmp/prof_default -name "*kernel_stats.csv" -exec cp {} $O/default_batch_kernel_stats.csv \;
cd $R
