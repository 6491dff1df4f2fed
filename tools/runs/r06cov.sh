#!/bin/bash
# r06cov: the whole GPU suite under rocprofv3 --kernel-trace --marker-trace with a ROCTx range per test (tests/conftest.py) ->
# profiles/r06_suite_kernel_coverage.txt (tools/kernel_coverage.py); smoke() on every dispatch path; the default bench line
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06cov; mkdir -p $O /tmp/cov
cd $R
DIB_COVERAGE_LOG=/tmp/cov/tests.tsv timeout 3000 rocprofv3 --kernel-trace --marker-trace -M -f csv -d /tmp/cov/trace -- \
  python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite_profiled.txt 2>&1
echo "suite rc $?"; grep -v "rocprofv3\|^W2026\|^E2026" $O/gpu_suite_profiled.txt | tail -n 15
python tools/kernel_coverage.py build /tmp/cov/trace /tmp/cov/tests.tsv $O/r06_suite_kernel_coverage.txt 2>&1 | tail -n 3
grep "^MISSING\|^# " $O/r06_suite_kernel_coverage.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
