#!/bin/bash
# r06a: (1) the whole GPU suite under rocprofv3 --kernel-trace --marker-trace with a ROCTx range per test (tests/conftest.py) ->
# the kernel-coverage record (tools/kernel_coverage.py); (2) smoke() on every dispatch path; (3) the bench line on this box;
# (4) the data-parallel step + dp_breakdown at the per-GPU operating point B = 8192 under ONE RCCL rank
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06a; mkdir -p $O /tmp/cov
cd $R
DIB_COVERAGE_LOG=/tmp/cov/tests.tsv timeout 3000 rocprofv3 --kernel-trace --marker-trace -M -f csv -d /tmp/cov/trace -- \
  python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite_profiled.txt 2>&1
echo "suite rc $?"; tail -n 15 $O/gpu_suite_profiled.txt
ls -la /tmp/cov/trace/* | head; du -sh /tmp/cov
python tools/kernel_coverage.py build /tmp/cov/trace /tmp/cov/tests.tsv $O/r06_suite_kernel_coverage.txt 2>&1 | tail -n 5
grep -c "^KERNEL" $O/r06_suite_kernel_coverage.txt; grep "^MISSING\|^# " $O/r06_suite_kernel_coverage.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 50 --warmup 5 \
  --batch 8192 --no-cpu-baseline --force-dp-extras --only-dp-breakdown --extra-timeout 400 > $O/bench_dp1_b8192.json 2> $O/bench_dp1_b8192.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06a/bench_dp1_b8192.json") if l.startswith("{")][-1])
print("dp1 b8192", d["ms_per_step"], d["config"].get("parallelism"), json.dumps(d.get("extra", {}).get("dp_breakdown")))
print(json.dumps(d.get("roofline_by_kernel"))[:1500])
PY
tail -n 3 $O/bench_dp1_b8192.err
