#!/bin/bash
# r04p: the paper's 10-input Boolean circuit through DistributedIBNet.fit on compressed schedules: in which order do the gates'
# KL fall below 0.1 bits, against the order the reference notebook's own TensorFlow run printed; + the test built on it
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04p; mkdir -p $O
timeout 600 python tools/paper_circuit_run.py > $O/paper_circuit_b.txt 2> $O/err.txt; cat $O/paper_circuit_b.txt; tail -n 3 $O/err.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "science_level" 2>&1 | tail -n 5 | tee $O/pytest_science.txt
