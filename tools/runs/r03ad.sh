#!/bin/bash
# r03ad: streaming skinny-K GEMM (dib_gemm_skinny_k) for the set transformer's q / k / v projections and context gradient:
# kernel unit tests, the set-transformer parity file (the 2 x 4096-particle case runs on it), and the same-box A/B of the
# config-5 step and of two smaller shapes (DIB_SKINNY_K_MIN_TOKENS=1000000000 = the tiled grouped GEMM as before)
O=gpurun_out/r03ad; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "skinny_k" > $O/pytest_kernel.log 2>&1; echo "rc=$?" >> $O/pytest_kernel.log); tail -3 $O/pytest_kernel.log
(timeout 1500 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x > $O/pytest_st.log 2>&1; echo "rc=$?" >> $O/pytest_st.log); tail -3 $O/pytest_st.log
for rep in 1 2; do for v in 1000000000 2048; do echo "min_tokens=$v $(DIB_SKINNY_K_MIN_TOKENS=$v python bench.py --config5-only --steps 4 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"])')"; done; done | tee $O/config5_ab.txt
for v in 1000000000 2048; do for a in "--batch 4 --particles 512 --steps 20" "--batch 2 --particles 2048 --steps 10"; do echo "min_tokens=$v $a $(DIB_SKINNY_K_MIN_TOKENS=$v python tools/set_transformer_bench.py $a 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; done; done | tee $O/st_ab.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o kt -- python $R/bench.py --config5-only --steps 3 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -mindepth 2 -type f -exec mv {} $O/prof/ \; 2>/dev/null; ls $O/prof | head
