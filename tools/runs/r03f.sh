#!/bin/bash
# GPU box: set-transformer parity after the split-K / slab retune, notebook-size timing + trace, attention early-prefetch A/B
O=gpurun_out/r03f; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x > $O/pytest_st.log 2>&1; echo "rc=$?" >> $O/pytest_st.log); tail -4 $O/pytest_st.log
for g in 0 1; do python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 --graphs $g; done 2>&1 | grep -v amdgpu.ids | tee $O/st_notebook_size.txt
python tools/set_transformer_bench.py --batch 4 --particles 512 --steps 20 2>&1 | grep -v amdgpu.ids | tee -a $O/st_notebook_size.txt
python tools/set_transformer_bench.py --batch 2 --particles 2048 --steps 10 2>&1 | grep -v amdgpu.ids | tee -a $O/st_notebook_size.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_g0 -o kt -- python $R/tools/set_transformer_bench.py --batch 32 --particles 50 --steps 20 --warmup 3 --graphs 0 > $R/$O/kt_g0.log 2>&1
cd $R; find $O/kt_g0 -mindepth 2 -type f -exec mv {} $O/kt_g0/ \; 2>/dev/null
for rep in 1 2; do for v in ATT0 ATTE; do for s in 1 0; do echo "$v stash=$s $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash $s 2>/dev/null)"; done; done; done 2>&1 | tee $O/attn_ab.txt
