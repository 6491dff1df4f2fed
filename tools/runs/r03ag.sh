#!/bin/bash
# r03ag: attention backward, dQ partials stored as whole 128-byte lines (rows turned through a wave-private LDS patch):
# AQN = 32-byte pieces per row and instruction (state of r03final5), AQL = full lines (product).  Parity file on the product,
# same-box kernel A/B, WRITE_SIZE of the launch under both.
O=gpurun_out/r03ag; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "not config5_size" > $O/pytest_st.log 2>&1; echo "rc=$?" >> $O/pytest_st.log); tail -3 $O/pytest_st.log
for rep in 1 2 3; do for v in AQN AQL; do echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash 1 2>/dev/null)"; done; done | tee $O/attn_ab.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in AQN AQL; do DIB_LIB_PATH=$R/exp/lib_$v.so timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/w_$v -o w -- python $R/tools/attn_bench.py --batch 4 --particles 4096 --stash 1 > $R/$O/w_$v.log 2>&1; done
cd $R; for v in AQN AQL; do find $O/w_$v -mindepth 2 -type f -exec mv {} $O/w_$v/ \; 2>/dev/null; python - <<PY
import csv
rows=[r for r in csv.DictReader(open('$O/w_$v/w_counter_collection.csv')) if 'attn_bwd' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE']
vals=[float(r['Counter_Value']) for r in rows]
print('$v', 'attn_bwd launches', len(vals), 'WRITE_SIZE mean (KB units as reported)', sum(vals)/max(1,len(vals)))
PY
done | tee $O/write_size.txt
