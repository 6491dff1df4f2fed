#!/bin/bash
# r03ab: attention backward with the query-tile images double-buffered in LDS (next tile staged between the MFMAs of the dQ
# product).  AB0 = single buffer (round state), AD8 / AD10 / AD12 = double buffer, Q image written in dQ step 8 / 10 / 12,
# ADD = AD10 + dQ read-back deferred into the next tile's dP product.  Parity subset on the product (= AD10) and on ADD.
O=gpurun_out/r03ab; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "not config5_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
(DIB_LIB_PATH=exp/lib_ADD.so timeout 600 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "score_stash or forward_backward_parity" > $O/pytest_ADD.log 2>&1; echo "rc=$?" >> $O/pytest_ADD.log); tail -2 $O/pytest_ADD.log
for rep in 1 2 3; do for v in AB0 AD8 AD10 AD12 ADD; do echo "$v $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash 1 2>/dev/null)"; done; done | tee $O/attn_ab.txt
for v in AB0 AD10; do echo "$v recompute $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash 0 2>/dev/null)"; done | tee -a $O/attn_ab.txt
