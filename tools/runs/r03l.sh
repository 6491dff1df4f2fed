#!/bin/bash
# r03l: forward tile rule - 64x64 tiles below DIB_FWD_NARROW_WGS 64x128 workgroups (512 = round-2 rule): set transformer at
# small sizes and the encoder step at mid-size batches; the new backward-consistency test
O=gpurun_out/r03l; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x -k "backward_follows" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
for n in 512 1024 2048 4096; do
  for a in "--batch 32 --particles 50 --steps 50 --warmup 5" "--batch 4 --particles 512 --steps 20" "--batch 2 --particles 2048 --steps 10"; do
    echo "NARROW=$n $(DIB_FWD_NARROW_WGS=$n python tools/set_transformer_bench.py $a 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["workload"][20:60], d["ms_per_step"])')"
  done
  for b in 4096 16384 32768; do
    echo "NARROW=$n B=$b $(DIB_FWD_NARROW_WGS=$n python bench.py --steps 20 --warmup 3 --blocks 3 --no-cpu-baseline --no-extra --no-kernel-timing --batch $b 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["timing"]["blocks_ms_per_step"])')"
  done
done 2>&1 | tee $O/narrow_rule.txt
