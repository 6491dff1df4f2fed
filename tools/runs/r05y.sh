#!/bin/bash
# r05y: the row-tile regime as (row tiles x features) <= "small_wgs" = 512, up to 2048 rows (was: batch <= 1024): equivalence /
# parity tests incl. B = 2048, the small-batch bench functions, crossover spot checks
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05y; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "small_batch or dense or companion or workspace_needs or step_tail or tuning or custom or north_star or infonce_training_loop" ) > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
timeout 300 python -c "
import json, bench
k = bench.keras_path_default_batch('cuda:0')
print('keras pair us', k['us_per_train_plus_validation_step'], 'one-by-one', k['validation_batches_one_by_one']['us_per_train_plus_validation_step'])
for b in (128, 1024, 2048):
    print('infonce loop', json.dumps(bench.config2_infonce_loop('cuda:0', b)))
import sys; sys.path.insert(0, 'tools')
import small_batch_crossover as c
for F, B in ((10, 1024), (64, 1024), (64, 256), (4, 2048)):
    print('crossover F', F, 'B', B, 'default rule us', round(c.step_us(F, B, 1), 1), 'large us', round(c.step_us(F, B, 0), 1))
" 2>&1 | grep -v "amdgpu.ids\|^F  \|^[0-9]" | tee $O/bench.txt
