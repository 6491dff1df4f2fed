#!/bin/bash
# round-4 GPU call A: new parity tests (>=100-step trajectories of the InfoNCE loop and the Keras path, late-annealing regime,
# lazy score stash, config 5 at full depth), config-4 (F = 50) rocprofv3 passes, full bench line (fit_surface, config-4 table)
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_trajectories.py "tests/test_gpu_parity.py::test_late_annealing_regime_small_sigma_large_mu" \
    "tests/test_gpu_parity.py::test_infonce_training_loop_on_pendulum" \
    "tests/test_gpu_set_transformer.py::test_score_stash_is_allocated_lazily_and_capped_across_plans" \
    "tests/test_gpu_set_transformer.py::test_evaluation_forward_skips_the_score_stash_and_backward_still_agrees" \
    "tests/test_gpu_set_transformer.py::test_step_plans_are_lru_capped_and_graph_plans_pinned" \
    "tests/test_gpu_set_transformer.py::test_config5_full_depth_six_blocks_at_4096_particles" -q -s --durations=8 ) > $O/tests.log 2>&1
tail -n 40 $O/tests.log
bash tools/collect_profiles.sh $O/c4 --features 50 > $O/collect.log 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04a/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["frac"])
e=d["extra"]
print(json.dumps(e.get("fit_surface")))
print(json.dumps(e.get("config4_F50")))
print(json.dumps(e.get("config2_infonce_loop")))
PY
