"""Kernel-coverage record of the GPU test suite.

  on the GPU box (tools/runs/r06a.sh):
      DIB_COVERAGE_LOG=/tmp/cov/tests.tsv rocprofv3 --kernel-trace --marker-trace -M -f csv -d /tmp/cov/trace -- \
          python -m pytest tests -m gpu -q
      python tools/kernel_coverage.py build /tmp/cov/trace /tmp/cov/tests.tsv gpurun_out/r06_suite_kernel_coverage.txt
  anywhere (tests/test_kernel_coverage.py, no GPU):
      python tools/kernel_coverage.py check profiles/r06_suite_kernel_coverage.txt

`build` assigns every kernel dispatch of every traced process (pytest itself and the subprocesses its tests start) to the test
whose ROCTx range (tests/conftest.py) contains the dispatch's start timestamp, and writes, for every kernel symbol of
libdib_hip.so's gfx950 code object, the number of launches and the tests that launched it ('*' = the test compares with the
float64 / NumPy oracle, see ORACLE_TESTS).  `check` = every kernel symbol of the library as built here is on that list with at
least one test.
"""
import bisect
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "distributed-information-bottleneck.github.io_amd", "libdib_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# tests (file::function) whose assertions compare device results with an INDEPENDENT CPU checker: oracle/dib_oracle.py (float64
# NumPy), oracle/dib_torch_cpu.py (float64 autograd), the float64 loop / set-transformer oracles, plain NumPy float64 formulas
ORACLE_TESTS = {
    "test_gpu_parity.py": {
        "test_gemm_vs_numpy", "test_skinny_k_gemm_vs_numpy", "test_eps_matches_oracle_and_host_ref", "test_forward_backward_parity",
        "test_late_annealing_regime_small_sigma_large_mu", "test_split_batch_wgrad_and_dp_equivalence", "test_adam_matches_keras_form",
        "test_encode_deterministic_and_bhattacharyya", "test_fit_trajectory_matches_oracle_fit", "test_north_star_shape_properties",
        "test_mi_sandwich_bounds_match_oracle", "test_ib_flag_one_wide_feature_on_the_boolean_circuit",
        "test_infonce_loss_and_grads_match_oracle", "test_infonce_at_working_batch_sizes", "test_infonce_edge_shapes",
        "test_infonce_one_launch_path", "test_mi_sandwich_bounds_at_the_reference_evaluation_size", "test_dense_stack_matches_numpy",
        "test_dense_stack_row_tile_kernels", "test_autograd_bridge_matches_oracle_gradients",
        "test_hipgraph_fit_is_bit_identical_to_eager_and_matches_oracle", "test_north_star_architecture_multi_step_trajectory",
        "test_random_architectures_forward_backward"},
    "test_gpu_fullsize.py": {
        "test_config3_full_batch_step_all_gradients", "test_config3_full_batch_forward_against_an_independent_float64_forward", "test_config4_f50_full_batch_step_all_gradients",
        "test_config4_f50_ragged_batch_general_and_fused_paths_agree", "test_config3_fit_trajectory_beta_ramp_full_batch"},
    "test_gpu_set_transformer.py": {
        "test_forward_replays_the_notebook_fixture", "test_forward_backward_parity", "test_config5_size_4096_particles_flash_all_gradients",
        "test_config5_full_depth_six_blocks_at_4096_particles", "test_attention_backward_score_stash_equals_recompute",
        "test_train_steps_match_oracle_adam", "test_probe_grid_information_bounds_match_oracle",
        "test_backward_follows_the_forward_that_ran"},
    "test_gpu_trajectories.py": {
        "test_infonce_loop_trajectory_matches_float64_oracle", "test_keras_path_trajectory_160_steps_through_the_ramp"},
    "test_gpu_building_blocks.py": {
        "test_softmax_rows_forward_backward_vs_float64", "test_add_layernorm_forward_backward_vs_float64", "test_act_grad_mul_vs_numpy",
        "test_sgd_step_vs_numpy", "test_gemm_tile_shapes_the_default_rules_rarely_pick", "test_weight_gradient_flat_tile_vs_numpy", "test_attention_forward_with_projections_vs_float64", "test_attention_backward_with_projection_gradient_vs_float64"},
}
# tests that demand the bits (or fp32 summation-order tolerance) of a path the tests above check against an oracle
EQUIVALENCE_TESTS = {
    "test_gpu_parity.py": {
        "test_companion_grids_equal_the_separate_launches", "test_fused_output_head_equals_the_unfused_sequence",
        "test_step_tail_equals_the_separate_launches", "test_small_batch_row_tile_kernels_equal_the_large_batch_path",
        "test_workspace_needs_only_dib_workspace_init", "test_inference_forward_skips_stashes_but_not_results"},
    "test_gpu_set_transformer.py": {
        "test_attention_backward_8_waves_equals_4_waves", "test_token_chain_kernels_equal_the_layer_by_layer_path",
        "test_deferred_grouped_weight_gradients_equal_the_per_block_launches",
        "test_graph_replay_of_the_train_step_is_bit_identical_to_eager",
        "test_evaluation_forward_skips_the_score_stash_and_backward_still_agrees",
        "test_data_parallel_shards_reproduce_the_full_batch_gradient", "test_large_token_count_uses_split_weight_gradients"},
    "test_gpu_dp_and_cache.py": {
        "test_fit_under_rccl_one_rank_equals_single_process", "test_three_bucket_backward_hooks_equal_the_plain_backward",
        "test_set_transformer_train_step_under_rccl_one_rank_equals_single_process"},
    "test_gpu_concurrency.py": {"test_two_threads_two_streams_equal_the_serial_run"},
}


def _kind(nodeid):
    """'*' oracle-comparing, '=' equivalence with an oracle-checked path, ' ' other"""
    parts = nodeid.split("::")
    f, fn = os.path.basename(parts[0]), parts[-1].split("[")[0]
    if fn in ORACLE_TESTS.get(f, ()):
        return "*"
    if fn in EQUIVALENCE_TESTS.get(f, ()):
        return "="
    return " "


def library_kernels(lib=LIB):
    """Mangled names of the kernels in the library's gfx950 code object (the .kd kernel-descriptor symbols)."""
    with tempfile.TemporaryDirectory() as d:
        fb, co = os.path.join(d, "fatbin"), os.path.join(d, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fb}", lib, os.path.join(d, "x")], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fb}", f"--output={co}"], check=True)
        out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], check=True, capture_output=True, text=True).stdout
    names = set()
    for line in out.splitlines():
        parts = line.split()
        if parts and parts[-1].endswith(".kd"):
            names.add(parts[-1][:-3])
    return sorted(names)


def demangle(names):
    import shutil
    cxxfilt = next((c for c in (os.path.join(LLVM, "llvm-cxxfilt"), shutil.which("c++filt")) if c and os.path.exists(c)), None)
    if cxxfilt is None:
        return {n: n for n in names}
    out = subprocess.run([cxxfilt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def _read_ranges(trace_dir, tests_tsv):
    """[(start, end, nodeid)] in the profiler's clock: ROCTx ranges if the marker trace has them, else the host-clock log."""
    ranges = []
    for path in glob.glob(os.path.join(trace_dir, "**", "*marker_api_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = next((v for v in row.values() if isinstance(v, str) and v.startswith("dibtest::")), None)
                if name:
                    ranges.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name[len("dibtest::"):]))
    if ranges:
        return sorted(ranges), "ROCTx ranges (rocprofv3 --marker-trace)"
    return None, None


def _ranges_from_clock_log(tests_tsv, kstart_min, kstart_max):
    rows = [l.rstrip("\n").split("\t") for l in open(tests_tsv)]
    best = None
    for c, label in enumerate(("CLOCK_MONOTONIC", "CLOCK_BOOTTIME", "CLOCK_REALTIME")):
        r = sorted((int(x[1 + c]), int(x[4 + c]), x[0]) for x in rows)
        if r and r[0][0] <= kstart_min and kstart_max <= r[-1][1]:
            best = (r, f"host clock log ({label})")
            break
    if best is None:
        raise SystemExit("no clock of the host log brackets the kernel timestamps and the marker trace is empty")
    return best


def build(trace_dir, tests_tsv, out_path):
    disp = []   # (start, name)
    for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if name.endswith(".kd"):
                    name = name[:-3]
                disp.append((int(row["Start_Timestamp"]), name))
    disp.sort()
    ranges, how = _read_ranges(trace_dir, tests_tsv)
    if ranges is None:
        ranges, how = _ranges_from_clock_log(tests_tsv, disp[0][0], disp[-1][0])
    starts = [r[0] for r in ranges]
    by_kernel = {}
    unassigned = 0
    for t, name in disp:
        i = bisect.bisect_right(starts, t) - 1
        while i >= 0 and ranges[i][1] < t:   # ranges of one pytest process do not nest; step back over finished ones
            i -= 1
            if i < 0 or ranges[i][0] + 10**12 < t:
                i = -1
                break
        if i < 0:
            unassigned += 1
            continue
        d = by_kernel.setdefault(name, {})
        d[ranges[i][2]] = d.get(ranges[i][2], 0) + 1
    lib_kernels = library_kernels()
    dm = demangle(lib_kernels)
    lines = ["# kernel coverage of `python -m pytest tests -m gpu` under rocprofv3 --kernel-trace --marker-trace -M",
             f"# test ranges from: {how}; {len(disp)} dispatches, {unassigned} outside every test range (session set-up)",
             f"# {len(lib_kernels)} kernel symbols in libdib_hip.so (gfx950 code object); '*' = the test compares with the float64 / NumPy oracle,",
             "# '=' = the test demands equality (bits or fp32 summation-order tolerance) with a path that '*' tests check",
             "# format: KERNEL <mangled>  |  <demangled>  |  launches  |  tests  |  '*' tests  |  '=' tests ; then one line per test (at most 12, '*' first)"]
    missing = []
    for k in lib_kernels:
        tests = by_kernel.get(k, {})
        oracle = [t for t in tests if _kind(t) == "*"]
        equiv = [t for t in tests if _kind(t) == "="]
        lines.append(f"KERNEL {k}  |  {dm[k]}  |  {sum(tests.values())}  |  {len(tests)}  |  {len(oracle)}  |  {len(equiv)}")
        if not tests:
            missing.append(k)
        order = sorted(tests, key=lambda t: ("*= ".index(_kind(t)), t))
        for t in order[:12]:
            lines.append(f"    {_kind(t)} {t}  ({tests[t]})")
        if len(order) > 12:
            lines.append(f"      ... and {len(order) - 12} more tests")
    other = sorted(k for k in by_kernel if k not in set(lib_kernels))
    lines.append(f"# kernels launched by the suite that are not the library's (torch / RCCL): {len(other)}")
    lines.append(f"# library kernels no test launched: {len(missing)}")
    for k in missing:
        lines.append(f"MISSING {k}  |  {dm[k]}")
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{out_path}: {len(lib_kernels)} kernels, {len(missing)} not launched, {unassigned} dispatches unassigned")


def read_record(path):
    """{mangled kernel: (launches, number of tests, number of oracle-comparing tests, number of equivalence tests)}"""
    rec = {}
    for line in open(path):
        if line.startswith("KERNEL "):
            parts = [p.strip() for p in line[len("KERNEL "):].split("  |  ")]
            rec[parts[0]] = (int(parts[2]), int(parts[3]), int(parts[4]), int(parts[5]) if len(parts) > 5 else 0)
    return rec


def check(path):
    rec = read_record(path)
    bad = [k for k in library_kernels() if rec.get(k, (0, 0, 0, 0))[1] == 0]
    for k in bad:
        print("not covered:", k)
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(*sys.argv[2:5])
    elif sys.argv[1] == "check":
        sys.exit(check(sys.argv[2]))
    elif sys.argv[1] == "list":
        for k, v in demangle(library_kernels()).items():
            print(k, " | ", v)
