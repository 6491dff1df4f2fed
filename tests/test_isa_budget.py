"""Resource audit of the generated gfx950 code (no GPU needed: hipcc cross-compiles to assembly).

The performance of the hot kernels rests on properties the compiler decides and a harmless-looking source edit can lose
(DESIGN.md 4.4 / 9 list how each was found): no stack objects or spills in the attention kernels (a `float4 (&)[4]` helper once
left a prefetched tile in scratch: every global load was followed by a wait + scratch store), the attention backward's
loop-carried accumulators resident in AGPRs (a divergent `if` around the MFMAs made the compiler copy all 128 of them into
AGPRs and back on every query tile: 441 + 264 `v_accvgpr_*` per tile instead of the zero-initialisations only), register
budgets that keep 2 waves per SIMD where the design assumes them, and the MFMA count of each tile body (= the algorithmic
FLOPs; a changed count means the arithmetic changed).  The thresholds are the current values plus a little slack."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "distributed-information-bottleneck.github.io_amd", "csrc", "dib_api.hip")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "dib_api.s")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", SRC, "-o", out],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = open(out).read()
    info = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        if end < 0:
            continue
        body, tail = text[m.end():end], text[end:end + 4000]
        meta = {k: int(v) for k, v in re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", tail)}
        if "NumVgprs" not in meta:
            continue
        meta["mfma"] = len(re.findall(r"^\s*v_mfma", body, re.M))
        meta["accvgpr_write"] = len(re.findall(r"v_accvgpr_write", body))
        meta["accvgpr_read"] = len(re.findall(r"v_accvgpr_read", body))
        info[m.group(1)] = meta
    return info


def _one(kernels, *needles):
    hits = [k for k in kernels if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, hits)
    return kernels[hits[0]]


def test_attention_forward_two_waves_per_simd_no_scratch(kernels):
    k = _one(kernels, "dib_attn_fwd_kernel")
    assert k["ScratchSize"] == 0 and k["NumAgprs"] == 0
    assert k["NumVgprs"] <= 256 and k["Occupancy"] == 2
    assert k["mfma"] == 128               # 64 (S^T = K Q^T) + 64 (O^T += V^T P^T) per 32-key tile


def test_attention_backward_accumulators_stay_in_agprs(kernels):
    ks = _one(kernels, "dib_attn_bwd_kernelILb1E")   # stash mode: scores read back from the forward's stash
    assert ks["ScratchSize"] == 0 and ks["mfma"] == 256             # dP (64) + dV, dK (128) + dQ (64): the algorithmic four
    assert ks["accvgpr_write"] <= 320 and ks["accvgpr_read"] <= 120 and ks["NumVgprs"] + ks["NumAgprs"] <= 512
    k = _one(kernels, "dib_attn_bwd_kernelILb0E")    # recompute mode
    assert k["ScratchSize"] == 0
    assert k["mfma"] == 320               # S, dP (128) + dV, dK (128) + dQ (64) per 32-query tile
    # zero-initialisations of the accumulators are the only v_accvgpr_write in the kernel (128 dV/dK once + 96 per tile + the
    # zero-trip copy), the reads are the S / dP / dQ tiles leaving the matrix pipe: no per-tile shuffling of dV / dK
    assert k["accvgpr_write"] <= 300, k
    assert k["accvgpr_read"] <= 100, k
    assert k["NumVgprs"] + k["NumAgprs"] <= 512


def test_fused_encoder_kernels_keep_two_waves_per_simd(kernels):
    fwd = _one(kernels, "dib_fused_encoder_fwd_kernelILi128ELi128ELi32ELb1")
    bwd = _one(kernels, "dib_fused_encoder_bwd_kernelILi128ELi128ELi32ELb1")
    assert fwd["NumVgprs"] <= 256 and fwd["Occupancy"] == 2 and fwd["ScratchSize"] == 0 and fwd["NumAgprs"] == 0
    assert fwd["mfma"] == 416             # 32 (layer 1) + 256 (layer 2) + 128 (layer 3) per 32-sample tile
    assert bwd["NumVgprs"] <= 256 and bwd["Occupancy"] == 2 and bwd["NumAgprs"] == 0
    assert bwd["ScratchSize"] <= 32       # 2-3 spilled registers today
    assert bwd["mfma"] == 448             # 128 + 256 (dgrads) + 64 (layer-1 weight gradient, 16x16x4); act'(h1) is a stashed bit mask


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_big_gemm_tiles_fit_two_workgroups_per_cu(kernels, mode):
    k = _one(kernels, f"dib_gemm_kernelILi{mode}ELi2ELi2ELi64E")
    assert k["ScratchSize"] == 0 and k["NumAgprs"] == 0
    assert k["NumVgprs"] <= 256 and k["Occupancy"] >= 2
    assert k["LDSByteSize"] <= 80 * 1024  # two workgroups per CU (160 KB of LDS)
    assert k["mfma"] == 128               # one 64-deep K tile of the 128 x 128 output tile per wave


def test_no_kernel_of_the_library_uses_scratch_except_the_fused_backward(kernels):
    """83 kernels; the only private-segment use is the 128/128/32 fused backward's 2-3 spilled registers (<= 32 bytes).  A new
    entry here is either a spill or - worse - an array that was not promoted to registers (see the module docstring)."""
    assert len(kernels) >= 80
    offenders = {k: v["ScratchSize"] for k, v in kernels.items() if v["ScratchSize"] > 0}
    assert all("dib_fused_encoder_bwd_kernelILi128ELi128ELi32E" in k and v <= 32 for k, v in offenders.items()), offenders


def test_round5_small_batch_kernels_keep_two_waves_per_simd(kernels):
    """The kernels round 5 added for small batches / <= 64 particles run ONE workgroup per CU and rest on its 8 waves
    being two per SIMD (profiles/HISTORY.md 13): 512-thread workgroups need <= 256 registers per wave.  The paired
    integration kernel (argument set picked by blockIdx.y from the kernarg segment) must cost what the single one costs."""
    single = _one(kernels, "dib_small_integration_kernel")
    pair = _one(kernels, "dib_small_integration_pair_kernel")
    assert pair["NumVgprs"] == single["NumVgprs"] and pair["ScratchSize"] == 0 and pair["mfma"] == single["mfma"]
    for kind in (0, 1, 4):
        k = _one(kernels, f"dib_infonce_small_kernelILi{kind}E")
        assert k["NumVgprs"] + k["NumAgprs"] <= 128 and k["ScratchSize"] == 0   # (<= 128: four waves per SIMD would fit too)
    b8, b4 = _one(kernels, "dib_attn_small_bwd8_kernelILb0E"), _one(kernels, "dib_attn_small_bwd_kernel")
    assert b8["NumVgprs"] + b8["NumAgprs"] <= 256 and b8["ScratchSize"] == 0
    assert b4["mfma"] == 320 and b8["mfma"] == 64 + 64 + 64 + 32   # S | dP (one per wave group), dV | dK (shared code), dQ tile


def test_round6_kernels_keep_their_budgets(kernels):
    """Round 6: the <= 64-particle attention kernels with MultiHeadAttention's input projections inside (forward: + 96 MFMAs for
    the head's q, k, v with all 96 weight values in flight - still two workgroups per CU; backward: + 48 MFMAs per wave for the
    projections' input gradient, two waves per SIMD), and the 32 x 256 weight-gradient tile."""
    f0, f1 = _one(kernels, "dib_attn_small_fwd_kernelILb0E"), _one(kernels, "dib_attn_small_fwd_kernelILb1E")
    assert f0["ScratchSize"] == 0 and f1["ScratchSize"] == 0 and f0["mfma"] == 128 and f1["mfma"] == 128 + 96
    assert f1["NumVgprs"] + f1["NumAgprs"] <= 256 and f1["Occupancy"] >= 2
    b8p = _one(kernels, "dib_attn_small_bwd8_kernelILb1E")
    assert b8p["ScratchSize"] == 0 and b8p["NumVgprs"] + b8p["NumAgprs"] <= 256 and b8p["mfma"] == 64 + 64 + 64 + 32 + 48
    flat = _one(kernels, "dib_gemm_kernelILi2ELi1ELi2ELi32ELb1E")
    assert flat["ScratchSize"] == 0 and flat["Occupancy"] >= 2 and flat["LDSByteSize"] <= 40 * 1024 and flat["mfma"] == 32
    single = _one(kernels, "dib_small_integration_kernel")
    assert single["ScratchSize"] == 0 and single["NumVgprs"] <= 256       # + the head-step reduce (dib_mlp_small_head_step)
    # cluster mode: the slice primitives keep a share's whole batch of weight loads in registers (16 float4 a lane at most) - no
    # scratch, two waves per SIMD
    fwd8 = _one(kernels, "dib_attn_fwd8_kernel")   # the 8-wave flash forward: the 4-wave kernel's wave code, two waves per SIMD
    assert fwd8["ScratchSize"] == 0 and fwd8["NumVgprs"] + fwd8["NumAgprs"] <= 256 and fwd8["mfma"] == 128
    cluster = _one(kernels, "dib_small_integration_cluster_kernel")
    assert cluster["ScratchSize"] == 0 and cluster["NumVgprs"] + cluster["NumAgprs"] <= 256
