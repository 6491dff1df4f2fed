// dib_fused.h - fused encoder-bank kernels for the common architecture class (two hidden layers of
// widths that are multiples of 32, E in {8,16} or a multiple of 32, encoder input width <= 16).
//
// FORWARD  (reference models.py:101-112 in ONE launch): tf.split + PositionalEncoding + Dense(H1,act) +
// Dense(H2,act) + Dense(2E) + split(mu,logvar) + reparameterised sample + per-feature KL.
//
// Design (MI355X first, not a GEMM-library call sequence):
//   * one workgroup = one feature, persistent over batch tiles; the feature's three weight matrices
//     (100.9 KB fp32 for 5->128->128->64) are staged ONCE into LDS (160 KB/CU) and stay there;
//   * each wave owns 32 samples and keeps its activations in REGISTERS for the whole chain.  The
//     layers are evaluated as the transposed product  H_out^T[n, m] = sum_k W[k, n] * H_in^T[k, m]  with
//     v_mfma_f32_32x32x2_f32: lane (m = lane&31, h = lane>>5) ends up holding
//     H_out[m][32*j + (r&3) + 8*(r>>2) + 4*h] in accumulator register r of tile j - which is exactly the
//     B-operand layout the next layer's MFMAs need (contraction index k(j,r,h) = 32j+(r&3)+8(r>>2)+4h),
//     so activations never touch LDS or HBM between layers; only weights are read from LDS
//     (transposed image Wt[n][k], pitch K+4: one conflict-free ds_read_b128 feeds four MFMAs);
//   * mu/logvar of one (sample, dim) land in the same lane, so reparameterisation, the Philox noise
//     (keyed by global row/feature/dim/step, regenerated in backward) and the KL reduction are lane-local
//     epilogue work; KL partial sums are accumulated per wave in a fixed order (deterministic);
//   * h1, h2, (mu|logvar) are stashed feature-major for the backward pass (HBM writes overlap the MFMAs;
//     recomputing them instead would add 25% MFMA work to an MFMA-bound step).
// Exact fp32 throughout (f32-input MFMA = fmaf chain).
#pragma once
#include "dib_common.h"
#include "dib_gemm.h"

struct DibFusedFwdArgs {
  const float* P;           // positional-encoded inputs, feature-major ragged [F][B][in_dim_f] (dib_posenc_kernel)
  const int* row_idx; long long row0; int batch;
  const float* params;
  const long long* w_off;   // [3][F] kernel offsets (layer-major), from the layout
  const long long* b_off;   // [3][F] bias offsets
  const int4* featmap;      // [F] {d_f, in_dim_f, x column offset, sum of in_dim of earlier features}
  int n_blocks;             // 1 + number of sinusoids
  int act;
  float* h1; float* h2; float* enc_out; float* U; float* kl_partial;  // kl_partial[gridDim.x*8][F]
  unsigned long long* h2mask;  // [F][B][2] sign bits of h2 in fragment order (bit 16*tile + reg), for the fused backward
  unsigned long long* h1mask;  // same for h1 (NULL: no fused backward follows)
  int F; unsigned long long seed; unsigned step; int deterministic;
  const unsigned* step_dev;  // if non-NULL the noise step is read from device memory (hipGraph replay)
};

template <int H1, int H2, int E>
struct DibFusedCfg {
  static constexpr int E2 = 2 * E;
  static constexpr int N3 = (E2 + 31) / 32 * 32;      // layer-3 output rows padded to a multiple of 32
  static constexpr int K1P = 36;                      // layer-1 image pitch: K padded to 32 (+4)
  static constexpr int P2 = H1 + 4, P3 = H2 + 4;
  static constexpr int W1_FLOATS = H1 * K1P, W2_FLOATS = H2 * P2, W3_FLOATS = N3 * P3;
  static constexpr int B_FLOATS = H1 + H2 + N3;
  static constexpr int PATCH = 32 * 36;               // per-wave 32x32 transpose patch (pitch 36)
  static constexpr int LDS_FLOATS = W1_FLOATS + W2_FLOATS + W3_FLOATS + B_FLOATS + 8 * PATCH;
  static constexpr int T1 = H1 / 32, T2 = H2 / 32, T3 = N3 / 32;
};

typedef float dib_nt4 __attribute__((ext_vector_type(4)));  // native vector type for non-temporal 16-byte stores

// C-fragment row of accumulator register r for lane-half h:  (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ int dib_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// sigma = exp(logvar / 2) (reference models.py:108)
__device__ __forceinline__ float dib_sigma(float lv) {
  return __builtin_amdgcn_exp2f(0.72134752044448170f * lv);  // v_exp_f32: 2^(lv * log2(e)/2); exp(lv) = sigma^2
}

// Piecewise-linear activations only on the fused path (linear / relu / leaky_relu): act(v) = max(v,0) + slope*min(v,0)
// is branch-free, so the hot loop stays straight-line code.  Other activations use the general GEMM path.
// slope of the negative branch: relu 0, Keras 'leaky_relu' 0.2, tf.keras.layers.LeakyReLU(0.1) (DIB_ACT_LEAKY_RELU_01 = 7) 0.1, linear 1
__device__ __forceinline__ float dib_neg_slope(int act) { return act == 1 ? 0.f : (act == 2 ? 0.2f : (act == 7 ? 0.1f : 1.f)); }
// RELU (the reference default, train.py:37) is a compile-time specialisation: one v_max per element instead of three ops.
template <bool RELU>
__device__ __forceinline__ void dib_act_tile(float slope, dib_f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = RELU ? fmaxf(v[r], 0.f) : fmaxf(v[r], 0.f) + slope * fminf(v[r], 0.f);
}

// Row of the 32x36 patch that `lane` touches in pass pss of a row-major sweep (8 lanes x 16 B per row).  Each 16-lane
// group gets rows r and r+8: 8 rows = 288 floats = 32 banks apart, so the two 128-byte row segments of a group fall
// on disjoint halves of the 64 LDS banks (rows r, r+1 - 36 floats apart - collide on 4 banks).
__device__ __forceinline__ int dib_patch_row(int lane, int pss) {
  return (lane >> 4) + 8 * ((lane >> 3) & 1) + 4 * (pss & 1) + 16 * (pss >> 1);
}

// (row, first column) of the patch that `lane` re-reads in pass pss of dib_store_tile.  ds_read_b128 is serviced in the
// HARDWARE lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32 for the upper half) - MI355X_MICROARCH.md, LDS table -
// not in contiguous 16-lane groups, so the round-1 map (dib_patch_row: contiguous 8-lane runs per row) put rows r, r+1 and
// r+8, r+9 into one service group and conflicted 2-way (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42 in the fused
// forward, whose only LDS traffic besides the conflict-free weight reads is this transpose).  Here service group G
// (0..3) reads exactly rows r = G (+4, +16 by pass) and r + 8, eight lanes per row: 2 x 32 consecutive dwords whose bank
// offsets differ by 8*36 mod 64 = 32 - all 64 banks once.  The same lanes then store full 128-byte row segments
// (aligned lane quads write 64 contiguous bytes).
__device__ __forceinline__ int dib_store_idx(int lane) {  // index 0..15 of the lane inside its ds_read_b128 service group
  return 4 * (((lane & 31) >> 2) >> 1) + (lane & 3);
}
__device__ __forceinline__ int dib_store_col(int lane) { return (dib_store_idx(lane) & 7) * 4; }
__device__ __forceinline__ int dib_store_row(int lane, int pss) {
  const int q = (lane & 31) >> 2;
  const int G = 2 * (lane >> 5) + ((0x96 >> q) & 1);
  return G + 4 * (pss & 1) + 16 * (pss >> 1) + 8 * (dib_store_idx(lane) >> 3);
}

// Write one 32(samples) x 32(units) tile held as a transposed-product C fragment to row-major global memory
// with full 128-byte lines: C fragment -> wave-private LDS patch T[m][n] (4 ds_write_b128) -> each lane re-reads 4
// consecutive units of one sample row (ds_read_b128) -> 16-byte global stores, 8 lanes per 128-byte row segment.
// dst points at (sample row 0 of the wave, first unit of the tile); ld = row pitch in floats;
// rows_valid = number of the wave's 32 sample rows that exist.
__device__ __forceinline__ void dib_store_tile(float* __restrict__ patch, const dib_f32x16& c, float* __restrict__ dst,
                                               long long ld, int rows_valid, int lane) {
  const int m = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(patch + m * 36 + 8 * g + 4 * h) = make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
  __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; keep the compiler from reordering them
  // pass pss reads row r0 + {0, 4, 16, 20}[pss]: one live row index, constant offsets fold into the addresses
  const int cc = dib_store_col(lane);
  const int r0 = dib_store_row(lane, 0);
  const float* src = patch + r0 * 36 + cc;
  float4 v[4];
#pragma unroll
  for (int pss = 0; pss < 4; ++pss) v[pss] = *reinterpret_cast<const float4*>(src + (4 * (pss & 1) + 16 * (pss >> 1)) * 36);
  __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; keep the compiler from reordering them
  float* out = dst + (long long)r0 * ld + cc;
  if (rows_valid >= 32) {  // wave-uniform fast path: four unconditional 16-byte stores
#pragma unroll
    for (int pss = 0; pss < 4; ++pss)
      __builtin_nontemporal_store(dib_nt4{v[pss].x, v[pss].y, v[pss].z, v[pss].w},
                                  reinterpret_cast<dib_nt4*>(out + (long long)(4 * (pss & 1) + 16 * (pss >> 1)) * ld));
  } else {
#pragma unroll
    for (int pss = 0; pss < 4; ++pss)
      if (r0 + 4 * (pss & 1) + 16 * (pss >> 1) < rows_valid)
        *reinterpret_cast<float4*>(out + (long long)(4 * (pss & 1) + 16 * (pss >> 1)) * ld) = v[pss];
  }
}

// Stage COUNT elements global -> LDS with 512 threads: element i is produced by load(i) and consumed by store(i, v);
// 8 loads are in flight per thread before the first store of a batch.
template <int COUNT, typename LoadF, typename StoreF>
__device__ __forceinline__ void dib_stage_batched(int tid, LoadF load, StoreF store) {
  constexpr int ITERS = (COUNT + 511) / 512;
#pragma unroll
  for (int it0 = 0; it0 < ITERS; it0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + (it0 + u) * 512;
      v[u] = (it0 + u < ITERS && i < COUNT) ? load(i) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + (it0 + u) * 512;
      if (it0 + u < ITERS && i < COUNT) store(i, v[u]);
    }
  }
}

// Phase timing of the fused forward (diagnostic build -DDIB_FUSED_TIMING; tools/fused_phase_timing.py): wave 0 of workgroup
// (0, 0) accumulates s_memtime deltas per phase of the tile loop into dib_fused_dbg.  (The per-workgroup / per-wave timeline
// marks of round 2 - profiles/r02w_*, r02ag_* - are in the git history.)
#ifdef DIB_FUSED_TIMING
__device__ long long dib_fused_dbg[16];
#define DIB_FT(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); tacc_[i] += now_ - tprev_; tprev_ = now_; \
                       __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIB_FT(i) do { } while (0)
#endif

// act'(tile) as one bit per unit, fragment order (bit 16*tile + reg), for T tiles of a lane (T <= 4: one 64-bit word).
// v > 0  <=>  its bit pattern is a positive integer (+0 -> 0, negatives and -0 -> negative): v_med3_i32 clamps it to {0,1} and
// v_lshl_or_b32 shifts it in - 2 VALU ops per unit, no compare / SGPR round trip.  (Inline asm: the compiler otherwise
// canonicalises the clamp back into v_cmp + v_cndmask + v_or3 with one live constant register per bit, which cost 22
// spilled VGPRs and 0.1 ms.)
template <int T>
__device__ __forceinline__ unsigned long long dib_sign_bits(const dib_f32x16 (&t)[T]) {
  static_assert(T <= 4, "one 64-bit mask word per lane");
  unsigned int word[2] = {0u, 0u};
#pragma unroll
  for (int w = 0; w < (T + 1) / 2; ++w)
#pragma unroll
    for (int idx = (16 * T - 32 * w - 1 < 31 ? 16 * T - 32 * w - 1 : 31); idx >= 0; --idx) {  // high bit first: word = (word << 1) | on
      const int e = 32 * w + idx;
      int on;
      asm("v_med3_i32 %0, %1, 0, 1" : "=v"(on) : "v"(t[e >> 4][e & 15]));
      asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(word[w]) : "v"(word[w]), "v"(on));
    }
  return ((unsigned long long)word[1] << 32) | word[0];
}
// apply the stashed act' bits of tile jo to a gradient tile: keep = 0 or -1 via v_bfe_i32, then v_and (relu) / a select
template <bool RELU>
__device__ __forceinline__ void dib_mask_tile(dib_f32x16& acc, unsigned long long bits, int jo, float slope) {
  const unsigned int word = (unsigned int)(bits >> (32 * ((16 * jo) >> 5)));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int keep = __builtin_amdgcn_sbfe(word, (16 * jo + r) & 31, 1);
    acc[r] = RELU ? __int_as_float(__float_as_int(acc[r]) & keep) : acc[r] * (keep ? 1.f : slope);
  }
}

template <int H1, int H2, int E, bool RELU>
__global__ void __launch_bounds__(512)
dib_fused_encoder_fwd_kernel(DibFusedFwdArgs a) {
  using C = DibFusedCfg<H1, H2, E>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wt1 = lds;                       // [H1][K1P]   Wt1[n][k] = W1[k][n]
  float* Wt2 = Wt1 + C::W1_FLOATS;        // [H2][H1+4]
  float* Wt3 = Wt2 + C::W2_FLOATS;        // [N3][H2+4]  rows >= 2E are zero
  float* Bs = Wt3 + C::W3_FLOATS;         // b1 | b2 | b3(padded)
  float* patch = Bs + C::B_FLOATS + (threadIdx.x >> 6) * C::PATCH;  // wave-private transpose patch

  const int f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int4 fm = a.featmap[f];
  const int in_dim = fm.y;
  const int F = a.F;
  const float* Pf = a.P + (long long)fm.w * a.batch;  // this feature's dense [B][in_dim] block

  // ---- stage this feature's weights into LDS once (transposed images), zero padding included ----
  {
    const float* W1 = a.params + a.w_off[0 * F + f];
    const float* W2 = a.params + a.w_off[1 * F + f];
    const float* W3 = a.params + a.w_off[2 * F + f];
    const float* b1 = a.params + a.b_off[0 * F + f];
    const float* b2 = a.params + a.b_off[1 * F + f];
    const float* b3 = a.params + a.b_off[2 * F + f];
    // Global loads are issued in batches of 8 before their LDS stores so that 8 L2 round trips overlap (the naive
    // load -> wait -> store loop serialised ~56 of them per thread: ~20-30 us of fixed cost per launch, 10 % of the kernel
    // at an 8192-row batch).
    dib_stage_batched<H1 * 32>(tid, [&](int i) { const int k = i / H1; return (k < in_dim) ? W1[i] : 0.f; },
                               [&](int i, float v) { const int k = i / H1, n = i - k * H1; Wt1[n * C::K1P + k] = v; });
    dib_stage_batched<H1 * H2>(tid, [&](int i) { return W2[i]; },
                               [&](int i, float v) { const int k = i / H2, n = i - k * H2; Wt2[n * C::P2 + k] = v; });
    dib_stage_batched<H2 * C::N3>(tid, [&](int i) { const int k = i / C::N3, n = i - k * C::N3;
                                                    return (n < C::E2) ? W3[(long long)k * C::E2 + n] : 0.f; },
                                  [&](int i, float v) { const int k = i / C::N3, n = i - k * C::N3; Wt3[n * C::P3 + k] = v; });
    for (int i = tid; i < H1; i += 512) Bs[i] = b1[i];
    for (int i = tid; i < H2; i += 512) Bs[H1 + i] = b2[i];
    for (int i = tid; i < C::N3; i += 512) Bs[H1 + H2 + i] = (i < C::E2) ? b3[i] : 0.f;
  }
  __syncthreads();

  const int n_tiles = (a.batch + 255) / 256;
  const unsigned nstep = a.step_dev ? a.step_dev[0] : a.step;
  const float slope = dib_neg_slope(a.act);
  const int ksteps1 = 4 * ((in_dim + 7) / 8);  // layer-1 MFMA steps per output tile (k-blocks of 8 that hold data)
  float kl_acc = 0.f;

  // lane (m,h) supplies p[k] for k = (r&3) + 8*(r>>2) + 4*h as the layer-1 B operand (reference models.py:22-23
  // values, produced by dib_posenc_kernel).  Loads are branch-free (clamped row / column, masked value) and the next
  // tile's values are fetched while the current tile computes.
  auto load_p = [&](int tile, float (&dstp)[8]) {
    const int bb = min(tile * 256 + wave * 32 + m, a.batch - 1);
    const float* src = Pf + (long long)bb * in_dim;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = dib_crow(r, h);
      const float v = src[min(k, in_dim - 1)];
      dstp[r] = (k < in_dim) ? v : 0.f;
    }
  };
  float p[8], pn[8];
  if ((int)blockIdx.x < n_tiles) load_p(blockIdx.x, p);

#ifdef DIB_FUSED_TIMING
  long long tacc_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev_ = clock64();
  const long long tstart_ = tprev_;
  const long long wstart_ = wall_clock64();   // constant 100 MHz
  int ntl_ = 0;
#endif
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef DIB_FUSED_TIMING
    ++ntl_;
#endif
    const int wrow0 = tile * 256 + wave * 32;    // first local batch row of this wave
    const int rows_valid = min(32, a.batch - wrow0);
    const int b = wrow0 + m;                     // local batch row of this lane
    const bool valid = b < a.batch;
    const long long grow = a.row_idx ? (long long)a.row_idx[valid ? b : 0] : a.row0 + b;  // dataset row id
    {
      const int nt = tile + gridDim.x;
      load_p(nt < n_tiles ? nt : tile, pn);  // prefetch (harmless re-read on the last tile)
    }

    // ---- layer 1: h1^T = act(W1^T p^T + b1) ----
    dib_f32x16 h1[C::T1];
#pragma unroll
    for (int jo = 0; jo < C::T1; ++jo) {
      dib_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[32 * jo + dib_crow(r, h)];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (4 * g < ksteps1) {  // uniform
          const float4 w = *reinterpret_cast<const float4*>(Wt1 + (32 * jo + m) * C::K1P + 8 * g + 4 * h);
          acc = DIB_MFMA(w.x, p[4 * g + 0], acc);
          acc = DIB_MFMA(w.y, p[4 * g + 1], acc);
          acc = DIB_MFMA(w.z, p[4 * g + 2], acc);
          acc = DIB_MFMA(w.w, p[4 * g + 3], acc);
        }
      }
      dib_act_tile<RELU>(slope, acc);
      h1[jo] = acc;
    }
    DIB_FT(0);   // prefetch issue + layer 1
    // stash h1 (feature-major [F][B][H1]) for the layer-2 weight gradient (skipped for inference: DIB_FWD_INFERENCE) and
    // act'(h1) as bits for the fused backward
    if (a.h1 != nullptr) {
      float* dst = a.h1 + ((long long)f * a.batch + wrow0) * H1;
#pragma unroll
      for (int jo = 0; jo < C::T1; ++jo) dib_store_tile(patch, h1[jo], dst + 32 * jo, H1, rows_valid, lane);
    }
    if (a.h1mask != nullptr && valid) a.h1mask[((long long)f * a.batch + b) * 2 + h] = dib_sign_bits<C::T1>(h1);
    DIB_FT(1);   // stash h1
    // ---- layer 2: h2^T = act(W2^T h1^T + b2) ----
    dib_f32x16 h2[C::T2];
    if constexpr (C::T2 % 2 == 0) {
    // two output tiles at a time, MFMAs alternating between their accumulators (a filler between two MFMAs on the SAME
    // accumulator costs ~43 cycles, between independent ones only its issue slot), weight fragments of step s + 1 issued
    // before the 8 MFMAs of step s (sched_barrier + an empty asm pin: the instruction selector floats pure MFMAs across
    // sched_barrier)
#pragma unroll
    for (int jo = 0; jo < C::T2; jo += 2) {
      dib_f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = Bs[H1 + 32 * jo + dib_crow(r, h)]; acc1[r] = Bs[H1 + 32 * jo + 32 + dib_crow(r, h)]; }
      const float* w0p = Wt2 + (32 * jo + m) * C::P2 + 4 * h;
      const float* w1p = w0p + 32 * C::P2;
      float4 wa = *reinterpret_cast<const float4*>(w0p), wb = *reinterpret_cast<const float4*>(w1p);
#pragma unroll
      for (int st = 0; st < 4 * C::T1; ++st) {
        const int ji = st >> 2, g = st & 3, sn = st + 1 < 4 * C::T1 ? st + 1 : st;
        const float4 wan = *reinterpret_cast<const float4*>(w0p + 8 * sn), wbn = *reinterpret_cast<const float4*>(w1p + 8 * sn);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = DIB_MFMA(wa.x, h1[ji][4 * g + 0], acc0);
        acc1 = DIB_MFMA(wb.x, h1[ji][4 * g + 0], acc1);
        acc0 = DIB_MFMA(wa.y, h1[ji][4 * g + 1], acc0);
        acc1 = DIB_MFMA(wb.y, h1[ji][4 * g + 1], acc1);
        acc0 = DIB_MFMA(wa.z, h1[ji][4 * g + 2], acc0);
        acc1 = DIB_MFMA(wb.z, h1[ji][4 * g + 2], acc1);
        acc0 = DIB_MFMA(wa.w, h1[ji][4 * g + 3], acc0);
        acc1 = DIB_MFMA(wb.w, h1[ji][4 * g + 3], acc1);
        asm volatile("" : "+v"(acc0));
        asm volatile("" : "+v"(acc1));
        __builtin_amdgcn_sched_barrier(0);
        wa = wan; wb = wbn;
      }
      dib_act_tile<RELU>(slope, acc0);
      dib_act_tile<RELU>(slope, acc1);
      h2[jo] = acc0;
      h2[jo + 1] = acc1;
    }
    } else {
#pragma unroll
    for (int jo = 0; jo < C::T2; ++jo) {
      dib_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[H1 + 32 * jo + dib_crow(r, h)];
#pragma unroll
      for (int ji = 0; ji < C::T1; ++ji) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w = *reinterpret_cast<const float4*>(Wt2 + (32 * jo + m) * C::P2 + 32 * ji + 8 * g + 4 * h);
          acc = DIB_MFMA(w.x, h1[ji][4 * g + 0], acc);
          acc = DIB_MFMA(w.y, h1[ji][4 * g + 1], acc);
          acc = DIB_MFMA(w.z, h1[ji][4 * g + 2], acc);
          acc = DIB_MFMA(w.w, h1[ji][4 * g + 3], acc);
        }
      }
      dib_act_tile<RELU>(slope, acc);
      h2[jo] = acc;
    }
    }
    DIB_FT(2);   // layer 2 + activation
    if (a.h2 != nullptr) {
      float* dst = a.h2 + ((long long)f * a.batch + wrow0) * H2;
#pragma unroll
      for (int jo = 0; jo < C::T2; ++jo) dib_store_tile(patch, h2[jo], dst + 32 * jo, H2, rows_valid, lane);
    }
    DIB_FT(3);   // stash h2
    // act'(h2) as one bit per unit: the fused backward needs nothing else of h2
    if (a.h2mask != nullptr && valid) a.h2mask[((long long)f * a.batch + b) * 2 + h] = dib_sign_bits<C::T2>(h2);
    DIB_FT(4);   // activation mask bits
    // ---- layer 3 (linear, reference models.py:78): out^T = W3^T h2^T + b3 ; rows [0,E) = mu, [E,2E) = logvar ----
    dib_f32x16 o[C::T3];
    if constexpr (C::T3 == 2) {
      dib_f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = Bs[H1 + H2 + dib_crow(r, h)]; acc1[r] = Bs[H1 + H2 + 32 + dib_crow(r, h)]; }
      const float* w0p = Wt3 + m * C::P3 + 4 * h;
      const float* w1p = w0p + 32 * C::P3;
      float4 wa = *reinterpret_cast<const float4*>(w0p), wb = *reinterpret_cast<const float4*>(w1p);
#pragma unroll
      for (int st = 0; st < 4 * C::T2; ++st) {
        const int ji = st >> 2, g = st & 3, sn = st + 1 < 4 * C::T2 ? st + 1 : st;
        const float4 wan = *reinterpret_cast<const float4*>(w0p + 8 * sn), wbn = *reinterpret_cast<const float4*>(w1p + 8 * sn);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = DIB_MFMA(wa.x, h2[ji][4 * g + 0], acc0);
        acc1 = DIB_MFMA(wb.x, h2[ji][4 * g + 0], acc1);
        acc0 = DIB_MFMA(wa.y, h2[ji][4 * g + 1], acc0);
        acc1 = DIB_MFMA(wb.y, h2[ji][4 * g + 1], acc1);
        acc0 = DIB_MFMA(wa.z, h2[ji][4 * g + 2], acc0);
        acc1 = DIB_MFMA(wb.z, h2[ji][4 * g + 2], acc1);
        acc0 = DIB_MFMA(wa.w, h2[ji][4 * g + 3], acc0);
        acc1 = DIB_MFMA(wb.w, h2[ji][4 * g + 3], acc1);
        asm volatile("" : "+v"(acc0));
        asm volatile("" : "+v"(acc1));
        __builtin_amdgcn_sched_barrier(0);
        wa = wan; wb = wbn;
      }
      o[0] = acc0;
      o[1] = acc1;
    } else {
#pragma unroll
    for (int jo = 0; jo < C::T3; ++jo) {
      dib_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[H1 + H2 + 32 * jo + dib_crow(r, h)];
#pragma unroll
      for (int ji = 0; ji < C::T2; ++ji) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w = *reinterpret_cast<const float4*>(Wt3 + (32 * jo + m) * C::P3 + 32 * ji + 8 * g + 4 * h);
          acc = DIB_MFMA(w.x, h2[ji][4 * g + 0], acc);
          acc = DIB_MFMA(w.y, h2[ji][4 * g + 1], acc);
          acc = DIB_MFMA(w.z, h2[ji][4 * g + 2], acc);
          acc = DIB_MFMA(w.w, h2[ji][4 * g + 3], acc);
        }
      }
      o[jo] = acc;
    }
    }
    DIB_FT(5);   // layer 3
    // ---- epilogue: stash (mu|logvar), reparameterise (reference models.py:108), KL (models.py:111-112) ----
    // E % 32 == 0: mu tiles [0, E/32), logvar tiles [E/32, 2E/32), same register index.
    // E in {8,16}: one tile; mu in register groups g < E/8, logvar in groups g + E/8 (same lane).
    float klp = 0.f;
    float* eo = a.enc_out + ((long long)f * a.batch + b) * C::E2;
    float* up = a.U + (long long)b * ((long long)F * E) + (long long)f * E;
    constexpr int NG = E / 8;  // register groups (4 dims x 2 lane halves) holding mu
    dib_f32x16 ut[(E >= 32) ? (E / 32) : 1];  // sampled embeddings, same fragment layout as mu
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int t_mu = (E >= 32) ? (gi >> 2) : 0;
      const int g_mu = (E >= 32) ? (gi & 3) : gi;
      const int t_lv = (E >= 32) ? (t_mu + E / 32) : 0;
      const int g_lv = (E >= 32) ? g_mu : (gi + E / 8);
      const int e0 = 32 * t_mu + 8 * g_mu + 4 * h;  // first of 4 consecutive embedding dims held by this lane
      float eps[4] = {0.f, 0.f, 0.f, 0.f};
      if (!a.deterministic) dib_eps4(a.seed, nstep, (uint32_t)grow, (uint32_t)f, (uint32_t)(e0 >> 2), eps);
      float4 mu = make_float4(o[t_mu][4 * g_mu], o[t_mu][4 * g_mu + 1], o[t_mu][4 * g_mu + 2], o[t_mu][4 * g_mu + 3]);
      float4 lv = make_float4(o[t_lv][4 * g_lv], o[t_lv][4 * g_lv + 1], o[t_lv][4 * g_lv + 2], o[t_lv][4 * g_lv + 3]);
      float4 u;
      const float4 sg = make_float4(dib_sigma(lv.x), dib_sigma(lv.y), dib_sigma(lv.z), dib_sigma(lv.w));
      u.x = mu.x + sg.x * eps[0];
      u.y = mu.y + sg.y * eps[1];
      u.z = mu.z + sg.z * eps[2];
      u.w = mu.w + sg.w * eps[3];
      ut[t_mu][4 * g_mu] = u.x; ut[t_mu][4 * g_mu + 1] = u.y; ut[t_mu][4 * g_mu + 2] = u.z; ut[t_mu][4 * g_mu + 3] = u.w;
      if (valid) {
        if (E < 32) {  // narrow embeddings: direct 16-byte stores
          *reinterpret_cast<float4*>(eo + e0) = mu;
          *reinterpret_cast<float4*>(eo + E + e0) = lv;
          *reinterpret_cast<float4*>(up + e0) = u;
        }
        klp += 0.5f * ((mu.x * mu.x + sg.x * sg.x - lv.x - 1.f) + (mu.y * mu.y + sg.y * sg.y - lv.y - 1.f) +
                       (mu.z * mu.z + sg.z * sg.z - lv.z - 1.f) + (mu.w * mu.w + sg.w * sg.w - lv.w - 1.f));
      }
    }
    DIB_FT(6);   // eps, sigma, u, KL
    if (E >= 32) {  // full-line stores of (mu|logvar) [F][B][2E] and of u [B][F*E]
      float* eo_w = a.enc_out + ((long long)f * a.batch + wrow0) * C::E2;
#pragma unroll
      for (int jo = 0; jo < C::T3; ++jo) dib_store_tile(patch, o[jo], eo_w + 32 * jo, C::E2, rows_valid, lane);
      float* u_w = a.U + (long long)wrow0 * ((long long)F * E) + (long long)f * E;
#pragma unroll
      for (int jo = 0; jo < ((E >= 32) ? (E / 32) : 1); ++jo)
        dib_store_tile(patch, ut[jo], u_w + 32 * jo, (long long)F * E, rows_valid, lane);
    }
    DIB_FT(7);   // stash (mu|logvar), u
    kl_acc += dib_wave_sum(klp);
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = pn[r];
    DIB_FT(8);   // KL wave sum, loop end
  }
#ifdef DIB_FUSED_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    for (int i = 0; i < 9; ++i) dib_fused_dbg[i] = tacc_[i];
    dib_fused_dbg[9] = clock64() - tstart_;
    dib_fused_dbg[10] = ntl_;
    dib_fused_dbg[11] = wall_clock64() - wstart_;
  }
#endif
  if (lane == 0) a.kl_partial[((long long)blockIdx.x * 8 + wave) * F + f] = kl_acc;
}

// =====================================================================================================
// BACKWARD "dgrad chain" (what tape.gradient derives for reference models.py:106-118, explicit at
// train.py:203-219), ONE launch replacing the reparam/KL backward and both encoder dgrad GEMMs:
//   dmu = g_u + beta*mu/Bg ; dlogvar = g_u*(u - mu)*0.5 + beta*0.5*(exp(lv)-1)/Bg           -> dout  [F][B][2E]
//        (eps*sigma = u - mu: the forward's own sample is read back, 128 B per (sample, feature), instead of regenerating
//        Philox + Box-Muller - ~100 VALU instructions per 4 values; it is also the gradient of whatever forward ran)
//   dh2 = (dout @ W3^T) * act'(h2)   (act'(h2) from the 1-bit/unit mask the forward stashed)  -> dh2   [F][B][H2]
//   dh1 = (dh2  @ W2^T) * act'(h1)   (act'(h1) from its bit mask too: round 2 recomputed the h1 tile - 32 MFMAs + W1 / b1
//                                     fragments per 32 samples; the A/B is profiles/r03a_fused_diet_ab.txt)
//                                                                                              (never leaves the CU)
//   d(W1|b1) += [P | 1]^T @ dh1         (v_mfma_f32_16x16x4_f32 on the dh1 tile while it sits in the LDS patch;
//                                        per-wave partials, reduced in fixed order by dib_dw1_reduce_kernel)
// Same structure as the forward: one workgroup per feature, W2 / W3 resident in LDS in their natural
// [in][out] orientation (they are the A operand of the transposed product dH_in^T = W * dH_out^T, fetched with
// conflict-free ds_read_b128), gradients chained in MFMA accumulator registers.  Row-major tiles (mu|logvar, u, g_u in;
// dout, dh2 out) cross between HBM and the fragment layout through a wave-private LDS patch so every global access is a
// full 128-byte line.  The layer-2/3 weight gradients (contractions over the batch) then run as the grouped wgrad GEMMs on
// these buffers; the layer-1 gradient (5 x H1 per feature: hopeless as a GEMM tile, and dh1 is 2.1 GB) is finished here.
// =====================================================================================================
typedef float dib_f32x4 __attribute__((ext_vector_type(4)));
#define DIB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct DibFusedBwdArgs {
  const float* P; int batch;
  const float* params; const long long* w_off; const long long* b_off; const int4* featmap;
  int act;
  const unsigned long long* h1mask; const unsigned long long* h2mask;   // act' bits stashed by the forward
  const float* enc_out; const float* U; const float* GU;                // (mu|logvar) [F][B][2E]; u, dL/du [B][F*E]
  float* dout; float* dh2;
  float* dw1_partial;       // [gridDim.x*8 waves][F][16][H1]: per-wave partial of d(W1|b1) (row in_dim = bias gradient)
  const float* beta_dev; float inv_bg;
  int F;
};

template <int H1, int H2, int E>
struct DibFusedBwdCfg {
  static constexpr int E2 = 2 * E;
  static constexpr int N3 = (E2 + 31) / 32 * 32;
  static constexpr int P2 = H2 + 4, P3 = N3 + 4;        // W2 image [H1][H2+4], W3 image [H2][N3+4]
  static constexpr int W2_FLOATS = H1 * P2, W3_FLOATS = H2 * P3;
  static constexpr int PATCH = 32 * 36;
  static constexpr int LDS_FLOATS = W2_FLOATS + W3_FLOATS + 8 * PATCH;
  static constexpr int T1 = H1 / 32, T2 = H2 / 32, T3 = N3 / 32;
};

// issue the 4 coalesced 16-byte loads of one 32x32 row-major tile (rows clamped to the valid range)
struct DibTile4 { float4 a, b, c, d; };
__device__ __forceinline__ DibTile4 dib_tile_gload(const float* __restrict__ src, long long ld, int rows_valid, int lane) {
  const int cc = (lane & 7) * 4;
  DibTile4 t;
  // read-once tiles (mu|logvar, u, g_u): non-temporal loads (same-box A/B: step 8.11 -> 8.08 ms, profiles/r03v_*)
  auto ld4 = [&](int pss) {
    const dib_nt4 v = __builtin_nontemporal_load(reinterpret_cast<const dib_nt4*>(src + (long long)min(dib_patch_row(lane, pss), rows_valid - 1) * ld + cc));
    return make_float4(v.x, v.y, v.z, v.w);
  };
  t.a = ld4(0); t.b = ld4(1); t.c = ld4(2); t.d = ld4(3);
  return t;
}
// row-major tile (already in registers) -> transposed-product C fragment, through the wave-private LDS patch
__device__ __forceinline__ dib_f32x16 dib_tile_to_frag(float* __restrict__ patch, const DibTile4 t, int lane) {
  const int cc = (lane & 7) * 4, m = lane & 31, h = lane >> 5;
  *reinterpret_cast<float4*>(patch + dib_patch_row(lane, 0) * 36 + cc) = t.a;
  *reinterpret_cast<float4*>(patch + dib_patch_row(lane, 1) * 36 + cc) = t.b;
  *reinterpret_cast<float4*>(patch + dib_patch_row(lane, 2) * 36 + cc) = t.c;
  *reinterpret_cast<float4*>(patch + dib_patch_row(lane, 3) * 36 + cc) = t.d;
  __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; keep the compiler from reordering them
  const float4 q0 = *reinterpret_cast<const float4*>(patch + m * 36 + 4 * h);
  const float4 q1 = *reinterpret_cast<const float4*>(patch + m * 36 + 8 + 4 * h);
  const float4 q2 = *reinterpret_cast<const float4*>(patch + m * 36 + 16 + 4 * h);
  const float4 q3 = *reinterpret_cast<const float4*>(patch + m * 36 + 24 + 4 * h);
  __builtin_amdgcn_wave_barrier();
  dib_f32x16 c;
  c[0] = q0.x; c[1] = q0.y; c[2] = q0.z; c[3] = q0.w; c[4] = q1.x; c[5] = q1.y; c[6] = q1.z; c[7] = q1.w;
  c[8] = q2.x; c[9] = q2.y; c[10] = q2.z; c[11] = q2.w; c[12] = q3.x; c[13] = q3.y; c[14] = q3.z; c[15] = q3.w;
  return c;
}

// Sample row (of the wave's 32) that lane group lg (= lane >> 4) contracts in step s8 of the d(W1|b1) product: rows
// s8 + 8*lg - the four lane groups of a step read rows 8 apart (288 floats = bank offset 32), 16 lanes x 8 bytes each (one
// ds_read_b64 = hidden units 2j, 2j+1): the two groups of a 32-lane service half cover all 64 banks once.
__device__ __forceinline__ int dib_dw1_row(int s8, int lg) { return s8 + 8 * lg; }

template <int H1, int H2, int E, bool RELU>
__global__ void __launch_bounds__(512)
dib_fused_encoder_bwd_kernel(DibFusedBwdArgs a) {
  static_assert(E % 32 == 0, "fused backward: embedding dimension must be a multiple of 32");
  static_assert(H1 / 32 <= 4 && H2 / 32 <= 4, "act' bit masks: one 64-bit word per (lane, layer)");
  using C = DibFusedBwdCfg<H1, H2, E>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W2i = lds;                      // [H1][H2+4] W2[k1][k2]
  float* W3i = W2i + C::W2_FLOATS;       // [H2][N3+4] W3[k2][n], columns >= 2E zero
  float* patch = W3i + C::W3_FLOATS + (threadIdx.x >> 6) * C::PATCH;

  const int f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int4 fm = a.featmap[f];
  const int in_dim = fm.y;
  const int F = a.F;
  const float* Pf = a.P + (long long)fm.w * a.batch;
  {
    const float* W2 = a.params + a.w_off[1 * F + f];
    const float* W3 = a.params + a.w_off[2 * F + f];
    dib_stage_batched<H1 * H2>(tid, [&](int i) { return W2[i]; },
                               [&](int i, float v) { const int k1 = i / H2, k2 = i - k1 * H2; W2i[k1 * C::P2 + k2] = v; });
    dib_stage_batched<H2 * C::N3>(tid, [&](int i) { const int k2 = i / C::N3, n = i - k2 * C::N3;
                                                    return (n < C::E2) ? W3[(long long)k2 * C::E2 + n] : 0.f; },
                                  [&](int i, float v) { const int k2 = i / C::N3, n = i - k2 * C::N3; W3i[k2 * C::P3 + n] = v; });
  }
  __syncthreads();

  const int n_tiles = (a.batch + 255) / 256;
  const float slope = dib_neg_slope(a.act);
  const float kb = a.beta_dev[0] * a.inv_bg;
  // d(W1|b1) accumulators: 16x16 tiles (rows = encoder-input index k, row in_dim = bias; cols = 16 hidden units)
  dib_f32x4 dw1[2 * C::T1];
#pragma unroll
  for (int t = 0; t < 2 * C::T1; ++t) dw1[t] = dib_f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, lg = lane >> 4;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int wrow0 = tile * 256 + wave * 32;
    if (wrow0 >= a.batch) continue;  // wave-uniform; no barriers inside the loop
    const int rows_valid = min(32, a.batch - wrow0);
    const int b = min(wrow0 + m, a.batch - 1);
    const unsigned long long hbits = a.h2mask[((long long)f * a.batch + b) * 2 + h];  // act'(h2) bits, fragment order
    const unsigned long long h1bits = a.h1mask[((long long)f * a.batch + b) * 2 + h];
    // ---- dout = d(loss + beta*KL)/d(mu|logvar), lane-local in fragment layout ----
    dib_f32x16 dout[C::T3];
    {
      const float* eo = a.enc_out + ((long long)f * a.batch + wrow0) * C::E2;
      const long long so = (long long)wrow0 * ((long long)F * E) + (long long)f * E;
#pragma unroll
      for (int t = 0; t < E / 32; ++t) {
        const DibTile4 vm = dib_tile_gload(eo + 32 * t, C::E2, rows_valid, lane);
        const DibTile4 vl = dib_tile_gload(eo + E + 32 * t, C::E2, rows_valid, lane);
        const DibTile4 vg = dib_tile_gload(a.GU + so + 32 * t, (long long)F * E, rows_valid, lane);
        const DibTile4 vu = dib_tile_gload(a.U + so + 32 * t, (long long)F * E, rows_valid, lane);
        const dib_f32x16 mu = dib_tile_to_frag(patch, vm, lane);
        const dib_f32x16 lv = dib_tile_to_frag(patch, vl, lane);
        const dib_f32x16 g = dib_tile_to_frag(patch, vg, lane);
        const dib_f32x16 u = dib_tile_to_frag(patch, vu, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s = dib_sigma(lv[r]);
          dout[t][r] = g[r] + kb * mu[r];
          dout[t + E / 32][r] = g[r] * (u[r] - mu[r]) * 0.5f + kb * 0.5f * (s * s - 1.f);
        }
      }
      float* dd = a.dout + ((long long)f * a.batch + wrow0) * C::E2;
#pragma unroll
      for (int t = 0; t < C::T3; ++t) dib_store_tile(patch, dout[t], dd + 32 * t, C::E2, rows_valid, lane);
    }

    // encoded inputs for the layer-1 weight gradient (A operand: lane (i = lane&15, g = lane>>4) supplies
    // [P | 1][row dib_dw1_row(s, g)][i]); issued here so they land during the dh2 MFMAs
    float pa[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int rl = dib_dw1_row(s8, lg);            // row within the wave's 32 samples (see the dW1 MFMAs)
      const float v = Pf[(long long)min(wrow0 + rl, a.batch - 1) * in_dim + min(l15, in_dim - 1)];
      pa[s8] = (rl < rows_valid) ? (l15 < in_dim ? v : (l15 == in_dim ? 1.f : 0.f)) : 0.f;
    }

    // ---- dh2^T = W3 dout^T, masked by act'(h2) ----
    dib_f32x16 dh2[C::T2];
    {
      float* dg = a.dh2 + ((long long)f * a.batch + wrow0) * H2;
#pragma unroll
      for (int jo = 0; jo < C::T2; ++jo) {
        dib_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int jt = 0; jt < C::T3; ++jt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(W3i + (32 * jo + m) * C::P3 + 32 * jt + 8 * g + 4 * h);
            acc = DIB_MFMA(w.x, dout[jt][4 * g + 0], acc);
            acc = DIB_MFMA(w.y, dout[jt][4 * g + 1], acc);
            acc = DIB_MFMA(w.z, dout[jt][4 * g + 2], acc);
            acc = DIB_MFMA(w.w, dout[jt][4 * g + 3], acc);
          }
        dib_mask_tile<RELU>(acc, hbits, jo, slope);
        dh2[jo] = acc;
        dib_store_tile(patch, acc, dg + 32 * jo, H2, rows_valid, lane);
      }
    }

    // ---- dh1^T = W2 dh2^T, masked by act'(h1) ----
    {
#pragma unroll
      for (int jo = 0; jo < C::T1; ++jo) {
        dib_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int jt = 0; jt < C::T2; ++jt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(W2i + (32 * jo + m) * C::P2 + 32 * jt + 8 * g + 4 * h);
            acc = DIB_MFMA(w.x, dh2[jt][4 * g + 0], acc);
            acc = DIB_MFMA(w.y, dh2[jt][4 * g + 1], acc);
            acc = DIB_MFMA(w.z, dh2[jt][4 * g + 2], acc);
            acc = DIB_MFMA(w.w, dh2[jt][4 * g + 3], acc);
          }
        dib_mask_tile<RELU>(acc, h1bits, jo, slope);
        // dh1 tile -> LDS patch as row-major [m][n] (it never goes to HBM), then d(W1|b1) += [P|1]^T dh1 on 16x16x4 MFMAs:
        // step s contracts 4 samples (rows dib_dw1_row(s, g)); B operand: lane (j, g) reads dh1[row][.] from the patch.
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(patch + m * 36 + 8 * g + 4 * h) =
              make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int rl = dib_dw1_row(s8, lg);
          // accumulator 2*jo + c holds hidden units 32*jo + 2*j + c (j = lane & 15): the pair is one 8-byte read
          const float2 bb = *reinterpret_cast<const float2*>(patch + rl * 36 + 2 * l15);
          dw1[2 * jo] = DIB_MFMA16(pa[s8], bb.x, dw1[2 * jo]);
          dw1[2 * jo + 1] = DIB_MFMA16(pa[s8], bb.y, dw1[2 * jo + 1]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // per-wave partial of d(W1|b1): C/D map of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
  {
    float* dst = a.dw1_partial + (((long long)blockIdx.x * 8 + wave) * F + f) * (16ll * H1);
#pragma unroll
    for (int t = 0; t < 2 * C::T1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 32 * (t >> 1) + 2 * l15 + (t & 1);
        dst[(long long)(lg * 4 + r) * H1 + n] = dw1[t][r];
      }
  }
}

// grads[W1 block of feature f][k][n] = sum_p partial[p][f][k][n] (k < in_dim); grads[b1 block][n] = sum_p partial[p][f][in_dim][n]
// grid (F, 16 rows k): one workgroup per (feature, row) so the F*(in_dim+1) rows reduce in parallel; the sum over the
// partials keeps its fixed order (deterministic), 8 independent loads in flight per thread.
__global__ void __launch_bounds__(256)
dib_dw1_reduce_kernel(const float* __restrict__ partial, int nparts, int F, int H1, const long long* __restrict__ w_off,
                      const long long* __restrict__ b_off, const int4* __restrict__ featmap, float* __restrict__ grads) {
  const int f = blockIdx.x, k = blockIdx.y;
  const int in_dim = featmap[f].y;
  if (k > in_dim) return;
  const long long pstride = (long long)F * 16 * H1;
  for (int n = threadIdx.x; n < H1; n += 256) {
    const float* src = partial + ((long long)f * 16 + k) * H1 + n;
    float s = 0.f;
    int pz = 0;
    for (; pz + 8 <= nparts; pz += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(pz + u) * pstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; pz < nparts; ++pz) s += src[pz * pstride];
    if (k < in_dim) grads[w_off[f] + (long long)k * H1 + n] = s;
    else grads[b_off[f] + n] = s;
  }
}
