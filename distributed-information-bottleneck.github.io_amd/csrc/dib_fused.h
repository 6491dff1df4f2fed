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
  int F; unsigned long long seed; unsigned step; int deterministic;
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

// C-fragment row of accumulator register r for lane-half h:  (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ int dib_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Piecewise-linear activations only on the fused path (linear / relu / leaky_relu): act(v) = max(v,0) + slope*min(v,0)
// is branch-free, so the hot loop stays straight-line code.  Other activations use the general GEMM path.
__device__ __forceinline__ float dib_neg_slope(int act) { return act == 1 ? 0.f : (act == 2 ? 0.2f : 1.f); }
__device__ __forceinline__ void dib_act_tile(float slope, dib_f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f) + slope * fminf(v[r], 0.f);
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
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS ops are in order; this also pins the compiler
  const int rr = lane >> 3, cc = (lane & 7) * 4;
  float4 v[4];
#pragma unroll
  for (int pss = 0; pss < 4; ++pss) v[pss] = *reinterpret_cast<const float4*>(patch + (rr + 8 * pss) * 36 + cc);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch is reused by the next tile
  if (rows_valid >= 32) {  // wave-uniform fast path: four unconditional 16-byte stores
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) *reinterpret_cast<float4*>(dst + (long long)(rr + 8 * pss) * ld + cc) = v[pss];
  } else {
#pragma unroll
    for (int pss = 0; pss < 4; ++pss)
      if (rr + 8 * pss < rows_valid) *reinterpret_cast<float4*>(dst + (long long)(rr + 8 * pss) * ld + cc) = v[pss];
  }
}

template <int H1, int H2, int E>
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
    for (int i = tid; i < H1 * 32; i += 512) {  // Wt1[n][k], k < 32
      const int k = i / H1, n = i - k * H1;      // consecutive threads -> consecutive n (coalesced global reads)
      Wt1[n * C::K1P + k] = (k < in_dim) ? W1[(long long)k * H1 + n] : 0.f;
    }
    for (int i = tid; i < H1 * H2; i += 512) {
      const int k = i / H2, n = i - k * H2;
      Wt2[n * C::P2 + k] = W2[(long long)k * H2 + n];
    }
    for (int i = tid; i < H2 * C::N3; i += 512) {
      const int k = i / C::N3, n = i - k * C::N3;
      Wt3[n * C::P3 + k] = (n < C::E2) ? W3[(long long)k * C::E2 + n] : 0.f;
    }
    const float* b1 = a.params + a.b_off[0 * F + f];
    const float* b2 = a.params + a.b_off[1 * F + f];
    const float* b3 = a.params + a.b_off[2 * F + f];
    for (int i = tid; i < H1; i += 512) Bs[i] = b1[i];
    for (int i = tid; i < H2; i += 512) Bs[H1 + i] = b2[i];
    for (int i = tid; i < C::N3; i += 512) Bs[H1 + H2 + i] = (i < C::E2) ? b3[i] : 0.f;
  }
  __syncthreads();

  const int n_tiles = (a.batch + 255) / 256;
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

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile * 256 + wave * 32 + m;   // local batch row of this lane
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
      dib_act_tile(slope, acc);
      h1[jo] = acc;
    }
    // stash h1 (feature-major [F][B][H1]) for the backward pass
    {
      const int wrow0 = tile * 256 + wave * 32;
      const int rows_valid = min(32, a.batch - wrow0);
      float* dst = a.h1 + ((long long)f * a.batch + wrow0) * H1;
#pragma unroll
      for (int jo = 0; jo < C::T1; ++jo) dib_store_tile(patch, h1[jo], dst + 32 * jo, H1, rows_valid, lane);
    }

    // ---- layer 2: h2^T = act(W2^T h1^T + b2) ----
    dib_f32x16 h2[C::T2];
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
      dib_act_tile(slope, acc);
      h2[jo] = acc;
    }
    {
      const int wrow0 = tile * 256 + wave * 32;
      const int rows_valid = min(32, a.batch - wrow0);
      float* dst = a.h2 + ((long long)f * a.batch + wrow0) * H2;
#pragma unroll
      for (int jo = 0; jo < C::T2; ++jo) dib_store_tile(patch, h2[jo], dst + 32 * jo, H2, rows_valid, lane);
    }

    // ---- layer 3 (linear, reference models.py:78): out^T = W3^T h2^T + b3 ; rows [0,E) = mu, [E,2E) = logvar ----
    dib_f32x16 o[C::T3];
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
      if (!a.deterministic) dib_eps4(a.seed, a.step, (uint32_t)grow, (uint32_t)f, (uint32_t)(e0 >> 2), eps);
      float4 mu = make_float4(o[t_mu][4 * g_mu], o[t_mu][4 * g_mu + 1], o[t_mu][4 * g_mu + 2], o[t_mu][4 * g_mu + 3]);
      float4 lv = make_float4(o[t_lv][4 * g_lv], o[t_lv][4 * g_lv + 1], o[t_lv][4 * g_lv + 2], o[t_lv][4 * g_lv + 3]);
      float4 u;
      u.x = mu.x + expf(0.5f * lv.x) * eps[0];
      u.y = mu.y + expf(0.5f * lv.y) * eps[1];
      u.z = mu.z + expf(0.5f * lv.z) * eps[2];
      u.w = mu.w + expf(0.5f * lv.w) * eps[3];
      ut[t_mu][4 * g_mu] = u.x; ut[t_mu][4 * g_mu + 1] = u.y; ut[t_mu][4 * g_mu + 2] = u.z; ut[t_mu][4 * g_mu + 3] = u.w;
      if (valid) {
        if (E < 32) {  // narrow embeddings: direct 16-byte stores
          *reinterpret_cast<float4*>(eo + e0) = mu;
          *reinterpret_cast<float4*>(eo + E + e0) = lv;
          *reinterpret_cast<float4*>(up + e0) = u;
        }
        klp += 0.5f * ((mu.x * mu.x + expf(lv.x) - lv.x - 1.f) + (mu.y * mu.y + expf(lv.y) - lv.y - 1.f) +
                       (mu.z * mu.z + expf(lv.z) - lv.z - 1.f) + (mu.w * mu.w + expf(lv.w) - lv.w - 1.f));
      }
    }
    if (E >= 32) {  // full-line stores of (mu|logvar) [F][B][2E] and of u [B][F*E]
      const int wrow0 = tile * 256 + wave * 32;
      const int rows_valid = min(32, a.batch - wrow0);
      float* eo_w = a.enc_out + ((long long)f * a.batch + wrow0) * C::E2;
#pragma unroll
      for (int jo = 0; jo < C::T3; ++jo) dib_store_tile(patch, o[jo], eo_w + 32 * jo, C::E2, rows_valid, lane);
      float* u_w = a.U + (long long)wrow0 * ((long long)F * E) + (long long)f * E;
#pragma unroll
      for (int jo = 0; jo < ((E >= 32) ? (E / 32) : 1); ++jo)
        dib_store_tile(patch, ut[jo], u_w + 32 * jo, (long long)F * E, rows_valid, lane);
    }
    kl_acc += dib_wave_sum(klp);
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = pn[r];
  }
  if (lane == 0) a.kl_partial[((long long)blockIdx.x * 8 + wave) * F + f] = kl_acc;
}
