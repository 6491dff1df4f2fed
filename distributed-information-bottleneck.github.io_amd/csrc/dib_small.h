// dib_small.h - the Distributed-IB step for SMALL batches (round 5): 16-row tiles instead of "one workgroup owns a feature".
//
// The reference's own runs are small-batch (train.py:30-34: B = 128, 11 000 epochs; chaos notebook: B = 2048).  There the
// fused kernels of dib_fused.h put ONE workgroup on each feature - 10 workgroups on 256 CUs at the Boolean-circuit default,
// 31 us of which 11 us are the MFMAs of one CU - and the integration network ran as five GEMM launches of 8 workgroups
// each.  While (row tiles x features) stays within two rounds of the CUs (dib_api.hip small_regime; measured crossover
// profiles/r05x_small_batch_crossover.txt) the right decomposition is the other one: a workgroup owns 16 batch ROWS (the smallest MFMA tile,
// v_mfma_f32_16x16x4_f32, exact fp32), keeps those rows' activations in LDS through the whole layer chain and streams the
// weights from L2 straight into the MFMA B operand (each weight is used once per workgroup: nothing to stage).
//   dib_small_encoder_fwd_kernel  grid (row tiles, F): gather + PositionalEncoding (models.py:22-23) + Dense chain
//                                 (models.py:73-78,106) + reparameterisation (models.py:108) + KL partial (models.py:111-112)
//   dib_small_integration_kernel  grid (row tiles): concat(u) -> integration MLP (models.py:122) -> [1-unit head + loss + its
//                                 gradient | general output layer] -> dgrad chain back to dL/du; modes select the pieces
//   dib_small_encoder_bwd_kernel  grid (row tiles, F): d(mu|logvar) incl. beta*KL, dgrads of layers 3 and 2, and the
//                                 d(W1|b1) partial of the tile
// Weight gradients that contract over the batch (layers >= 2) are left to the grouped wgrad GEMM on the stashed operands.
// Every sum has a fixed order: bit-exact replay, like the large-batch path; results differ from it by fp32 rounding only.
#pragma once
#include "dib_common.h"
#include "dib_fused.h"   // dib_sigma, DIB_MFMA16, dib_f32x4

#define DIB_SMALL_ROWS 16
#define DIB_STD(i) do { if (dbg >= 0) DIB_ST(dbg + (i)); } while (0)   // marks inside a layer primitive (diagnostic build only)

// (phase marks DIB_ST(i) of the diagnostic build -DDIB_SMALL_TIMING: dib_common.h)
#define DIB_SMALL_THREADS 512   // 8 waves = 2 per SIMD: the contraction of every layer is split between wave w and w + 4, so
                                // that one of the pair issues MFMAs while the other waits for its weights (each weight is read
                                // once per workgroup, straight from L2: the kernels are latency-bound, not bandwidth-bound)
#define DIB_SMALL_XCH_FLOATS (4 * 5 * 64 * 4)   // exchange buffer of the wave pairs: [4 column slots][<= 5 tiles][64 lanes] float4
#define DIB_SMALL_XCH_FLOATS_WIDE (8 * 5 * 64 * 4)   // ... of the cluster-mode primitives: [8 waves][<= 5 tiles][64 lanes] float4

__host__ __device__ inline int dib_small_pick_nt(int n) {   // column tiles of 16 per wave pass: balance the 4 column slots first
  if (n % 64 == 0 && (n / 64) % 4 == 0) return 4;
  if (n % 32 == 0 && (n / 32) % 4 == 0) return 2;
  if ((n / 16) % 4 == 0) return 1;
  if (n % 64 == 0) return 4;
  if (n % 32 == 0) return 2;
  return 1;
}
// backward (tiles of 16 input units per wave pass): one pass when the tile count is 4 x {5, 4, 2, 1} (320 = 4 x 5 x 16: the
// dL/du of the default integration network), else like the forward
__host__ __device__ inline int dib_small_pick_nt_bwd(int kin) {
  const int tiles = kin / 16;
  if (tiles % 20 == 0) return 5;
  if (tiles % 16 == 0) return 4;
  if (tiles % 8 == 0) return 2;
  if (tiles % 4 == 0) return 1;
  if (tiles % 5 == 0) return 5;
  if (tiles % 4 == 0) return 4;
  if (tiles % 2 == 0) return 2;
  return 1;
}

// sum over the workgroup's 8 waves; result valid in thread 0.  `red` = 8 floats of LDS.
__device__ __forceinline__ float dib_small_block_sum(float v, float* red) {
  v = dib_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

template <int NT>
__device__ __forceinline__ void dib_small_loadw(const float* __restrict__ p, float (&b)[NT]) {
  if constexpr (NT == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
  } else if constexpr (NT == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    b[0] = v.x; b[1] = v.y;
  } else {
    b[0] = p[0];
  }
}

template <int NT>
__device__ __forceinline__ void dib_small_storev(float* p, const float (&v)[NT]) {
  if constexpr (NT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (NT == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  else p[0] = v[0];
}

// Piecewise-linear activations only (linear / relu / leaky_relu, like the fused path): act(v) = max(v,0) + slope min(v,0),
// act'(y) = y > 0 ? 1 : slope - branch-free epilogues.  Other activations take the general GEMM path.
__device__ __forceinline__ float dib_small_act(float slope, float v) { return fmaxf(v, 0.f) + slope * fminf(v, 0.f); }
__device__ __forceinline__ float dib_small_act_grad(float slope, float y) { return y > 0.f ? 1.f : slope; }

// out[16][N] = act(in[16][K] @ W[K][N] + bias).  in / out: LDS tiles (pitches pin / pout, pitch % 64 == 4: the A-operand
// dword reads of a wave hit 64 distinct banks); W: global, row-major, leading dimension N, rows >= kvalid read as zero
// (ragged first encoder layer).  Wave w owns column groups w, w + 4, ... of 16 NT columns; lane (j = lane & 15,
// q = lane >> 4) feeds A[row j][k = 4 s + q] and B[k = 4 s + q][columns n0 + NT j .. + NT) - NT consecutive floats of a
// weight row = one 4 NT-byte load, and the C fragment then holds NT CONSECUTIVE columns of rows 4 q .. 4 q + 3.
// The weights come straight from L2 (latency ~ 1 us): the loads of the NEXT batch of UB k-steps are issued before the MFMAs of
// the current one (register double buffer), so a wave always has UB loads in flight.
// gdst (optional): row-major global stash of the tile, leading dimension gld, rows < rows_valid.
template <int NT>
__device__ __forceinline__ void dib_small_fwd_nt(const float* in, int pin, int K, int kvalid, const float* __restrict__ W, int N,
                                                 const float* __restrict__ bias, float slope, float* out, int pout,
                                                 float* __restrict__ gdst, long long gld, int rows_valid, float* xch, int dbg = -1) {
  DIB_STD(0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  constexpr int CG = 16 * NT;
  // k-steps (of 4) per batch = weight loads a wave has in flight (32 k-steps for single-tile groups: with one
  // tile per group a batch of 16 steps is 0.25 us of MFMAs against a 1.3 us round trip - the set transformer's 1536 -> 32
  // output projection ran 12 exposed round trips per wave); batches alternate between the two waves of a pair
  constexpr int UB = NT == 4 ? 8 : (NT == 2 ? 16 : 32);
  const int ngroups = N / CG;
  // 8 waves = nslots column slots x kways shares of the contraction (batches interleaved between the shares).  A one-batch
  // contraction has nothing to split; few column groups give their idle slots to the contraction (the set transformer's
  // 1536 -> 32 output projection: 2 groups x 4 shares instead of 2 x 2 with 4 waves idle)
  const int nb_all = (K + 4 * UB - 1) / (4 * UB);
  // (same-box A/B against "always 4 x 2": set-transformer step 1.93 -> 1.87 ms, profiles/r05n_row_tile_variants_ab.txt)
  const int kways = nb_all <= 1 ? 1 : (ngroups >= 4 ? 2 : (ngroups >= 2 || NT > 2 ? 4 : 8));
  const bool ksplit = kways > 1;
  const int nslots = 8 / kways;
  const int wc = wave % nslots, kh = wave / nslots;   // column slot, contraction share
  for (int g0 = 0; g0 < ngroups; g0 += nslots) {   // block-uniform trip count: the barriers below are reached by every wave
    const bool active = g0 + wc < ngroups;
    const int n0 = (active ? g0 + wc : 0) * CG;
    dib_f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const float bv = kh == 0 ? bias[n0 + NT * j + c] : 0.f;
      acc[c] = dib_f32x4{bv, bv, bv, bv};
    }
    if (active) {
      const float* ap = in + j * pin;
      const float* wp = W + n0 + NT * j;
      // the weights are double-buffered in registers; the A operand (LDS, ~100 cycles) is too for NT >= 2 - with one tile per
      // group (NT == 1, 32-step batches) it is read when its batch is computed: the registers go to the weights in flight
      constexpr bool kBufA = NT >= 2;
      constexpr int UA = kBufA ? UB : 1;
      float acur[UA], bcur[UB][NT];
      auto load = [&](int s0, float (&av)[UA], float (&bv)[UB][NT]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = s0 + 4 * u + q;
          const bool ok = k < kvalid;
          const int kc = ok ? k : 0;
          if constexpr (kBufA) av[u] = ok ? ap[kc] : 0.f;
          dib_small_loadw<NT>(wp + (long long)kc * N, bv[u]);
        }
      };
      load(4 * UB * kh, acur, bcur);
      DIB_STD(1);
      for (int s0 = 4 * UB * kh; s0 < K; s0 += 4 * UB * kways) {
        float anxt[UA], bnxt[UB][NT];
        load(s0 + 4 * UB * kways, anxt, bnxt);   // past the end: masked (k >= kvalid), harmless re-read of row 0
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (s0 + 4 * u < K) {   // wave-uniform: the ragged last batch skips its empty k-steps
            float av;
            if constexpr (kBufA) av = acur[u];
            else { const int k = s0 + 4 * u + q; av = k < kvalid ? ap[k] : 0.f; }
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = DIB_MFMA16(av, bcur[u][c], acc[c]);
          }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if constexpr (kBufA) acur[u] = anxt[u];
#pragma unroll
          for (int c = 0; c < NT; ++c) bcur[u][c] = bnxt[u][c];
        }
      }
    }
    DIB_STD(2);
    // shares 1 .. kways - 1 hand their partial sums to share 0, which finishes the tile (fixed order: share 0 + 1 + 2 + ...);
    // exchange slot (kh - 1) * nslots + wc: at most 7 x NT <= 20 entries (kways == 8 only with NT <= 2)
    if (kh >= 1 && active) {
#pragma unroll
      for (int c = 0; c < NT; ++c)
        *reinterpret_cast<float4*>(xch + ((((kh - 1) * nslots + wc) * NT + c) * 64 + lane) * 4) =
            make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }
    __syncthreads();
    DIB_STD(3);
    if (kh == 0 && active) {
      for (int p = 1; p < kways; ++p) {
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          const float4 o = *reinterpret_cast<const float4*>(xch + ((((p - 1) * nslots + wc) * NT + c) * 64 + lane) * 4);
          acc[c][0] += o.x; acc[c][1] += o.y; acc[c][2] += o.z; acc[c][3] += o.w;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        float v[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) v[c] = dib_small_act(slope, acc[c][r]);
        if (out != nullptr) dib_small_storev<NT>(out + row * pout + n0 + NT * j, v);
        if (gdst != nullptr && row < rows_valid) dib_small_storev<NT>(gdst + (long long)row * gld + n0 + NT * j, v);
      }
    }
    DIB_STD(4);
    __syncthreads();   // the exchange buffer is free again, the output tile is visible
    DIB_STD(5);
  }
}

// (ends with a workgroup barrier: the output tile is visible to every wave on return)
__device__ __forceinline__ void dib_small_fwd(const float* in, int pin, int K, int kvalid, const float* __restrict__ W, int N,
                                              const float* __restrict__ bias, float slope, float* out, int pout,
                                              float* __restrict__ gdst, long long gld, int rows_valid, float* xch, int dbg = -1) {
  switch (dib_small_pick_nt(N)) {   // block-uniform
    case 4: dib_small_fwd_nt<4>(in, pin, K, kvalid, W, N, bias, slope, out, pout, gdst, gld, rows_valid, xch, dbg); break;
    case 2: dib_small_fwd_nt<2>(in, pin, K, kvalid, W, N, bias, slope, out, pout, gdst, gld, rows_valid, xch, dbg); break;
    default: dib_small_fwd_nt<1>(in, pin, K, kvalid, W, N, bias, slope, out, pout, gdst, gld, rows_valid, xch, dbg); break;
  }
}

// gin[16][Kin] = (g[16][N] @ W[Kin][N]^T) (.) act'(h[16][Kin])   (h == nullptr: no activation in front, e.g. dL/du).
// The contraction runs along the weight rows: lane (j, q) feeds A[row j][n = 16 S + 4 q + c] (one ds_read_b128 per 4 MFMAs)
// and B[n][k = k0 + 16 t + j] = W[k][16 S + 4 q + c] (one 16-byte load per tile t and 4 MFMAs); batches of UB S-steps are
// double-buffered in registers like the forward's.  gin: LDS tile or nullptr; gdst: optional global row-major stash.
template <int NT>
__device__ __forceinline__ void dib_small_bwd_nt(const float* g, int pg, int N, const float* __restrict__ W, int Kin,
                                                 const float* h, int ph, float slope, float* gin, int pgi,
                                                 float* __restrict__ gdst, long long gld, int rows_valid, float* xch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  constexpr int CG = 16 * NT;
  constexpr int UB = NT >= 4 ? 2 : 4;   // S-steps (of 16 contraction indices) per batch
  const int ngroups = Kin / CG;
  const bool ksplit = N > 16 * UB;   // a one-batch contraction is not split: 8 column slots instead of 4 x 2
  const int nslots = ksplit ? 4 : 8;
  const int wc = ksplit ? (wave & 3) : wave, kh = ksplit ? (wave >> 2) : 0;
  for (int g0 = 0; g0 < ngroups; g0 += nslots) {
    const bool active = g0 + wc < ngroups;
    const int k0 = (active ? g0 + wc : 0) * CG;
    dib_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = dib_f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      const float* ap = g + j * pg + 4 * q;
      const float* wp = W + (long long)(k0 + j) * N + 4 * q;
      float4 acur[UB], bcur[UB][NT];
      auto load = [&](int S0, float4 (&av)[UB], float4 (&bv)[UB][NT]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int S = S0 + 16 * u;
          const bool ok = S < N;
          const int Sc = ok ? S : 0;
          av[u] = ok ? *reinterpret_cast<const float4*>(ap + Sc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int t = 0; t < NT; ++t) bv[u][t] = *reinterpret_cast<const float4*>(wp + (long long)(16 * t) * N + Sc);
        }
      };
      load(16 * UB * kh, acur, bcur);
      for (int S0 = 16 * UB * kh; S0 < N; S0 += 32 * UB) {
        float4 anxt[UB], bnxt[UB][NT];
        load(S0 + 32 * UB, anxt, bnxt);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (S0 + 16 * u >= N) break;   // wave-uniform
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].x, bcur[u][t].x, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].y, bcur[u][t].y, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].z, bcur[u][t].z, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].w, bcur[u][t].w, acc[t]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          acur[u] = anxt[u];
#pragma unroll
          for (int t = 0; t < NT; ++t) bcur[u][t] = bnxt[u][t];
        }
      }
    }
    if (kh == 1 && active) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        *reinterpret_cast<float4*>(xch + ((wc * 5 + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
    __syncthreads();
    if (kh == 0 && active) {
      if (ksplit) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 o = *reinterpret_cast<const float4*>(xch + ((wc * 5 + t) * 64 + lane) * 4);
          acc[t][0] += o.x; acc[t][1] += o.y; acc[t][2] += o.z; acc[t][3] += o.w;
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * q + r, k = k0 + 16 * t + j;
          float v = acc[t][r];
          if (h != nullptr) v *= dib_small_act_grad(slope, h[row * ph + k]);
          if (gin != nullptr) gin[row * pgi + k] = v;
          if (gdst != nullptr && row < rows_valid) gdst[(long long)row * gld + k] = v;
        }
    }
    __syncthreads();
  }
}

// (ends with a workgroup barrier)
__device__ __forceinline__ void dib_small_bwd(const float* g, int pg, int N, const float* __restrict__ W, int Kin,
                                              const float* h, int ph, float slope, float* gin, int pgi,
                                              float* __restrict__ gdst, long long gld, int rows_valid, float* xch) {
  switch (dib_small_pick_nt_bwd(Kin)) {   // block-uniform
    case 5: dib_small_bwd_nt<5>(g, pg, N, W, Kin, h, ph, slope, gin, pgi, gdst, gld, rows_valid, xch); break;
    case 4: dib_small_bwd_nt<4>(g, pg, N, W, Kin, h, ph, slope, gin, pgi, gdst, gld, rows_valid, xch); break;
    case 2: dib_small_bwd_nt<2>(g, pg, N, W, Kin, h, ph, slope, gin, pgi, gdst, gld, rows_valid, xch); break;
    default: dib_small_bwd_nt<1>(g, pg, N, W, Kin, h, ph, slope, gin, pgi, gdst, gld, rows_valid, xch); break;
  }
}

// =====================================================================================================================
// cluster-mode layer primitives (a column SLICE of a layer on one workgroup, see "cluster mode" below)
// =====================================================================================================================
// A slice is narrow - 64 columns of a 256-wide layer on a cluster of 4 - and what bounds a layer on a row tile is the latency of
// its weight stream (the weights are read once per XCD: every batch of loads is an L2 miss, ~ 2 us).  So: the slice's columns are
// ONE group (two for 128 columns) and the waves the group rule would leave idle take shares of the CONTRACTION, sized so that a
// share's loads are a single batch in flight - one memory round trip per layer; every share writes its partial tile to the exchange
// buffer and the tiles are reduced in share order (fixed: 0 + 1 + ... ) by ONE WAVE PER TILE, which also applies bias / activation
// (its bias values fetched at the start of the pass, under the weight stream) and stores - 8 partial reads per wave instead of 28
// on the share-0 wave.
template <int NT, int UB>
__device__ __forceinline__ void dib_small_fwd_wide(const float* in, int pin, int K, const float* __restrict__ W, int ldw, int ncols,
                                                   const float* __restrict__ bias, float slope, float* out, int pout,
                                                   float* __restrict__ gdst, long long gld, int rows_valid, float* xch, int dbg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  constexpr int CG = 16 * NT;
  const int ngroups = ncols / CG;
  const int kways = ngroups <= 1 ? 8 : (ngroups == 2 ? 4 : 2), nslots = 8 / kways;
  const int wc = wave % nslots, kh = wave / nslots;   // column slot, contraction share
  DIB_STD(0);
  for (int g0 = 0; g0 < ngroups; g0 += nslots) {   // block-uniform trip count
    const bool active = g0 + wc < ngroups;
    const int n0 = (active ? g0 + wc : 0) * CG;
    // this wave's reduction items of the pass: tile (slot, c) = item / NT, item % NT for items wave, wave + 8
    float bv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = wave + 8 * it, slot = item / NT, c = item - slot * NT;
      bv[it] = (slot < nslots && g0 + slot < ngroups) ? bias[(g0 + slot) * CG + NT * j + c] : 0.f;
    }
    dib_f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = dib_f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      const float* ap = in + j * pin;
      const float* wp = W + n0 + NT * j;
      float acur[UB], bcur[UB][NT];
      auto load = [&](int s0, float (&av)[UB], float (&bw)[UB][NT]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k = s0 + 4 * u + q;
          const bool ok = k < K;
          const int kc = ok ? k : 0;
          av[u] = ok ? ap[kc] : 0.f;
          dib_small_loadw<NT>(wp + (long long)kc * ldw, bw[u]);
        }
      };
      // (no register double buffer: a share is ONE batch wherever the caller could size it so - a second batch pays its round trip)
      for (int s0 = 4 * UB * kh; s0 < K; s0 += 4 * UB * kways) {
        load(s0, acur, bcur);
        DIB_STD(1);
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (s0 + 4 * u < K) {   // wave-uniform
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = DIB_MFMA16(acur[u], bcur[u][c], acc[c]);
          }
      }
    }
    DIB_STD(2);
    if (active) {
#pragma unroll
      for (int c = 0; c < NT; ++c)
        *reinterpret_cast<float4*>(xch + (((kh * nslots + wc) * NT + c) * 64 + lane) * 4) =
            make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }
    __syncthreads();
    DIB_STD(3);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = wave + 8 * it, slot = item / NT, c = item - slot * NT;
      if (slot < nslots && g0 + slot < ngroups) {   // wave-uniform
        float4 sum = *reinterpret_cast<const float4*>(xch + ((slot * NT + c) * 64 + lane) * 4);
        for (int p = 1; p < kways; ++p) {
          const float4 o = *reinterpret_cast<const float4*>(xch + (((p * nslots + slot) * NT + c) * 64 + lane) * 4);
          sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
        }
        const int n = (g0 + slot) * CG + NT * j + c;
        const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * q + r;
          const float v = dib_small_act(slope, v4[r] + bv[it]);
          if (out != nullptr) out[row * pout + n] = v;
          if (gdst != nullptr && row < rows_valid) gdst[(long long)row * gld + n] = v;
        }
      }
    }
    DIB_STD(4);
    __syncthreads();   // the exchange buffer is free again, the output tile is visible
    DIB_STD(5);
  }
}

// columns [col0, col0 + ncols) of the layer only (W, bias, out, gdst: the FULL layer's pointers; ldn = its width)
__device__ __forceinline__ void dib_small_fwd_cols(const float* in, int pin, int K, const float* __restrict__ W, int ldn, int col0,
                                                   int ncols, const float* __restrict__ bias, float slope, float* out, int pout,
                                                   float* __restrict__ gdst, int rows_valid, float* xch, int dbg = -1) {
  W += col0; bias += col0; gdst += col0;
  if (out != nullptr) out += col0;
  // the widest column group the slice divides into; k-steps per batch = a share's whole part of the contraction where it fits
  const int nt = ncols % 64 == 0 ? 4 : (ncols % 32 == 0 ? 2 : 1);
  const int ngroups = ncols / (16 * nt);
  const int kw = ngroups <= 1 ? 8 : (ngroups == 2 ? 4 : 2);
  const int per_share = ((K + 3) / 4 + kw - 1) / kw;
  const int ub = per_share <= 8 ? 8 : (per_share <= 10 ? 10 : 16);
#define DIB_FW(NT_, UB_) dib_small_fwd_wide<NT_, UB_>(in, pin, K, W, ldn, ncols, bias, slope, out, pout, gdst, ldn, rows_valid, xch, dbg)
  if (nt == 4) { if (ub == 8) DIB_FW(4, 8); else if (ub == 10) DIB_FW(4, 10); else DIB_FW(4, 16); }
  else if (nt == 2) { if (ub == 8) DIB_FW(2, 8); else if (ub == 10) DIB_FW(2, 10); else DIB_FW(2, 16); }
  else { if (ub == 8) DIB_FW(1, 8); else if (ub == 10) DIB_FW(1, 10); else DIB_FW(1, 16); }
#undef DIB_FW
}

// gin[16][16 NT G] = (g[16][N] @ W[16 NT G][N]^T) (.) act'(h): G = 1 or 2 groups of NT <= 5 input-unit tiles, the contraction on
// `kways` <= 8 / G shares of UB S-steps a batch; partial tiles reduced in share order, one wave per tile (see dib_small_fwd_wide).
template <int NT, int UB>
__device__ __forceinline__ void dib_small_bwd_wide(const float* g, int pg, int N, const float* __restrict__ W, const float* h, int ph,
                                                   float slope, float* gin, int pgi, float* __restrict__ gdst, long long gld,
                                                   int rows_valid, float* xch, int kways, int dbg, int G = 1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int wc = wave % G, kh = wave / G;   // tile group, contraction share
  const bool active = kh < kways;
  DIB_STD(0);
  dib_f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = dib_f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float* ap = g + j * pg + 4 * q;
    const float* wp = W + (long long)(16 * NT * wc + j) * N + 4 * q;
    float4 acur[UB], bcur[UB][NT];
    auto load = [&](int S0, float4 (&av)[UB], float4 (&bw)[UB][NT]) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int S = S0 + 16 * u;
        const bool ok = S < N;
        const int Sc = ok ? S : 0;
        av[u] = ok ? *reinterpret_cast<const float4*>(ap + Sc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < NT; ++t) bw[u][t] = *reinterpret_cast<const float4*>(wp + (long long)(16 * t) * N + Sc);
      }
    };
    for (int S0 = 16 * UB * kh; S0 < N; S0 += 16 * UB * kways) {   // (no double buffer: see dib_small_fwd_wide)
      load(S0, acur, bcur);
      DIB_STD(1);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (S0 + 16 * u >= N) break;   // wave-uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].x, bcur[u][t].x, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].y, bcur[u][t].y, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].z, bcur[u][t].z, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = DIB_MFMA16(acur[u].w, bcur[u][t].w, acc[t]);
      }
    }
    DIB_STD(2);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      *reinterpret_cast<float4*>(xch + (((kh * G + wc) * NT + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  __syncthreads();
  DIB_STD(3);
  for (int item = wave; item < G * NT; item += 8) {   // tile (slot, t) = item / NT, item % NT (two groups of 5: ten tiles on 8 waves)
    const int slot = item / NT, t = item - slot * NT;
    float4 sum = *reinterpret_cast<const float4*>(xch + ((slot * NT + t) * 64 + lane) * 4);
    for (int p = 1; p < kways; ++p) {
      const float4 o = *reinterpret_cast<const float4*>(xch + (((p * G + slot) * NT + t) * 64 + lane) * 4);
      sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
    }
    const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r, k = 16 * (NT * slot + t) + j;
      float v = v4[r];
      if (h != nullptr) v *= dib_small_act_grad(slope, h[row * ph + k]);
      if (gin != nullptr) gin[row * pgi + k] = v;
      if (gdst != nullptr && row < rows_valid) gdst[(long long)row * gld + k] = v;
    }
  }
  DIB_STD(4);
  __syncthreads();
  DIB_STD(5);
}

// input units [k0, k0 + kcols) of the layer only (W, h, gin, gdst: the FULL layer's pointers; ldk = its input width)
__device__ __forceinline__ void dib_small_bwd_cols(const float* g, int pg, int N, const float* __restrict__ W, int ldk, int k0,
                                                   int kcols, const float* h, int ph, float slope, float* gin, int pgi,
                                                   float* __restrict__ gdst, int rows_valid, float* xch, int dbg = -1) {
  W += (long long)k0 * N;
  if (h != nullptr) h += k0;
  if (gin != nullptr) gin += k0;
  if (gdst != nullptr) gdst += k0;
  int tiles = kcols >> 4;
  // up to 5 tiles: one group; 6 / 8 / 10: two groups of 3 / 4 / 5 on half the shares each; anything else: the general primitive
  const int G = tiles <= 5 ? 1 : ((tiles <= 10 && (tiles & 1) == 0) ? 2 : 0);
  if (G == 0) { dib_small_bwd(g, pg, N, W, kcols, h, ph, slope, gin, pgi, gdst, ldk, rows_valid, xch); return; }
  tiles /= G;
  // 2 S-steps a batch (N <= 256: a share's loads are one batch in flight), 4 for longer contractions
  const int kmax = 8 / G;
  const int ub = N <= 16 * kmax ? 1 : (N <= 256 ? 2 : 4);   // (a short contraction: one S-step a share, every wave busy)
  const int nb = (N + 16 * ub - 1) / (16 * ub);
  const int kw = nb >= 8 && kmax >= 8 ? 8 : (nb >= 4 && kmax >= 4 ? 4 : (nb >= 2 ? 2 : 1));
#define DIB_BW(NT_) do { if (ub == 1) dib_small_bwd_wide<NT_, 1>(g, pg, N, W, h, ph, slope, gin, pgi, gdst, ldk, rows_valid, xch, kw, dbg, G); \
                         else if (ub == 2) dib_small_bwd_wide<NT_, 2>(g, pg, N, W, h, ph, slope, gin, pgi, gdst, ldk, rows_valid, xch, kw, dbg, G); \
                         else dib_small_bwd_wide<NT_, 4>(g, pg, N, W, h, ph, slope, gin, pgi, gdst, ldk, rows_valid, xch, kw, dbg, G); } while (0)
  switch (tiles) {   // block-uniform
    case 5: DIB_BW(5); break;
    case 4: DIB_BW(4); break;
    case 3: DIB_BW(3); break;
    case 2: DIB_BW(2); break;
    case 1: DIB_BW(1); break;
    default: break;   // an empty slice
  }
#undef DIB_BW
}

// global [rows_valid][width] (leading dimension ld) -> LDS tile [16][pitch]; rows >= rows_valid are zero-filled
__device__ __forceinline__ void dib_small_load_tile(const float* __restrict__ src, long long ld, int width, int rows_valid,
                                                    float* dst, int pitch) {
  // Four loads per thread in flight per trip (the rolled loop the compiler makes of the plain form issues ONE load per trip
  // and waits for it: the set-transformer chain's 16 x 1536 context tile was 12 dependent L2 round trips, ~9 us of a 45 us
  // kernel).  A variant with each thread owning a float4 column of all 16 rows - 16 loads in flight - measured 1.5 % SLOWER
  // on the set-transformer step (profiles/r05n_row_tile_variants_ab.txt): on the many 32-wide tiles only 8 threads had work.
  const int w4 = width >> 2, total = DIB_SMALL_ROWS * w4;
#pragma unroll 1
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * DIB_SMALL_THREADS) {
    float4 v[4];
    int off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * DIB_SMALL_THREADS;
      const int row = i / w4, c = (i - row * w4) * 4;
      off[u] = row * pitch + c;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total && row < rows_valid) v[u] = *reinterpret_cast<const float4*>(src + (long long)row * ld + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * DIB_SMALL_THREADS < total) *reinterpret_cast<float4*>(dst + off[u]) = v[u];
  }
}


__host__ __device__ inline int dib_small_pitch(int width) { return (width + 63) / 64 * 64 + 4; }   // pitch % 64 == 4

// =====================================================================================================================
// encoder bank forward
// =====================================================================================================================
struct DibSmallEncFwdArgs {
  const float* X; long long ldx; const int* row_idx; long long row0; int batch;
  const float* params; const long long* w_off; const long long* b_off; const int4* featmap;
  int n_blocks, act, F, E, H1, H2;
  float* P; float* h1; float* h2;            // stashes for the backward pass (nullptr: inference)
  float* enc_out; float* U; float* kl_partial;   // kl_partial[row tile][F]
  unsigned long long seed; unsigned step; int deterministic; const unsigned* step_dev;
};

__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_encoder_fwd_kernel(DibSmallEncFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[8];
  const int f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int r0 = tile * DIB_SMALL_ROWS, rows_valid = min(DIB_SMALL_ROWS, a.batch - r0);
  const int4 fm = a.featmap[f];   // {d_f, in_dim_f, x column offset, sum of in_dim of earlier features}
  const int d = fm.x, in_dim = fm.y, F = a.F, E = a.E, E2 = 2 * a.E;
  const int p1 = dib_small_pitch(a.H1), p2 = dib_small_pitch(a.H2), p3 = dib_small_pitch(E2);
  float* Pl = lds;                          // [16][20]: encoder input, columns >= in_dim zero
  float* h1s = Pl + DIB_SMALL_ROWS * 20;    // [16][p1]
  float* h2s = h1s + DIB_SMALL_ROWS * p1;   // [16][p2]
  float* os = h2s + DIB_SMALL_ROWS * p2;    // [16][p3]
  float* xch = os + DIB_SMALL_ROWS * p3;    // wave-pair exchange
  // ---- gather + positional encoding (the expressions of dib_posenc_kernel): P[b][j d + c] = j == 0 ? x : sin(2^j x) ----
  DIB_ST(0);
  for (int i = tid; i < DIB_SMALL_ROWS * 20; i += DIB_SMALL_THREADS) Pl[i] = 0.f;
  __syncthreads();
  if (tid < DIB_SMALL_ROWS * d) {
    const int row = tid / d, c = tid - row * d;
    if (row < rows_valid) {
      const int b = r0 + row;
      const long long grow = a.row_idx ? (long long)a.row_idx[b] : a.row0 + b;
      const float x = a.X[grow * a.ldx + fm.z + c];
      float* pg = a.P ? a.P + (long long)fm.w * a.batch + (long long)b * in_dim + c : nullptr;
      Pl[row * 20 + c] = x;
      if (pg) pg[0] = x;
      float fr = 2.0f;
      for (int jb = 1; jb < a.n_blocks; ++jb) {
        const float v = sinf(fr * x);
        Pl[row * 20 + jb * d + c] = v;
        if (pg) pg[(long long)jb * d] = v;
        fr *= 2.0f;
      }
    }
  }
  __syncthreads();
  DIB_ST(1);
  const float* W1 = a.params + a.w_off[0 * F + f];
  const float* W2 = a.params + a.w_off[1 * F + f];
  const float* W3 = a.params + a.w_off[2 * F + f];
  const float* b1 = a.params + a.b_off[0 * F + f];
  const float* b2 = a.params + a.b_off[1 * F + f];
  const float* b3 = a.params + a.b_off[2 * F + f];
  const long long frow = (long long)f * a.batch + r0;
  const float slope = dib_neg_slope(a.act);
  dib_small_fwd(Pl, 20, (in_dim + 3) & ~3, in_dim, W1, a.H1, b1, slope, h1s, p1, a.h1 ? a.h1 + frow * a.H1 : nullptr, a.H1,
                rows_valid, xch, 6);
  DIB_ST(2);
  // layers 2 and 3 on the slice primitives of the cluster mode (the whole layer as one "slice": float4 weight loads, every wave a
  // share of the contraction, one batch in flight): 4.3 + 4.3 -> see profiles/r06x_*; layer 1's ragged first dimension stays
  dib_small_fwd_cols(h1s, p1, a.H1, W2, a.H2, 0, a.H2, b2, slope, h2s, p2, a.h2 ? a.h2 + frow * a.H2 : nullptr, rows_valid, xch, 45);
  DIB_ST(3);
  dib_small_fwd_cols(h2s, p2, a.H2, W3, E2, 0, E2, b3, 1.f /* linear, models.py:78 */, os, p3, a.enc_out + frow * E2, rows_valid, xch, 34);
  DIB_ST(4);
  // ---- reparameterise + KL: thread = (row, 4 consecutive embedding dims) = one Philox call ----
  const int E4 = E >> 2;
  float klp = 0.f;
  const unsigned nstep = a.step_dev ? a.step_dev[0] : a.step;
  for (int i = tid; i < DIB_SMALL_ROWS * E4; i += DIB_SMALL_THREADS) {
    const int row = i / E4, qq = i - row * E4;
    if (row >= rows_valid) continue;
    const int b = r0 + row;
    const long long grow = a.row_idx ? (long long)a.row_idx[b] : a.row0 + b;
    const float4 mu = *reinterpret_cast<const float4*>(os + row * p3 + 4 * qq);
    const float4 lv = *reinterpret_cast<const float4*>(os + row * p3 + E + 4 * qq);
    float eps[4] = {0.f, 0.f, 0.f, 0.f};
    if (!a.deterministic) dib_eps4(a.seed, nstep, (uint32_t)grow, (uint32_t)f, (uint32_t)qq, eps);
    const float4 sg = make_float4(dib_sigma(lv.x), dib_sigma(lv.y), dib_sigma(lv.z), dib_sigma(lv.w));
    float4 u;
    u.x = mu.x + sg.x * eps[0];
    u.y = mu.y + sg.y * eps[1];
    u.z = mu.z + sg.z * eps[2];
    u.w = mu.w + sg.w * eps[3];
    *reinterpret_cast<float4*>(a.U + (long long)b * ((long long)F * E) + (long long)f * E + 4 * qq) = u;
    klp += 0.5f * ((mu.x * mu.x + sg.x * sg.x - lv.x - 1.f) + (mu.y * mu.y + sg.y * sg.y - lv.y - 1.f) +
                   (mu.z * mu.z + sg.z * sg.z - lv.z - 1.f) + (mu.w * mu.w + sg.w * sg.w - lv.w - 1.f));
  }
  const float tot = dib_small_block_sum(klp, red);
  if (tid == 0) a.kl_partial[(long long)tile * F + f] = tot;
  DIB_ST(5);
}

// =====================================================================================================================
// integration network: forward, 1-unit head + loss, dgrad chain
// =====================================================================================================================
#define DIB_SMALL_INT_FWD 1        // hidden layers u -> h_0 .. h_{n-1} (stashed unless DIB_SMALL_INT_INFER)
#define DIB_SMALL_INT_OUT 2        // general output layer -> pred (out_dim % 16 == 0)
#define DIB_SMALL_INT_HEAD 4       // 1-unit linear output + BCE-from-logits / MSE: pred, loss partials
#define DIB_SMALL_INT_HEAD_GRAD 8  // ... and its backward: g_pred, dL/dh_{n-1}, (W|b) gradient partial of the tile
#define DIB_SMALL_INT_BWD_OUT 16   // dL/dh_{n-1} = (g_pred @ W_out^T) (.) act'(h_{n-1}) from a given dL/dpred (out_dim % 16 == 0)
#define DIB_SMALL_INT_BWD 32       // dgrad chain dL/dh_{n-1} -> ... -> dL/du
#define DIB_SMALL_INT_INFER 64     // no stashes (validation)
#define DIB_SMALL_INT_LOAD_H 128   // hidden activations come from the global stashes (a backward launched on its own)
#define DIB_SMALL_INT_LOAD_G 256   // dL/dh_{n-1} comes from its global stash (written by a separately launched output head)
#define DIB_SMALL_INT_POSENC_IN 512  // the input tile is gather + PositionalEncoding of X rows (a plain MLP: dib_mlp_small_*)
#define DIB_SMALL_INT_NO_GU 1024   // the dgrad chain stops at dL/dh_0 (no gradient with respect to the input)
#define DIB_SMALL_INT_HEAD_REDUCE 2048  // the last workgroup to arrive sums the head's per-tile partials itself (a plain MLP with
                                        // a 1-unit head, dib_mlp_small_head_step: no step tail follows to do it)

struct DibSmallIntArgs {
  const float* U; float* GU; int batch, K0;
  const float* params;
  int n_hidden; int width[4]; long long w_off[4], b_off[4];   // [n_hidden] = the output layer
  float* h[3]; float* g[3];                                   // global stashes int_h / g_int_h, [B][width]
  int act, out_act, out_dim, mode;
  float* pred; float* g_pred;
  int loss_kind; const float* Y; long long ldy; const int* row_idx; long long row0; float inv_bg;
  float* partial_w; float* partial_l;                         // head: [tile][K + 1], [tile][2]
  // DIB_SMALL_INT_POSENC_IN (a plain MLP over its own input, e.g. the InfoNCE path's output encoder, train.py:184-192): the
  // input tile is built from rows row_idx[b] (or row0 + b) of X: [x | sin 2x | sin 4x | ...] (the expressions of
  // dib_posenc_rows_kernel), K0 = in_dim * n_freq, and stashed in a0 [B][K0] (operand of the first weight gradient)
  const float* X; long long ldx; int in_dim, n_freq; float* a0;
  // DIB_SMALL_INT_HEAD_REDUCE: d(W_out | b_out) -> head_gw [K], head_gb [1]; {loss sum, #correct, loss sum * loss_scale} -> sums3;
  // sync: one zero-initialised word (self-cleaning)
  float* head_gw; float* head_gb; float* sums3; float loss_scale; unsigned* sync;
  // cluster mode (dib_small_integration_cluster_kernel): cl workgroups share a row tile, each owning a column slice of every
  // layer; xh[l]: [B][width[l]] exchange buffers of the hidden activations for launches that write no stashes (INFER);
  // cl_sync: DIB_SMALL_CL_SYNC_WORDS zero-initialised words per row tile (self-cleaning)
  int cl; float* xh[3]; unsigned* cl_sync;
  int cl_agent_scope;   // 1: every exchange takes the agent-scope protocol whatever the placement (dib_set_tuning "int_cluster_short_exchange" = 0)
};

// ---- cluster mode: the columns of a row tile's layers on `cl` co-resident workgroups (round 6) ----------------------------------
// At the reference's default batch the integration network is 8 row tiles = 8 workgroups on 256 CUs, and a layer's time on a
// row tile is the stream of the layer's WHOLE weight matrix through one CU plus its MFMA issue there (profiles/HISTORY.md
// sections 13, 19: 8 - 12 us per layer).  In cluster mode workgroup c of a tile's cl computes output columns [slice c) of every
// layer - 1 / cl of the weights and of the MFMAs - writes them to the layer's global buffer (the stash the weight gradients
// read anyway, or xh when inferring) and the cl workgroups exchange slices through L2: one arrival counter per (tile, layer),
// release / acquire at agent scope (relaxed to the XCD's own L2 once the cluster is known to share one: dib_small_cluster_exchange).
// Every sum has a fixed order - replays give the same bits - but not the single-workgroup kernel's order (a slice's contraction is
// split over up to 8 waves): the two differ by fp32 rounding.
// Progress: a workgroup waits only for the cl - 1 others of its own tile, whose ids are consecutive multiples of 8 apart inside one
// block of 8 cl ids; the launch has <= 256 workgroups of one per CU, so every workgroup is resident once the kernels ahead of it on
// other streams drain.  The wait is bounded all the same: after ~2 s of wall clock the kernel traps (the process sees a HIP error
// at its next synchronisation) instead of hanging the device.
#define DIB_SMALL_CL_SYNC_WORDS 32   // per tile: [0, 6) arrival counters (fwd layer l: l; output-layer dgrad: 3; dgrad into h_{l-1}: 3 + l), [6] XCC_ID bits, [7] departures, [8] hello
#define DIB_SMALL_CL_MAX 8

// publish this workgroup's slice, wait for the others'.  The layer primitives end with a workgroup barrier, but a barrier does not
// wait for the waves' global stores (s_waitcnt lgkmcnt(0) only): every wave drains its own (vmcnt(0), the gfx9 encoding 0x0F70)
// ahead of a second barrier, so that thread 0's release covers stores that HAVE reached L2.
// `same` (block-uniform, -1 before the launch's first exchange): 1 when the cluster sits on ONE XCD (every workgroup's XCC_ID is
// collected at kernel start: dib_small_cluster_hello).  The full agent-scope protocol is an L2 write-back before the arrival and
// an L2 + L1 invalidate after the wait: 1.7 us of a 3.2 us exchange on 4 workgroups, 3.9 of 5.3 us on 8
// (profiles/r06r_exchange_variants.txt) - and between CUs that share an L2 it buys nothing:
// the vector L1 is write-through (a store is in L2 when vmcnt says so) and the readers only have to drop their own L1's lines
// (buffer_inv sc0).  A cluster the dispatcher did NOT place on one XCD (the id -> XCD mapping is the hardware's round-robin, not a
// contract) keeps the full protocol for every exchange.
#define DIB_GETREG_XCC_ID ((3 << 11) | 20)   // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4)
// first thing in the kernel (thread 0): this workgroup's XCC_ID into word 6, its arrival into word 8 - no data behind it, no fence;
// by the first exchange (a layer later) the others' have long landed and that exchange already takes the short protocol
__device__ __forceinline__ void dib_small_cluster_hello(unsigned* words) {
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_or(words + 6, 1u << (__builtin_amdgcn_s_getreg(DIB_GETREG_XCC_ID) & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the or has been performed at L2 before the arrival is issued
    __hip_atomic_fetch_add(words + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void dib_small_cluster_wait(const unsigned* word, unsigned target) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > 200000000ll) __builtin_trap();   // 2 s at 100 MHz: never a hang
  }
}

__device__ __forceinline__ void dib_small_cluster_exchange(unsigned* words, int idx, int cl, int& same) {
  __shared__ int s_same;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (same < 0 && threadIdx.x == 0) {   // (block-uniform `same`)
    dib_small_cluster_wait(words + 8, (unsigned)cl);
    const unsigned xccs = __hip_atomic_load(words + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_same = (xccs & (xccs - 1u)) == 0u ? 1 : 0;
  }
  __syncthreads();
  if (same < 0) same = s_same;
  if (threadIdx.x == 0) {
    if (same == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(words + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dib_small_cluster_wait(words + idx, (unsigned)cl);
  }
  __syncthreads();
  if (same == 1) asm volatile("buffer_inv sc0" ::: "memory");   // this CU's L1 only
  else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// the last of the tile's workgroups to leave zeroes the tile's counters for the next launch
__device__ __forceinline__ void dib_small_cluster_leave(unsigned* words, int cl) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(words + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)cl - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// slice c of cl of a width (in 16-column tiles): [c T / cl, (c + 1) T / cl) - empty when the layer has fewer tiles than workgroups
__device__ __forceinline__ void dib_small_cluster_slice(int width, int c, int cl, int& col0, int& ncols) {
  const int T = width >> 4;
  const int t0 = c * T / cl, t1 = (c + 1) * T / cl;
  col0 = 16 * t0; ncols = 16 * (t1 - t0);
}

// CLUSTER: workgroup `crank` of the a.cl that share `tile` (see above); else one workgroup per tile
template <bool CLUSTER>
__device__ __forceinline__ void dib_small_integration_body(const DibSmallIntArgs& a, const int tile, const int crank = 0) {
  constexpr int ROWS = DIB_SMALL_ROWS;
  const int cl = CLUSTER ? a.cl : 1;
  const bool clustered = CLUSTER;   // the slice primitives (a network of the paired grid may be a "cluster" of ONE workgroup per tile)
  const bool multi = CLUSTER && cl > 1;   // ... with somebody to exchange with
  const bool lead = crank == 0;   // writes what every workgroup of the tile computes alike (the head, the encoded input)
  unsigned* const clw = multi ? a.cl_sync + (long long)tile * DIB_SMALL_CL_SYNC_WORDS : nullptr;
  int cl_same = (CLUSTER && a.cl_agent_scope) ? 0 : -1;   // is the cluster on one XCD?  (-1: read at its first exchange; 0: forced)
  if (multi) dib_small_cluster_hello(clw);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = tile * ROWS, rows_valid = min(ROWS, a.batch - r0);
  const int n = a.n_hidden;
  // LDS map: u | h_0 .. h_{n-1} | g_0 .. g_{n-1} | pred / g_pred tile | head scratch.  (n <= 3; every loop over layers is
  // unrolled with a compile-time index so that the pointer / pitch arrays stay in registers)
  const int pu = dib_small_pitch(a.K0);
  float* us = lds;
  float* hs[3]; float* gs[3]; int ph[3];
  float* cur = us + ROWS * pu;
#pragma unroll
  for (int l = 0; l < 3; ++l) { ph[l] = l < n ? dib_small_pitch(a.width[l]) : 0; hs[l] = cur; cur += ROWS * ph[l]; }
#pragma unroll
  for (int l = 0; l < 3; ++l) { gs[l] = cur; cur += ROWS * ph[l]; }
  const int po = dib_small_pitch(a.out_dim);
  float* ps = cur; cur += ROWS * po;
  float* xch = cur; cur += CLUSTER ? DIB_SMALL_XCH_FLOATS_WIDE : DIB_SMALL_XCH_FLOATS;
  float* scratch = cur;   // head: [8][K + 1] + 16
  const bool stash = !(a.mode & DIB_SMALL_INT_INFER);
  const float slope = dib_neg_slope(a.act);
  // the last hidden layer (n is block-uniform: scalar selects)
  const int KL = n == 1 ? a.width[0] : (n == 2 ? a.width[1] : a.width[2]);
  float* const hl = n == 1 ? hs[0] : (n == 2 ? hs[1] : hs[2]);
  float* const gl = n == 1 ? gs[0] : (n == 2 ? gs[1] : gs[2]);
  const int pl = n == 1 ? ph[0] : (n == 2 ? ph[1] : ph[2]);
  float* const g_last = n == 1 ? a.g[0] : (n == 2 ? a.g[1] : a.g[2]);
  const long long wo_off = n == 1 ? a.w_off[1] : (n == 2 ? a.w_off[2] : a.w_off[3]);
  const long long bo_off = n == 1 ? a.b_off[1] : (n == 2 ? a.b_off[2] : a.b_off[3]);

  DIB_ST(16);
  // the head's own inputs - output weights, bias, the rows' labels - are fetched NOW into LDS: their global round trips (two
  // dependent ones for a gathered label) run under the hidden layers instead of in front of every row's loss
  float* const head_w = scratch + 8 * (KL + 1) + 16;   // [KL] output-layer kernel, then [1] bias
  float* const head_y = head_w + KL + 1;               // [16] labels of the tile's rows
  if (a.mode & DIB_SMALL_INT_HEAD) {
    for (int k = tid; k <= KL; k += DIB_SMALL_THREADS) head_w[k] = k < KL ? a.params[wo_off + k] : a.params[bo_off];
    if (tid < ROWS && tid < rows_valid) {
      const int b = r0 + tid;
      const long long grow = a.row_idx ? (long long)a.row_idx[b] : a.row0 + b;
      head_y[tid] = a.Y[grow * a.ldy];
    }
  }
  if (a.mode & DIB_SMALL_INT_FWD) {
    if (a.mode & DIB_SMALL_INT_POSENC_IN) {
      const int d = a.in_dim;
      for (int i = tid; i < ROWS * d; i += DIB_SMALL_THREADS) {
        const int row = i / d, c = i - row * d;
        const bool ok = row < rows_valid;
        float x = 0.f;
        if (ok) {
          const int b = r0 + row;
          const long long grow = a.row_idx ? (long long)a.row_idx[b] : a.row0 + b;
          x = a.X[grow * a.ldx + c];
        }
        float* dst = us + row * pu + c;
        const bool keep = ok && a.a0 != nullptr && lead;
        float* gd = a.a0 + (long long)(r0 + row) * a.K0 + c;
        dst[0] = x;
        if (keep) gd[0] = x;
        float fr = 2.0f;
        for (int jb = 1; jb < a.n_freq; ++jb) {
          const float v = ok ? sinf(fr * x) : 0.f;
          dst[jb * d] = v;
          if (keep) gd[jb * d] = v;
          fr *= 2.0f;
        }
      }
    } else {
      dib_small_load_tile(a.U + (long long)r0 * a.K0, a.K0, a.K0, rows_valid, us, pu);
    }
    __syncthreads();
    DIB_ST(17);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      if (l < n) {
        const float* in = l == 0 ? us : hs[l > 0 ? l - 1 : 0];
        const int K = l == 0 ? a.K0 : a.width[l > 0 ? l - 1 : 0], pin = l == 0 ? pu : ph[l > 0 ? l - 1 : 0];
        if (clustered) {
          // (one workgroup per tile and no stash: nothing leaves the CU)
          float* const gx = (stash || multi) ? (stash ? a.h[l] : a.xh[l]) + (long long)r0 * a.width[l] : nullptr;
          int col0, ncols;
          dib_small_cluster_slice(a.width[l], crank, cl, col0, ncols);
          dib_small_fwd_cols(in, pin, K, a.params + a.w_off[l], a.width[l], col0, ncols, a.params + a.b_off[l], slope, hs[l], ph[l],
                             gx, rows_valid, xch, l == 0 ? 6 : 45);
          DIB_ST(30 + l);
          if (multi) {
            dib_small_cluster_exchange(clw, l, cl, cl_same);
            dib_small_load_tile(gx, a.width[l], a.width[l], rows_valid, hs[l], ph[l]);
            __syncthreads();
          }
        } else {
          dib_small_fwd(in, pin, K, K, a.params + a.w_off[l], a.width[l], a.params + a.b_off[l], slope, hs[l], ph[l],
                        stash ? a.h[l] + (long long)r0 * a.width[l] : nullptr, a.width[l], rows_valid, xch);
        }
        DIB_ST(18 + l);
      }
    }
  } else if (a.mode & DIB_SMALL_INT_LOAD_H) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
      if (l < n) dib_small_load_tile(a.h[l] + (long long)r0 * a.width[l], a.width[l], a.width[l], rows_valid, hs[l], ph[l]);
    if (a.mode & DIB_SMALL_INT_LOAD_G) dib_small_load_tile(g_last + (long long)r0 * KL, KL, KL, rows_valid, gl, pl);
    __syncthreads();
  }

  if (a.mode & DIB_SMALL_INT_OUT) {   // general output layer (reference models.py:83)
    if (clustered) {   // its columns go straight to pred: nothing to exchange
      int col0, ncols;
      dib_small_cluster_slice(a.out_dim, crank, cl, col0, ncols);
      dib_small_fwd_cols(hl, pl, KL, a.params + wo_off, a.out_dim, col0, ncols, a.params + bo_off, dib_neg_slope(a.out_act), nullptr, 0,
                         a.pred + (long long)r0 * a.out_dim, rows_valid, xch);
    } else {
      dib_small_fwd(hl, pl, KL, KL, a.params + wo_off, a.out_dim, a.params + bo_off, dib_neg_slope(a.out_act), nullptr, 0,
                    a.pred + (long long)r0 * a.out_dim, a.out_dim, rows_valid, xch);
    }
  }

  if (a.mode & DIB_SMALL_INT_HEAD) {
    // z = h . w + b per row (wave w: rows w, w + 8); Keras BinaryCrossentropy(from_logits=True) / 'mse', the
    // expressions of dib_head_fused_kernel; dL/dh = g w (.) act'(h); per-tile partial of d(w|b) and of {loss sum, #correct}
    const bool grad = (a.mode & DIB_SMALL_INT_HEAD_GRAD) != 0;
    const float* wv = head_w;            // LDS copies (written before the hidden layers, whose barriers publish them)
    const float b0 = head_w[KL];
    float* redw = scratch;               // [8][KL + 1]
    float* redl = scratch + 8 * (KL + 1); // [8][2]
    float lsum = 0.f, correct = 0.f, pb = 0.f;
    float pw[16];                        // KL <= 1024: 16 lane-strided columns
#pragma unroll
    for (int c = 0; c < 16; ++c) pw[c] = 0.f;
    // rows w and w + 8 of the tile TOGETHER (the two rows' chains - LDS reads, the wave sum, exp / log1p, the stores - are each a
    // string of latencies; side by side they overlap: head 4.9 -> see profiles/r06t_*).  Same operations in the same order per row.
    float z2[2], gg2[2], yy2[2];
    bool ok2[2];
    {
      float dot[2] = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int k = lane + 64 * c;
        if (k < KL) {
          const float w = wv[k];
#pragma unroll
          for (int r = 0; r < 2; ++r) dot[r] += hl[(wave + 8 * r) * pl + k] * w;
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        ok2[r] = wave + 8 * r < rows_valid;   // wave-uniform
        z2[r] = dib_wave_sum(dot[r]) + b0;
        yy2[r] = head_y[wave + 8 * r];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float z = z2[r], yy = yy2[r];
      float l, gg;
      if (a.loss_kind == 0) {
        l = fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
        gg = 1.0f / (1.0f + expf(-z)) - yy;
      } else {
        const float dd = z - yy;
        l = dd * dd;
        gg = 2.f * dd;
      }
      gg *= a.inv_bg;
      gg2[r] = gg;
      if (lane == 0 && ok2[r]) {
        const int b = r0 + wave + 8 * r;
        if (lead) {
          a.pred[b] = z;
          if (grad) a.g_pred[b] = gg;
        }
        lsum += l;
        correct += ((z > 0.5f ? 1.f : 0.f) == yy) ? 1.f : 0.f;
        pb += gg;
      }
    }
    if (grad) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int k = lane + 64 * c;
        if (k < KL) {
          const float w = wv[k];
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (ok2[r]) {
              const int row = wave + 8 * r;
              const float hv = hl[row * pl + k];
              const float gv = gg2[r] * w * dib_small_act_grad(slope, hv);
              gl[row * pl + k] = gv;
              if (lead) g_last[(long long)(r0 + row) * KL + k] = gv;
              pw[c] += hv * gg2[r];
            }
        }
      }
    }
    if (grad) {
      // rows of the tile that do not exist carry no gradient into the dgrad chain
      for (int row = rows_valid + wave; row < ROWS; row += 8)
        for (int k = lane; k < KL; k += 64) gl[row * pl + k] = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int k = lane + 64 * c;
        if (k < KL) redw[wave * (KL + 1) + k] = pw[c];
      }
    }
    if (lane == 0) { redw[wave * (KL + 1) + KL] = pb; redl[2 * wave] = lsum; redl[2 * wave + 1] = correct; }
    __syncthreads();
    if (grad && lead) {
      float* dst = a.partial_w + (long long)tile * (KL + 1);
      for (int i = tid; i <= KL; i += DIB_SMALL_THREADS) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += redw[w * (KL + 1) + i];
        dst[i] = t;
      }
    }
    if (tid < 2 && lead) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += redl[2 * w + tid];
      a.partial_l[2 * tile + tid] = t;
    }
  }

  DIB_ST(22);
  if (a.mode & DIB_SMALL_INT_BWD_OUT) {   // dL/dh_{n-1} from a given dL/dpred (custom loss: InfoNCE, train.py:216-219)
    dib_small_load_tile(a.g_pred + (long long)r0 * a.out_dim, a.out_dim, a.out_dim, rows_valid, ps, po);
    __syncthreads();
    if (clustered) {
      int k0, kc;
      dib_small_cluster_slice(KL, crank, cl, k0, kc);
      dib_small_bwd_cols(ps, po, a.out_dim, a.params + wo_off, KL, k0, kc, hl, pl, slope, gl, pl, g_last + (long long)r0 * KL, rows_valid, xch);
      if (multi) {
        dib_small_cluster_exchange(clw, 3, cl, cl_same);
        dib_small_load_tile(g_last + (long long)r0 * KL, KL, KL, rows_valid, gl, pl);
        __syncthreads();
      }
    } else {
      dib_small_bwd(ps, po, a.out_dim, a.params + wo_off, KL, hl, pl, slope, gl, pl, g_last + (long long)r0 * KL, KL, rows_valid, xch);
    }
  }

  if (a.mode & DIB_SMALL_INT_BWD) {
    DIB_ST(23);
#pragma unroll
    for (int l = 2; l >= 1; --l) {   // dL/dh_{l-1} = (dL/dh_l @ W_l^T) (.) act'(h_{l-1})
      if (l < n) {
        if (clustered) {
          float* const gx = a.g[l - 1] + (long long)r0 * a.width[l - 1];
          int k0, kc;
          dib_small_cluster_slice(a.width[l - 1], crank, cl, k0, kc);
          dib_small_bwd_cols(gs[l], ph[l], a.width[l], a.params + a.w_off[l], a.width[l - 1], k0, kc, hs[l - 1], ph[l - 1], slope,
                             gs[l - 1], ph[l - 1], gx, rows_valid, xch, 34);
          DIB_ST(32 + l);
          if (multi) {
            dib_small_cluster_exchange(clw, 3 + l, cl, cl_same);
            dib_small_load_tile(gx, a.width[l - 1], a.width[l - 1], rows_valid, gs[l - 1], ph[l - 1]);
            __syncthreads();
          }
        } else {
          dib_small_bwd(gs[l], ph[l], a.width[l], a.params + a.w_off[l], a.width[l - 1], hs[l - 1], ph[l - 1], slope, gs[l - 1],
                        ph[l - 1], a.g[l - 1] + (long long)r0 * a.width[l - 1], a.width[l - 1], rows_valid, xch);
        }
        DIB_ST(24 + l);
      }
    }
    // dL/du = dL/dh_0 @ W_0^T   (u is not an activation output; a plain MLP's input needs no gradient)
    if (!(a.mode & DIB_SMALL_INT_NO_GU)) {
      if (clustered) {
        int k0, kc;
        dib_small_cluster_slice(a.K0, crank, cl, k0, kc);
        dib_small_bwd_cols(gs[0], ph[0], a.width[0], a.params + a.w_off[0], a.K0, k0, kc, nullptr, 0, 1.f, nullptr, 0,
                           a.GU + (long long)r0 * a.K0, rows_valid, xch, 51);
      } else {
        dib_small_bwd(gs[0], ph[0], a.width[0], a.params + a.w_off[0], a.K0, nullptr, 0, 1.f, nullptr, 0,
                      a.GU + (long long)r0 * a.K0, a.K0, rows_valid, xch);
      }
    }
    DIB_ST(28);
  }
  if (multi) dib_small_cluster_leave(clw, cl);

  if (a.mode & DIB_SMALL_INT_HEAD_REDUCE) {   // block-uniform
    // the per-tile partials of the output layer's gradient and of the loss sums, summed in tile order by the last workgroup
    // to arrive (deterministic; the release / acquire pair of csrc/dib_st_chain.h)
    __shared__ bool s_last;
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const bool last = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (last) __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int ntl = (int)gridDim.x;
    for (int i = tid; i <= KL; i += DIB_SMALL_THREADS) {
      float t = 0.f;
      for (int tl = 0; tl < ntl; ++tl) t += a.partial_w[(long long)tl * (KL + 1) + i];
      if (i < KL) a.head_gw[i] = t; else a.head_gb[0] = t;
    }
    if (tid < 2) {
      float t = 0.f;
      for (int tl = 0; tl < ntl; ++tl) t += a.partial_l[2 * tl + tid];
      a.sums3[tid] = t;
      if (tid == 0) a.sums3[2] = t * a.loss_scale;
    }
  }
}

__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_integration_kernel(DibSmallIntArgs a) { dib_small_integration_body<false>(a, blockIdx.x); }

// cluster mode: grid = 8 ceil(tiles / 8) x cl workgroups; id -> (tile, rank) keeps a tile's workgroups on ONE XCD under the
// round-robin placement (XCD = id mod 8): tile = 8 (id / (8 cl)) + id mod 8, rank = (id / 8) mod cl - their exchange then stays
// in that XCD's L2.  (Placement is a performance matter only: the exchange is agent-scope.)
__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_integration_cluster_kernel(DibSmallIntArgs a) {
  const int id = blockIdx.x, cl = a.cl;
  const int tile = 8 * (id / (8 * cl)) + (id & 7), crank = (id >> 3) % cl;
  if (tile * DIB_SMALL_ROWS >= a.batch) return;
  dib_small_integration_body<true>(a, tile, crank);
}

// Two independent networks in ONE grid (blockIdx.y picks the argument set): the custom InfoNCE loop's X model and its output
// encoder between the encoder bank and the loss (train.py:203-219) - each is 8 workgroups at the reference's batch of 128, and
// a launch of its own costs more than its work.  The argument sets stay in the kernarg segment (uniform index: scalar loads).
struct DibSmallIntPair { DibSmallIntArgs s[2]; };
__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_integration_pair_kernel(DibSmallIntPair p) {
  const DibSmallIntArgs& a = p.s[blockIdx.y];
  if ((int)blockIdx.x * DIB_SMALL_ROWS >= a.batch) return;   // the two batches may differ
  dib_small_integration_body<false>(a, blockIdx.x);
}

// ... and the paired grid in cluster mode: each network with its own cluster size (s[i].cl; 1 = one workgroup per tile on the slice
// primitives - no exchange, another fp32 summation order than dib_small_integration_pair_kernel); gridDim.x = 8 ceil(tiles / 8) x
// the larger of the two
__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_integration_pair_cluster_kernel(DibSmallIntPair p) {
  const DibSmallIntArgs& a = p.s[blockIdx.y];
  const int id = blockIdx.x, cl = a.cl;
  const int tile = 8 * (id / (8 * cl)) + (id & 7), crank = (id >> 3) % cl;
  if (tile * DIB_SMALL_ROWS >= a.batch) return;
  dib_small_integration_body<true>(a, tile, crank);
}

// =====================================================================================================================
// encoder bank backward (dgrad chain + the layer-1 weight-gradient partial)
// =====================================================================================================================
struct DibSmallEncBwdArgs {
  const float* P; int batch;
  const float* params; const long long* w_off; const long long* b_off; const int4* featmap;
  int act, F, E, H1, H2;
  const float* h1; const float* h2; const float* enc_out; const float* U; const float* GU;
  float* dout; float* dh2;
  float* dw1_partial;   // [row tile][F][16][H1]
  const float* beta_dev; float inv_bg;
};

__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_small_encoder_bwd_kernel(DibSmallEncBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = tile * DIB_SMALL_ROWS, rows_valid = min(DIB_SMALL_ROWS, a.batch - r0);
  const int4 fm = a.featmap[f];
  const int in_dim = fm.y, F = a.F, E = a.E, E2 = 2 * a.E;
  const int p1 = dib_small_pitch(a.H1), p2 = dib_small_pitch(a.H2), p3 = dib_small_pitch(E2);
  float* Pl = lds;                            // [16][20]: [P | 1 | 0], rows >= rows_valid zero
  float* h1s = Pl + DIB_SMALL_ROWS * 20;      // [16][p1]
  float* h2s = h1s + DIB_SMALL_ROWS * p1;     // [16][p2]
  float* dos = h2s + DIB_SMALL_ROWS * p2;     // [16][p3]  d(mu|logvar)
  float* dh2s = dos + DIB_SMALL_ROWS * p3;    // [16][p2]
  float* dh1s = dh2s + DIB_SMALL_ROWS * p2;   // [16][p1]
  float* xch = dh1s + DIB_SMALL_ROWS * p1;    // wave-pair exchange
  const long long frow = (long long)f * a.batch + r0;
  DIB_ST(40);
  dib_small_load_tile(a.h1 + frow * a.H1, a.H1, a.H1, rows_valid, h1s, p1);
  dib_small_load_tile(a.h2 + frow * a.H2, a.H2, a.H2, rows_valid, h2s, p2);
  for (int i = tid; i < DIB_SMALL_ROWS * 20; i += DIB_SMALL_THREADS) {
    const int row = i / 20, c = i - row * 20;
    float v = 0.f;
    if (row < rows_valid) {
      if (c < in_dim) v = a.P[(long long)fm.w * a.batch + (long long)(r0 + row) * in_dim + c];
      else if (c == in_dim) v = 1.f;
    }
    Pl[i] = v;
  }
  // ---- d(loss + beta KL)/d(mu|logvar) (the expressions of dib_fused_encoder_bwd_kernel); eps sigma = u - mu ----
  const float kb = a.beta_dev[0] * a.inv_bg;
  const int E4 = E >> 2;
  for (int i = tid; i < DIB_SMALL_ROWS * E4; i += DIB_SMALL_THREADS) {
    const int row = i / E4, qq = i - row * E4;
    float4 dm = make_float4(0.f, 0.f, 0.f, 0.f), dl = dm;
    if (row < rows_valid) {
      const int b = r0 + row;
      const float* eo = a.enc_out + ((long long)f * a.batch + b) * E2;
      const float4 mu = *reinterpret_cast<const float4*>(eo + 4 * qq);
      const float4 lv = *reinterpret_cast<const float4*>(eo + E + 4 * qq);
      const long long so = (long long)b * ((long long)F * E) + (long long)f * E + 4 * qq;
      const float4 g = *reinterpret_cast<const float4*>(a.GU + so);
      const float4 u = *reinterpret_cast<const float4*>(a.U + so);
      const float sx = dib_sigma(lv.x), sy = dib_sigma(lv.y), sz = dib_sigma(lv.z), sw = dib_sigma(lv.w);
      dm = make_float4(g.x + kb * mu.x, g.y + kb * mu.y, g.z + kb * mu.z, g.w + kb * mu.w);
      dl = make_float4(g.x * (u.x - mu.x) * 0.5f + kb * 0.5f * (sx * sx - 1.f), g.y * (u.y - mu.y) * 0.5f + kb * 0.5f * (sy * sy - 1.f),
                       g.z * (u.z - mu.z) * 0.5f + kb * 0.5f * (sz * sz - 1.f), g.w * (u.w - mu.w) * 0.5f + kb * 0.5f * (sw * sw - 1.f));
      float* dd = a.dout + ((long long)f * a.batch + b) * E2;
      *reinterpret_cast<float4*>(dd + 4 * qq) = dm;
      *reinterpret_cast<float4*>(dd + E + 4 * qq) = dl;
    }
    *reinterpret_cast<float4*>(dos + row * p3 + 4 * qq) = dm;
    *reinterpret_cast<float4*>(dos + row * p3 + E + 4 * qq) = dl;
  }
  __syncthreads();
  DIB_ST(41);
  const float* W2 = a.params + a.w_off[1 * F + f];
  const float* W3 = a.params + a.w_off[2 * F + f];
  // dh2 = (dout @ W3^T) (.) act'(h2) -> stash (operand of the layer-2 weight gradient) ; dh1 = (dh2 @ W2^T) (.) act'(h1)
  const float slope = dib_neg_slope(a.act);
  dib_small_bwd_cols(dos, p3, E2, W3, a.H2, 0, a.H2, h2s, p2, slope, dh2s, p2, a.dh2 + frow * a.H2, rows_valid, xch);
  DIB_ST(42);
  dib_small_bwd_cols(dh2s, p2, a.H2, W2, a.H1, 0, a.H1, h1s, p1, slope, dh1s, p1, nullptr, rows_valid, xch);
  DIB_ST(43);
  // d(W1|b1) partial of the tile = [P | 1]^T @ dh1: 16 x 16 output tiles (rows = encoder-input index, row in_dim = bias),
  // contraction over the 16 rows in 4 MFMA steps; lane (i, q): A[i][row 4 s + q] = Pl[row][i], B[row][n0 + j] = dh1[row][n0 + j]
  {
    const int j = lane & 15, q = lane >> 4;
    float* dst = a.dw1_partial + ((long long)tile * F + f) * (16ll * a.H1);
    for (int n0 = 16 * wave; n0 < a.H1; n0 += 128) {
      dib_f32x4 acc = dib_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = DIB_MFMA16(Pl[(4 * s + q) * 20 + j], dh1s[(4 * s + q) * p1 + n0 + j], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(long long)(4 * q + r) * a.H1 + n0 + j] = acc[r];
    }
  }
  DIB_ST(44);
}
