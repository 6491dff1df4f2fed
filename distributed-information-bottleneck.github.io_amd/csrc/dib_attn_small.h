// dib_attn_small.h - self-attention of ONE (neighbourhood, head) per workgroup for neighbourhoods of at most 64 particles.
//
// The reference's own configuration of the per-particle set transformer is 32 neighbourhoods x 50 particles
// (...per_particle_measurements_and_set_transformer.ipynb:304-307; MultiHeadAttention(12 heads, key_dim 128)(x, x, x), cell 8).
// The flash kernels of dib_attn.h are built for thousands of particles: a workgroup owns 128 queries (forward) or 128 keys
// (backward) of one head with one wave per SIMD, streams the other side through LDS in tiles of 32 and pays a prologue (V row
// fragments, the K block) and an epilogue per workgroup that 4096 particles amortise and 50 do not - at the notebook's size
// they were 24 us (forward) and 54 + 6.5 us (backward + delta) per attention block, 0.48 ms of the 1.75 ms step
// (profiles/r04c_set_transformer_notebook_size_kernel_stats.csv).  Here all of q, k, v (dO) of the head sit in LDS at once:
//
//   forward   S = (scale Q) K^T [64 x 64] -> LDS -> row softmax in place (4 threads per query) -> O = P V; lse written
//   backward  S, dP = dO V^T -> P = exp(S - lse), delta = rowsum(P o dP) (no separate delta kernel, no O), dS = P o (dP - delta)
//             -> LDS; dV = P^T dO, dK = dS^T (scale Q), dQ = scale dS K; no partial buffers, no score stash
//
// 4 waves, every product on v_mfma_f32_32x32x2_f32 with 32 x 32 output tiles; operand fetches follow dib_attn.h ("KC": 4
// consecutive k of a row by one ds_read_b128; "MC": rows 8q + 4h + t of a column).  Rows beyond P are zero in LDS, keys beyond
// P are masked in the softmax, queries beyond P are never stored.  Exact fp32, deterministic.
#pragma once
#include "dib_attn.h"
#include "dib_infonce_mfma.h"   // dib_half_sum (DPP + swizzle reduction over a half-wave)

constexpr int kAttnSmallP = 64;                         // largest neighbourhood this path takes
constexpr int kAttnSP = kAttnSmallP + 4;                // pitch of the [64][64] score / probability tiles
constexpr int DibAttnSmallFwdLds = 2 * kAttnSmallP * kAttnPitch;                                       // Q|V, K|P
constexpr int DibAttnSmallBwdLds = 3 * kAttnSmallP * kAttnPitch + 2 * kAttnSmallP * kAttnSP + 4 * kAttnSmallP;

// 64 rows x 128 floats of head `head` (rows >= P zero) -> LDS tile [64][132], optionally scaled
__device__ __forceinline__ void dib_attn_small_load(float* __restrict__ T, const float* __restrict__ base, long long ld, int P,
                                                    int tid, float mul) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (tid >> 5) + 8 * p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < P) v = *reinterpret_cast<const float4*>(base + (long long)r * ld + (tid & 31) * 4);
    *reinterpret_cast<float4*>(T + r * kAttnPitch + (tid & 31) * 4) = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
  }
}

// C fragment (32 x 32, rows (r & 3) + 8 (r >> 2) + 4 h, column l31) of a [64 rows][128] result -> global rows < P
__device__ __forceinline__ void dib_attn_small_store(float* __restrict__ base, long long ld, int row0, int col, int P, int h,
                                                     const dib_f32x16& acc, float mul) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < P) base[(long long)row * ld + col] = acc[r] * mul;
  }
}

// grid (H, B), 256 threads, dynamic LDS DibAttnSmallFwdLds floats
__global__ void __launch_bounds__(256)
dib_attn_small_fwd_kernel(DibAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                               // scaled Q, then V
  float* Ks = Qs + kAttnSmallP * kAttnPitch;     // K, then the score / probability tile [64][68]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int head = blockIdx.x, b = blockIdx.y, P = a.P;
  const long long tok0 = (long long)b * P;
  const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
  const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
  const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
  dib_attn_small_load(Qs, Qb, a.ld, P, tid, a.scale);
  dib_attn_small_load(Ks, Kb, a.ld, P, tid, 1.0f);
  float4 vr[8];                                  // V rows: in flight during the S product, into Q's space after it
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (tid >> 5) + 8 * p;
    vr[p] = r < P ? *reinterpret_cast<const float4*>(Vb + (long long)r * a.ld + (tid & 31) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // ---- S[query wm*32.., key wn*32..] ----
  dib_f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
  const float* Qw = Qs + wm * 32 * kAttnPitch;
  const float* Kw = Ks + wn * 32 * kAttnPitch;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float4 qq = dib_attn_kc(Qw, q, l31, h), kk = dib_attn_kc(Kw, q, l31, h);
    s = DIB_MFMA(qq.x, kk.x, s);
    s = DIB_MFMA(qq.y, kk.y, s);
    s = DIB_MFMA(qq.z, kk.z, s);
    s = DIB_MFMA(qq.w, kk.w, s);
  }
  __syncthreads();                               // every wave is done reading Q and K
  float* St = Ks;
#pragma unroll
  for (int r = 0; r < 16; ++r) St[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kAttnSP + wn * 32 + l31] = s[r];
#pragma unroll
  for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(Qs + ((tid >> 5) + 8 * p) * kAttnPitch + (tid & 31) * 4) = vr[p];
  __syncthreads();
  // ---- row softmax in place: thread = (query tid >> 2, keys 16 (tid & 3) ..) ----
  {
    const int qi = tid >> 2, k0 = (tid & 3) * 16;
    float* row = St + qi * kAttnSP + k0;
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      v[j] = (k0 + j < P) ? row[j] : -INFINITY;
      m = fmaxf(m, v[j]);
    }
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));          // key 0 < P always: m is finite
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] = expf(v[j] - m); sum += v[j]; }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j] = v[j] * inv;
    if ((tid & 3) == 0 && qi < P) a.lse[((long long)b * a.H + head) * P + qi] = m + logf(sum);
  }
  __syncthreads();
  // ---- O[query wm*32.., d wn*64..] = P V ----
  dib_f32x16 o[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 pp = *reinterpret_cast<const float4*>(St + (wm * 32 + l31) * kAttnSP + q * 8 + h * 4);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float4 vv = dib_attn_mc(Qs, q, wn * 64 + n * 32 + l31, h);
      o[n] = DIB_MFMA(pp.x, vv.x, o[n]);
      o[n] = DIB_MFMA(pp.y, vv.y, o[n]);
      o[n] = DIB_MFMA(pp.z, vv.z, o[n]);
      o[n] = DIB_MFMA(pp.w, vv.w, o[n]);
    }
  }
  float* Ob = a.o + tok0 * a.ld + head * kAttnD;
#pragma unroll
  for (int n = 0; n < 2; ++n) dib_attn_small_store(Ob, a.ld, wm * 32, wn * 64 + n * 32 + l31, P, h, o[n], 1.0f);
}

// grid (H, B), 256 threads, dynamic LDS DibAttnSmallBwdLds floats
__global__ void __launch_bounds__(256)
dib_attn_small_bwd_kernel(DibAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                               // scaled Q
  float* Ks = Qs + kAttnSmallP * kAttnPitch;
  float* Gs = Ks + kAttnSmallP * kAttnPitch;     // dO
  float* Pt = Gs + kAttnSmallP * kAttnPitch;     // [64 queries][68] probabilities
  float* dSt = Pt + kAttnSmallP * kAttnSP;       // [64 queries][68] dS
  float* Ls = dSt + kAttnSmallP * kAttnSP;       // lse of the 64 queries (+inf beyond P)
  float* Dp = Ls + kAttnSmallP;                  // [2][64] partial delta (keys 0..31 | 32..63)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int head = blockIdx.x, b = blockIdx.y, P = a.P;
  const long long tok0 = (long long)b * P;
  const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
  const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
  const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
  const float* dOb = a.d_o + tok0 * a.ld + head * kAttnD;
  dib_attn_small_load(Qs, Qb, a.ld, P, tid, a.scale);
  dib_attn_small_load(Ks, Kb, a.ld, P, tid, 1.0f);
  dib_attn_small_load(Gs, dOb, a.ld, P, tid, 1.0f);
  if (tid < kAttnSmallP) Ls[tid] = tid < P ? a.lse[((long long)b * a.H + head) * P + tid] : INFINITY;
  // V row of this lane's key (B operand of dP = dO V^T), resident in registers
  const int key = wn * 32 + l31;
  float4 vf[16];
  dib_attn_rowfrag(vf, Vb, a.ld, min(key, P - 1), h, key < P ? 1.0f : 0.0f);
  __syncthreads();
  // ---- S and dP [query wm*32.., key wn*32..] ----
  dib_f32x16 s, dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
  {
    const float* Qw = Qs + wm * 32 * kAttnPitch;
    const float* Gw = Gs + wm * 32 * kAttnPitch;
    const float* Kw = Ks + wn * 32 * kAttnPitch;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 qq = dib_attn_kc(Qw, q, l31, h), gg = dib_attn_kc(Gw, q, l31, h), kk = dib_attn_kc(Kw, q, l31, h);
      s = DIB_MFMA(qq.x, kk.x, s);
      dp = DIB_MFMA(gg.x, vf[q].x, dp);
      s = DIB_MFMA(qq.y, kk.y, s);
      dp = DIB_MFMA(gg.y, vf[q].y, dp);
      s = DIB_MFMA(qq.z, kk.z, s);
      dp = DIB_MFMA(gg.z, vf[q].z, dp);
      s = DIB_MFMA(qq.w, kk.w, s);
      dp = DIB_MFMA(gg.w, vf[q].w, dp);
    }
  }
  // ---- P, delta, dS ----
  const float kmul = key < P ? 1.0f : 0.0f;
  float pv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    pv[r] = expf(s[r] - Ls[qi]) * kmul;          // query beyond P: lse = +inf -> 0
    const float part = dib_half_sum(pv[r] * dp[r]);   // over this wave's 32 keys (dib_infonce_mfma.h helpers)
    if (l31 == 0) Dp[wn * kAttnSmallP + qi] = part;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const float delta = Dp[qi] + Dp[kAttnSmallP + qi];
    Pt[qi * kAttnSP + key] = pv[r];
    dSt[qi * kAttnSP + key] = pv[r] * (dp[r] - delta);
  }
  __syncthreads();
  // ---- dV[key wm*32.., d wn*64..] = P^T dO,  dK = dS^T (scale Q): contraction over the 64 queries ----
  dib_f32x16 dv[2], dk[2], dq[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[n][r] = 0.f; dk[n][r] = 0.f; dq[n][r] = 0.f; }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float* pp = Pt + (q * 8 + h * 4) * kAttnSP + wm * 32 + l31;
    const float* dd = dSt + (q * 8 + h * 4) * kAttnSP + wm * 32 + l31;
    const float p0 = pp[0], p1 = pp[kAttnSP], p2 = pp[2 * kAttnSP], p3 = pp[3 * kAttnSP];
    const float d0 = dd[0], d1 = dd[kAttnSP], d2 = dd[2 * kAttnSP], d3 = dd[3 * kAttnSP];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float4 gg = dib_attn_mc(Gs, q, wn * 64 + n * 32 + l31, h), qq = dib_attn_mc(Qs, q, wn * 64 + n * 32 + l31, h);
      dv[n] = DIB_MFMA(p0, gg.x, dv[n]);
      dk[n] = DIB_MFMA(d0, qq.x, dk[n]);
      dv[n] = DIB_MFMA(p1, gg.y, dv[n]);
      dk[n] = DIB_MFMA(d1, qq.y, dk[n]);
      dv[n] = DIB_MFMA(p2, gg.z, dv[n]);
      dk[n] = DIB_MFMA(d2, qq.z, dk[n]);
      dv[n] = DIB_MFMA(p3, gg.w, dv[n]);
      dk[n] = DIB_MFMA(d3, qq.w, dk[n]);
    }
  }
  // ---- dQ[query wm*32.., d wn*64..] = scale dS K: contraction over the 64 keys ----
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 ds = *reinterpret_cast<const float4*>(dSt + (wm * 32 + l31) * kAttnSP + q * 8 + h * 4);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float4 kk = dib_attn_mc(Ks, q, wn * 64 + n * 32 + l31, h);
      dq[n] = DIB_MFMA(ds.x, kk.x, dq[n]);
      dq[n] = DIB_MFMA(ds.y, kk.y, dq[n]);
      dq[n] = DIB_MFMA(ds.z, kk.z, dq[n]);
      dq[n] = DIB_MFMA(ds.w, kk.w, dq[n]);
    }
  }
  float* dQb = a.dq + tok0 * a.ld + head * kAttnD;
  float* dKb = a.dk + tok0 * a.ld + head * kAttnD;
  float* dVb = a.dv + tok0 * a.ld + head * kAttnD;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = wn * 64 + n * 32 + l31;
    dib_attn_small_store(dVb, a.ld, wm * 32, col, P, h, dv[n], 1.0f);
    dib_attn_small_store(dKb, a.ld, wm * 32, col, P, h, dk[n], 1.0f);
    dib_attn_small_store(dQb, a.ld, wm * 32, col, P, h, dq[n], a.scale);
  }
}
