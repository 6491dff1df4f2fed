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
// PROJ (round 6): the head's q, k, v are not read but COMPUTED here - MultiHeadAttention's three input Dense layers on the
// neighbourhood's tokens, x [P, 32] @ W_i[:, head's 128 columns] + b_i - and written out for the backward: the projection launch
// in front of every attention block (14 us of 57 per block at the notebook's size) disappears.  Wave (wm, wn) owns rows
// [32 wm, +32) x columns [64 wn, +64) of each of the three tiles; its A operand (the tokens' 32 values, 16 registers) and the
// weight fragments (4-byte loads, 128-byte runs per k) come straight from global memory / L2 - no extra LDS, same occupancy.
template <bool PROJ>
__global__ void __launch_bounds__(256)
dib_attn_small_fwd_kernel(DibAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                               // scaled Q, then V
  float* Ks = Qs + kAttnSmallP * kAttnPitch;     // K, then the score / probability tile [64][68]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int head = blockIdx.x, b = blockIdx.y, P = a.P;
  const long long tok0 = (long long)b * P;
  float4 vr[8];                                  // V rows: in flight during the S product, into Q's space after it
  dib_f32x16 vacc[2];                            // PROJ: the wave's 32 x 64 piece of V in accumulator layout instead
  if constexpr (!PROJ) {
    const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
    const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
    const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
    dib_attn_small_load(Qs, Qb, a.ld, P, tid, a.scale);
    dib_attn_small_load(Ks, Kb, a.ld, P, tid, 1.0f);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int r = (tid >> 5) + 8 * p;
      vr[p] = r < P ? *reinterpret_cast<const float4*>(Vb + (long long)r * a.ld + (tid & 31) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    // A fragments: x[row 32 wm + l31][k = 8 q + 4 h + t], q = 0..3 (rows >= P: zero)
    const int xrow = wm * 32 + l31;
    float4 xa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      xa[q] = xrow < P ? *reinterpret_cast<const float4*>(a.px + (tok0 + xrow) * a.pldx + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int col0 = head * kAttnD + wn * 64 + l31;          // this lane's column of tile j = 0 (tile 1: + 32)
    // ALL weight fragments of the three projections are requested before the first MFMA (96 registers; the workgroup is
    // LDS-limited to two per CU anyway): one L2 round trip in front of the prologue instead of three
    float bw[3][2][16];                                       // W_i[k = 8 q + 4 h + t][col0 + 32 j], index 4 q + t
    float bb[3][2];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
      const float* W = a.pparams + a.pw[pi];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) bw[pi][j][4 * q + t] = W[(long long)(8 * q + 4 * h + t) * a.ld + col0 + 32 * j];
        bb[pi][j] = a.pparams[a.pb[pi] + col0 + 32 * j];
      }
    }
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
      dib_f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = bb[pi][0]; acc[1][r] = bb[pi][1]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[j] = DIB_MFMA(xa[q].x, bw[pi][j][4 * q + 0], acc[j]);
          acc[j] = DIB_MFMA(xa[q].y, bw[pi][j][4 * q + 1], acc[j]);
          acc[j] = DIB_MFMA(xa[q].z, bw[pi][j][4 * q + 2], acc[j]);
          acc[j] = DIB_MFMA(xa[q].w, bw[pi][j][4 * q + 3], acc[j]);
        }
      }
      // C fragment: column l31 of tile j, rows (r & 3) + 8 (r >> 2) + 4 h of the wave's 32: to global (rows < P) and to LDS
      float* Gout = (pi == 0 ? a.pq : (pi == 1 ? a.pk : a.pv)) + tok0 * a.ld;
      float* Ts = pi == 0 ? Qs : Ks;
      const float mul = pi == 0 ? a.scale : 1.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = row < P;
          if (ok) Gout[(long long)row * a.ld + col0 + 32 * j] = acc[j][r];
          if (pi < 2) Ts[row * kAttnPitch + wn * 64 + 32 * j + l31] = ok ? acc[j][r] * mul : 0.f;
        }
        if (pi == 2) vacc[j] = acc[j];
      }
    }
  }
  __syncthreads();
  // ---- S[query wm*32.., key wn*32..] ----
  dib_f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
  const float* Qw = Qs + wm * 32 * kAttnPitch;
  const float* Kw = Ks + wn * 32 * kAttnPitch;
  {   // two accumulators (even / odd k-blocks) and the next pair's fragments in flight: a single dependent MFMA chain of 64
      // would wait out every LDS read
    dib_f32x16 s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = 0.f;
    float4 qa = dib_attn_kc(Qw, 0, l31, h), ka = dib_attn_kc(Kw, 0, l31, h);
    float4 qb = dib_attn_kc(Qw, 1, l31, h), kb = dib_attn_kc(Kw, 1, l31, h);
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      const int qn_ = q < 14 ? q + 2 : 14;
      const float4 qan = dib_attn_kc(Qw, qn_, l31, h), kan = dib_attn_kc(Kw, qn_, l31, h);
      const float4 qbn = dib_attn_kc(Qw, qn_ + 1, l31, h), kbn = dib_attn_kc(Kw, qn_ + 1, l31, h);
      __builtin_amdgcn_sched_barrier(0);
      s = DIB_MFMA(qa.x, ka.x, s);
      s1 = DIB_MFMA(qb.x, kb.x, s1);
      s = DIB_MFMA(qa.y, ka.y, s);
      s1 = DIB_MFMA(qb.y, kb.y, s1);
      s = DIB_MFMA(qa.z, ka.z, s);
      s1 = DIB_MFMA(qb.z, kb.z, s1);
      s = DIB_MFMA(qa.w, ka.w, s);
      s1 = DIB_MFMA(qb.w, kb.w, s1);
      __builtin_amdgcn_sched_barrier(0);
      qa = qan; ka = kan; qb = qbn; kb = kbn;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += s1[r];
  }
  __syncthreads();                               // every wave is done reading Q and K
  float* St = Ks;
#pragma unroll
  for (int r = 0; r < 16; ++r) St[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kAttnSP + wn * 32 + l31] = s[r];
  if constexpr (!PROJ) {
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(Qs + ((tid >> 5) + 8 * p) * kAttnPitch + (tid & 31) * 4) = vr[p];
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Qs[row * kAttnPitch + wn * 64 + 32 * j + l31] = row < P ? vacc[j][r] : 0.f;
      }
  }
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
  {
    float4 pp = *reinterpret_cast<const float4*>(St + (wm * 32 + l31) * kAttnSP + h * 4);
    float4 v0 = dib_attn_mc(Qs, 0, wn * 64 + l31, h), v1 = dib_attn_mc(Qs, 0, wn * 64 + 32 + l31, h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int qn_ = q < 7 ? q + 1 : 7;
      const float4 ppn = *reinterpret_cast<const float4*>(St + (wm * 32 + l31) * kAttnSP + qn_ * 8 + h * 4);
      const float4 v0n = dib_attn_mc(Qs, qn_, wn * 64 + l31, h), v1n = dib_attn_mc(Qs, qn_, wn * 64 + 32 + l31, h);
      __builtin_amdgcn_sched_barrier(0);
      o[0] = DIB_MFMA(pp.x, v0.x, o[0]);
      o[1] = DIB_MFMA(pp.x, v1.x, o[1]);
      o[0] = DIB_MFMA(pp.y, v0.y, o[0]);
      o[1] = DIB_MFMA(pp.y, v1.y, o[1]);
      o[0] = DIB_MFMA(pp.z, v0.z, o[0]);
      o[1] = DIB_MFMA(pp.z, v1.z, o[1]);
      o[0] = DIB_MFMA(pp.w, v0.w, o[0]);
      o[1] = DIB_MFMA(pp.w, v1.w, o[1]);
      __builtin_amdgcn_sched_barrier(0);
      pp = ppn; v0 = v0n; v1 = v1n;
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
    // one wave per SIMD (137 KB of LDS: one workgroup per CU): nobody else hides an LDS latency, so the fragments of step
    // q + 1 are issued before the 8 MFMAs of step q (as in dib_attn_bwd_kernel; pinned by sched_barrier)
    float4 qq = dib_attn_kc(Qw, 0, l31, h), gg = dib_attn_kc(Gw, 0, l31, h), kk = dib_attn_kc(Kw, 0, l31, h);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int qn_ = q < 15 ? q + 1 : 15;
      const float4 qn = dib_attn_kc(Qw, qn_, l31, h), gn = dib_attn_kc(Gw, qn_, l31, h), kn = dib_attn_kc(Kw, qn_, l31, h);
      __builtin_amdgcn_sched_barrier(0);
      s = DIB_MFMA(qq.x, kk.x, s);
      dp = DIB_MFMA(gg.x, vf[q].x, dp);
      s = DIB_MFMA(qq.y, kk.y, s);
      dp = DIB_MFMA(gg.y, vf[q].y, dp);
      s = DIB_MFMA(qq.z, kk.z, s);
      dp = DIB_MFMA(gg.z, vf[q].z, dp);
      s = DIB_MFMA(qq.w, kk.w, s);
      dp = DIB_MFMA(gg.w, vf[q].w, dp);
      __builtin_amdgcn_sched_barrier(0);
      qq = qn; gg = gn; kk = kn;
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
  {
    // operands of query block q: the probability / dS columns of this lane's key (A, 4 scalars each) and the dO / Q rows (B, one
    // MC fragment per 32-wide d tile); block q + 1 is fetched before the 16 MFMAs of block q
    struct Ops { float4 p, d, g0, g1, q0, q1; };
    auto fetch = [&](int q) {
      Ops o;
      const float* pp = Pt + (q * 8 + h * 4) * kAttnSP + wm * 32 + l31;
      const float* dd = dSt + (q * 8 + h * 4) * kAttnSP + wm * 32 + l31;
      o.p = make_float4(pp[0], pp[kAttnSP], pp[2 * kAttnSP], pp[3 * kAttnSP]);
      o.d = make_float4(dd[0], dd[kAttnSP], dd[2 * kAttnSP], dd[3 * kAttnSP]);
      o.g0 = dib_attn_mc(Gs, q, wn * 64 + l31, h);
      o.g1 = dib_attn_mc(Gs, q, wn * 64 + 32 + l31, h);
      o.q0 = dib_attn_mc(Qs, q, wn * 64 + l31, h);
      o.q1 = dib_attn_mc(Qs, q, wn * 64 + 32 + l31, h);
      return o;
    };
    Ops cur = fetch(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const Ops nxt = fetch(q < 7 ? q + 1 : 7);
      __builtin_amdgcn_sched_barrier(0);
      dv[0] = DIB_MFMA(cur.p.x, cur.g0.x, dv[0]);
      dv[1] = DIB_MFMA(cur.p.x, cur.g1.x, dv[1]);
      dk[0] = DIB_MFMA(cur.d.x, cur.q0.x, dk[0]);
      dk[1] = DIB_MFMA(cur.d.x, cur.q1.x, dk[1]);
      dv[0] = DIB_MFMA(cur.p.y, cur.g0.y, dv[0]);
      dv[1] = DIB_MFMA(cur.p.y, cur.g1.y, dv[1]);
      dk[0] = DIB_MFMA(cur.d.y, cur.q0.y, dk[0]);
      dk[1] = DIB_MFMA(cur.d.y, cur.q1.y, dk[1]);
      dv[0] = DIB_MFMA(cur.p.z, cur.g0.z, dv[0]);
      dv[1] = DIB_MFMA(cur.p.z, cur.g1.z, dv[1]);
      dk[0] = DIB_MFMA(cur.d.z, cur.q0.z, dk[0]);
      dk[1] = DIB_MFMA(cur.d.z, cur.q1.z, dk[1]);
      dv[0] = DIB_MFMA(cur.p.w, cur.g0.w, dv[0]);
      dv[1] = DIB_MFMA(cur.p.w, cur.g1.w, dv[1]);
      dk[0] = DIB_MFMA(cur.d.w, cur.q0.w, dk[0]);
      dk[1] = DIB_MFMA(cur.d.w, cur.q1.w, dk[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  // ---- dQ[query wm*32.., d wn*64..] = scale dS K: contraction over the 64 keys ----
  {
    float4 ds = *reinterpret_cast<const float4*>(dSt + (wm * 32 + l31) * kAttnSP + h * 4);
    float4 k0 = dib_attn_mc(Ks, 0, wn * 64 + l31, h), k1 = dib_attn_mc(Ks, 0, wn * 64 + 32 + l31, h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int qn_ = q < 7 ? q + 1 : 7;
      const float4 dsn = *reinterpret_cast<const float4*>(dSt + (wm * 32 + l31) * kAttnSP + qn_ * 8 + h * 4);
      const float4 k0n = dib_attn_mc(Ks, qn_, wn * 64 + l31, h), k1n = dib_attn_mc(Ks, qn_, wn * 64 + 32 + l31, h);
      __builtin_amdgcn_sched_barrier(0);
      dq[0] = DIB_MFMA(ds.x, k0.x, dq[0]);
      dq[1] = DIB_MFMA(ds.x, k1.x, dq[1]);
      dq[0] = DIB_MFMA(ds.y, k0.y, dq[0]);
      dq[1] = DIB_MFMA(ds.y, k1.y, dq[1]);
      dq[0] = DIB_MFMA(ds.z, k0.z, dq[0]);
      dq[1] = DIB_MFMA(ds.z, k1.z, dq[1]);
      dq[0] = DIB_MFMA(ds.w, k0.w, dq[0]);
      dq[1] = DIB_MFMA(ds.w, k1.w, dq[1]);
      __builtin_amdgcn_sched_barrier(0);
      ds = dsn; k0 = k0n; k1 = k1n;
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

// =====================================================================================================================
// The backward with 8 waves (round 5).  With 137 KB of LDS the kernel above is one workgroup per CU = ONE wave per SIMD: nothing
// overlaps its LDS latencies, its exponentials or its global loads / stores, and it runs at 3 x its MFMA time (57.8 us per
// launch of 384 workgroups at the notebook's size; the same finding as the first one-launch InfoNCE kernel,
// profiles/HISTORY.md 13).  Same LDS map, 512 threads, the five products split between the wave groups lo = waves 0-3 and
// hi = waves 4-7 (tile position (wm, wn) = bits of wave & 3 as above):
//   lo: S            hi: dP = dO V^T -> handed over through the dS tile's space
//   lo: P, delta, dS (the 16 exponentials per lane)
//   lo: dV = P^T dO  hi: dK = dS^T (scale Q)        (the same code on different tiles: two 32 x 32 tiles per wave)
//   all: dQ = scale dS K, one 32 x 32 tile per wave (query block wave & 1, columns 32 (wave >> 1))
// 160 MFMAs per wave instead of 320; every sum in the order of the 4-wave kernel (bit-identical results).  Same-box A/B of the
// notebook-size step: 1.488 -> 1.446 ms (57.8 -> 49.5 us per launch; the launch moves 69 MB, ~ 17 us at HBM speed, in 1.5
// rounds of workgroups).  The same split of the FORWARD kernel measured 1.5 % SLOWER on the step (its exchange tile and the
// idle half of the softmax cost more than the overlap buys) and was not kept (profiles/r05t_attention_8_waves_ab.txt).
// =====================================================================================================================
__device__ __forceinline__ void dib_attn_small_load8(float* __restrict__ T, const float* __restrict__ base, long long ld, int P,
                                                     int tid, float mul) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = (tid >> 5) + 16 * p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < P) v = *reinterpret_cast<const float4*>(base + (long long)r * ld + (tid & 31) * 4);
    *reinterpret_cast<float4*>(T + r * kAttnPitch + (tid & 31) * 4) = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
  }
}

// grid (H, B), 512 threads, dynamic LDS DibAttnSmallBwdLds floats
// PROJ (round 6): the kernel ends with the head's share of the q / k / v projections' input gradient - dq_h, dk_h, dv_h go
// through the (now dead) Q / K / dO tiles and are contracted with the head's 128 columns of the three kernels; the 12 heads'
// shares land in 12 slabs that the consumer sums in order (the previous block's chain launch, dib_st_chain_bwd g_out_slabs):
// the split-K dgrad GEMM + its slab reduce per block (16 + 8 us at the notebook's size) disappear.
template <bool PROJ>
__global__ void __launch_bounds__(512)
dib_attn_small_bwd8_kernel(DibAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                               // scaled Q
  float* Ks = Qs + kAttnSmallP * kAttnPitch;
  float* Gs = Ks + kAttnSmallP * kAttnPitch;     // dO
  float* Pt = Gs + kAttnSmallP * kAttnPitch;     // [64 queries][68] probabilities
  float* dSt = Pt + kAttnSmallP * kAttnSP;       // [64 queries][68] dP, then dS
  float* Ls = dSt + kAttnSmallP * kAttnSP;       // lse of the 64 queries (+inf beyond P)
  float* Dp = Ls + kAttnSmallP;                  // [2][64] partial delta (keys 0..31 | 32..63)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int head = blockIdx.x, b = blockIdx.y, P = a.P;
  const long long tok0 = (long long)b * P;
  const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
  const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
  const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
  const float* dOb = a.d_o + tok0 * a.ld + head * kAttnD;
  dib_attn_small_load8(Qs, Qb, a.ld, P, tid, a.scale);
  dib_attn_small_load8(Ks, Kb, a.ld, P, tid, 1.0f);
  dib_attn_small_load8(Gs, dOb, a.ld, P, tid, 1.0f);
  if (tid < kAttnSmallP) Ls[tid] = tid < P ? a.lse[((long long)b * a.H + head) * P + tid] : INFINITY;
  const int key = wn * 32 + l31;
  const float kmul = key < P ? 1.0f : 0.0f;
  // hi waves: the V row of this lane's key (B operand of dP = dO V^T), resident in registers
  float4 vf[16];
  if (grp == 1) dib_attn_rowfrag(vf, Vb, a.ld, min(key, P - 1), h, kmul);
  __syncthreads();
  // ---- lo: S, hi: dP   [query wm*32.., key wn*32..] ----
  dib_f32x16 sd;
#pragma unroll
  for (int r = 0; r < 16; ++r) sd[r] = 0.f;
  if (grp == 0) {
    const float* Qw = Qs + wm * 32 * kAttnPitch;
    const float* Kw = Ks + wn * 32 * kAttnPitch;
    float4 qq = dib_attn_kc(Qw, 0, l31, h), kk = dib_attn_kc(Kw, 0, l31, h);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int qn_ = q < 15 ? q + 1 : 15;
      const float4 qn = dib_attn_kc(Qw, qn_, l31, h), kn = dib_attn_kc(Kw, qn_, l31, h);
      sd = DIB_MFMA(qq.x, kk.x, sd);
      sd = DIB_MFMA(qq.y, kk.y, sd);
      sd = DIB_MFMA(qq.z, kk.z, sd);
      sd = DIB_MFMA(qq.w, kk.w, sd);
      qq = qn; kk = kn;
    }
  } else {
    const float* Gw = Gs + wm * 32 * kAttnPitch;
    float4 gg = dib_attn_kc(Gw, 0, l31, h);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 gn = dib_attn_kc(Gw, q < 15 ? q + 1 : 15, l31, h);
      sd = DIB_MFMA(gg.x, vf[q].x, sd);
      sd = DIB_MFMA(gg.y, vf[q].y, sd);
      sd = DIB_MFMA(gg.z, vf[q].z, sd);
      sd = DIB_MFMA(gg.w, vf[q].w, sd);
      gg = gn;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dSt[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kAttnSP + key] = sd[r];
  }
  __syncthreads();
  // ---- lo: P, delta, dS ----
  float pv[16], dpv[16];
  if (grp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dpv[r] = dSt[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kAttnSP + key];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      pv[r] = expf(sd[r] - Ls[qi]) * kmul;          // query beyond P: lse = +inf -> 0
      const float part = dib_half_sum(pv[r] * dpv[r]);
      if (l31 == 0) Dp[wn * kAttnSmallP + qi] = part;
    }
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float delta = Dp[qi] + Dp[kAttnSmallP + qi];
      Pt[qi * kAttnSP + key] = pv[r];
      dSt[qi * kAttnSP + key] = pv[r] * (dpv[r] - delta);
    }
  }
  __syncthreads();
  dib_f32x16 pj0, pj1;   // PROJ: the wave's two dV (lo) / dK (hi) tiles, kept for the epilogue
  // ---- lo: dV[key wm*32.., d wn*64..] = P^T dO;  hi: dK = dS^T (scale Q): contraction over the 64 queries ----
  {
    const float* At = grp == 0 ? Pt : dSt;
    const float* Bt = grp == 0 ? Gs : Qs;
    dib_f32x16 t0, t1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t0[r] = 0.f; t1[r] = 0.f; }
    struct Ops { float4 a, b0, b1; };
    auto fetch = [&](int q) {
      Ops o;
      const float* pp = At + (q * 8 + h * 4) * kAttnSP + wm * 32 + l31;
      o.a = make_float4(pp[0], pp[kAttnSP], pp[2 * kAttnSP], pp[3 * kAttnSP]);
      o.b0 = dib_attn_mc(Bt, q, wn * 64 + l31, h);
      o.b1 = dib_attn_mc(Bt, q, wn * 64 + 32 + l31, h);
      return o;
    };
    Ops cur = fetch(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const Ops nxt = fetch(q < 7 ? q + 1 : 7);
      t0 = DIB_MFMA(cur.a.x, cur.b0.x, t0);
      t1 = DIB_MFMA(cur.a.x, cur.b1.x, t1);
      t0 = DIB_MFMA(cur.a.y, cur.b0.y, t0);
      t1 = DIB_MFMA(cur.a.y, cur.b1.y, t1);
      t0 = DIB_MFMA(cur.a.z, cur.b0.z, t0);
      t1 = DIB_MFMA(cur.a.z, cur.b1.z, t1);
      t0 = DIB_MFMA(cur.a.w, cur.b0.w, t0);
      t1 = DIB_MFMA(cur.a.w, cur.b1.w, t1);
      cur = nxt;
    }
    float* outb = (grp == 0 ? a.dv : a.dk) + tok0 * a.ld + head * kAttnD;
    dib_attn_small_store(outb, a.ld, wm * 32, wn * 64 + l31, P, h, t0, 1.0f);
    dib_attn_small_store(outb, a.ld, wm * 32, wn * 64 + 32 + l31, P, h, t1, 1.0f);
    if constexpr (PROJ) { pj0 = t0; pj1 = t1; }
  }
  // ---- dQ[query 32 (wave & 1).., d 32 (wave >> 1)..] = scale dS K: contraction over the 64 keys ----
  {
    const int qm = wave & 1, cb = wave >> 1;
    dib_f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    const float* dsrow = dSt + (qm * 32 + l31) * kAttnSP + h * 4;
    float4 ds = *reinterpret_cast<const float4*>(dsrow);
    float4 kc = dib_attn_mc(Ks, 0, cb * 32 + l31, h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int qn_ = q < 7 ? q + 1 : 7;
      const float4 dsn = *reinterpret_cast<const float4*>(dsrow + qn_ * 8);
      const float4 kn = dib_attn_mc(Ks, qn_, cb * 32 + l31, h);
      dq = DIB_MFMA(ds.x, kc.x, dq);
      dq = DIB_MFMA(ds.y, kc.y, dq);
      dq = DIB_MFMA(ds.z, kc.z, dq);
      dq = DIB_MFMA(ds.w, kc.w, dq);
      ds = dsn; kc = kn;
    }
    dib_attn_small_store(a.dq + tok0 * a.ld + head * kAttnD, a.ld, qm * 32, cb * 32 + l31, P, h, dq, a.scale);
    if constexpr (PROJ) {
      // ---- dx_h [64 tokens][32] = dq_h Wq_h^T + dk_h Wk_h^T + dv_h Wv_h^T ----
      __syncthreads();                                   // every wave is done reading Q, K, dO, P, dS
#pragma unroll
      for (int r = 0; r < 16; ++r) {                     // C fragments -> row-major tiles: dq -> Qs, dk -> Ks, dv -> Gs
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        Qs[(qm * 32 + rr) * kAttnPitch + cb * 32 + l31] = dq[r] * a.scale;
        float* T = grp == 0 ? Gs : Ks;
        T[(wm * 32 + rr) * kAttnPitch + wn * 64 + l31] = pj0[r];
        T[(wm * 32 + rr) * kAttnPitch + wn * 64 + 32 + l31] = pj1[r];
      }
      __syncthreads();
      // wave w: token block rt = w & 1, quarter kq = w >> 1 of each projection's 128-long contraction
      const int rt = wave & 1, kq = wave >> 1;
      dib_f32x16 dx;
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[r] = 0.f;
#pragma unroll
      for (int pi = 0; pi < 3; ++pi) {
        const float* T = (pi == 0 ? Qs : (pi == 1 ? Ks : Gs)) + rt * 32 * kAttnPitch;
        const float* Wr = a.pparams + a.pw[pi] + (long long)l31 * a.ld + head * kAttnD + 32 * kq + 4 * h;   // W_i[c = l31][head's k]
        float4 wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const float4*>(Wr + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 av = dib_attn_kc(T, 4 * kq + q, l31, h);
          dx = DIB_MFMA(av.x, wv[q].x, dx);
          dx = DIB_MFMA(av.y, wv[q].y, dx);
          dx = DIB_MFMA(av.z, wv[q].z, dx);
          dx = DIB_MFMA(av.w, wv[q].w, dx);
        }
      }
      float* xq = Pt;                                    // exchange: [3 quarters][2 token blocks][16][64] floats (24 KB of the P | dS space)
      if (kq >= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xq[(((kq - 1) * 2 + rt) * 16 + r) * 64 + lane] = dx[r];
      }
      __syncthreads();
      if (kq == 0) {
#pragma unroll
        for (int pq_ = 0; pq_ < 3; ++pq_)
#pragma unroll
          for (int r = 0; r < 16; ++r) dx[r] += xq[((pq_ * 2 + rt) * 16 + r) * 64 + lane];
        float* dst = a.pdx + (long long)(1 + head) * a.pdx_stride + tok0 * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < P) dst[(long long)row * 32 + l31] = dx[r];
        }
      }
    }
  }
}
