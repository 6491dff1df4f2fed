// dib_st_chain.h - the token-wise half of a set-transformer attention block as ONE launch per direction (round 5).
//
// A block of the reference's set transformer (...set_transformer.ipynb:332-389) is  MultiHeadAttention -> Add + LayerNorm ->
// feed-forward (Dense(relu))* -> Add + LayerNorm.  Everything after the attention itself acts on one token at a time:
//     mha = ctx @ W_o + b_o ;  h = LN1(x + mha) ;  f_0 = relu(h W_0 + b_0) ... f_last ;  x' = LN2(h + f_last)
// At the notebook's own size (32 neighbourhoods x 50 particles = 1600 tokens, 25 000 steps) these were 5 launches forward and
// 8-9 backward per block, each ~5-9 us of latency around microseconds of work: ~190 launches, 1.66 ms per step.  Here a
// workgroup owns 16 TOKENS (the row-tile machinery of dib_small.h: v_mfma_f32_16x16x4_f32, activations in LDS, weights streamed
// from L2 into the MFMA B operand) and runs the whole chain; the backward returns the operands of the block's weight
// gradients (which then run as ONE grouped launch together with the q/k/v projections') and reduces the LayerNorm parameter
// gradients itself (per-tile partials, summed in tile order by the last workgroup to arrive: deterministic).
// D (model width) % 32 == 0 and <= 256; feed-forward widths and heads x key_dim % 16 == 0; piecewise-linear activation.
#pragma once
#include "dib_small.h"
#include "dib_st.h"

#define DIB_ST_CHAIN_MAX_FF 3

struct DibStChainDesc {   // mirrors include/dib_st.h dib_st_block_desc
  long long o_w, o_b, ln1_g, ln1_b, ln2_g, ln2_b, ff_w[DIB_ST_CHAIN_MAX_FF], ff_b[DIB_ST_CHAIN_MAX_FF];
  int n_ff, ff_width[DIB_ST_CHAIN_MAX_FF];
  int D, HK;
  float eps;
  int act;
};

struct DibStChainFwdArgs {
  DibStChainDesc d; long long T; const float* params;
  const float* ctx; const float* x_in;
  float* h; float* xhat1; float* rstd1; float* ff[DIB_ST_CHAIN_MAX_FF]; float* x_out; float* xhat2; float* rstd2;
};

struct DibStChainBwdArgs {
  DibStChainDesc d; long long T; const float* params;
  const float* g_out; int g_slabs; long long g_stride;   // dL/dx' = sum of g_slabs buffers g_stride floats apart, in slab order
  const float* xhat2; const float* rstd2; const float* ff[DIB_ST_CHAIN_MAX_FF]; const float* xhat1;
  const float* rstd1;
  float* g_ff[DIB_ST_CHAIN_MAX_FF]; float* g_in; float* g_ctx;
  float* grads;          // LayerNorm gamma / beta gradients go to grads + ln{1,2}_{g,b}
  float* ln_partial;     // [tiles][4][D]
  unsigned* sync;        // one zero-initialised word (self-cleaning)
};

// y = LN(a + b) over the last axis for the tile's 16 rows: thread (row = tid >> 5, l = tid & 31) holds columns l + 32 c.
// a: global [T][D] rows r0.. (rows >= rows_valid read as 0); b: LDS tile; y -> LDS tile (+ global), xhat / rstd -> global.
// The expressions of dib_add_layernorm_fwd_kernel (two-pass variance, rstd = 1 / sqrt(var + eps)).
__device__ __forceinline__ void dib_st_ln_fwd_tile(const float* __restrict__ a, long long r0, int rows_valid, int D, const float* b,
                                                   int pb, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, float* y_lds, int py, float* __restrict__ y, float* __restrict__ xhat,
                                                   float* __restrict__ rstd) {
  const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
  const bool ok = row < rows_valid;
  float x[8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int j = l + 32 * c;
    x[c] = j < D ? (ok ? a[(r0 + row) * D + j] : 0.f) + b[row * pb + j] : 0.f;
    sum += x[c];
  }
  const float mean = dib_group_sum<32>(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float dlt = (l + 32 * c < D) ? x[c] - mean : 0.f;
    sq += dlt * dlt;
  }
  const float var = dib_group_sum<32>(sq) / (float)D;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int j = l + 32 * c;
    if (j < D) {
      const float xh = (x[c] - mean) * rs;
      const float yv = xh * gamma[j] + beta[j];
      if (y_lds != nullptr) y_lds[row * py + j] = yv;
      if (ok) {
        xhat[(r0 + row) * D + j] = xh;
        y[(r0 + row) * D + j] = yv;
      }
    }
  }
  if (ok && l == 0) rstd[r0 + row] = rs;
}

// ds = LN backward of dy (LDS tile, rows >= rows_valid zero): dxhat = dy gamma; ds = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
// (the expressions of dib_add_layernorm_bwd_kernel); ds -> LDS tile (+ global); this tile's sums over rows of dy xhat
// (dgamma) and dy (dbeta) -> partial[0 .. 2 D) in row order.  red: LDS [2][16][D].  Ends with a workgroup barrier.
__device__ __forceinline__ void dib_st_ln_bwd_tile(const float* dy, int pd, const float* __restrict__ xhat,
                                                   const float* __restrict__ rstd, long long r0, int rows_valid, int D,
                                                   const float* __restrict__ gamma, float* ds_lds, int ps, float* __restrict__ ds,
                                                   float* red, float* __restrict__ partial) {
  const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
  const bool ok = row < rows_valid;
  float dxh[8], xh[8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int j = l + 32 * c;
    const bool in = j < D;
    const float g = in ? dy[row * pd + j] : 0.f;
    xh[c] = (in && ok) ? xhat[(r0 + row) * D + j] : 0.f;
    dxh[c] = in ? g * gamma[j] : 0.f;
    if (in) {
      red[row * D + j] = g * xh[c];
      red[(DIB_SMALL_ROWS + row) * D + j] = g;
    }
    s1 += dxh[c];
    s2 += dxh[c] * xh[c];
  }
  const float m1 = dib_group_sum<32>(s1) / (float)D, m2 = dib_group_sum<32>(s2) / (float)D;
  const float rs = ok ? rstd[r0 + row] : 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int j = l + 32 * c;
    if (j < D) {
      const float v = rs * (dxh[c] - m1 - xh[c] * m2);
      ds_lds[row * ps + j] = v;
      if (ok && ds != nullptr) ds[(r0 + row) * D + j] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += DIB_SMALL_THREADS) {   // [dgamma (D) | dbeta (D)], rows summed in order
    const int which = i / D, j = i - which * D;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < DIB_SMALL_ROWS; ++r) t += red[(which * DIB_SMALL_ROWS + r) * D + j];
    partial[i] = t;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_st_chain_fwd_kernel(DibStChainFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DibStChainDesc& d = a.d;
  const int tile = blockIdx.x;
  const long long r0 = (long long)tile * DIB_SMALL_ROWS;
  const int rows_valid = (int)min((long long)DIB_SMALL_ROWS, a.T - r0);
  const int D = d.D, pD = dib_small_pitch(D), pC = dib_small_pitch(d.HK);
  // LDS: ctx tile | mha | h | ff_0 .. | exchange
  float* cs = lds;
  float* ms = cs + DIB_SMALL_ROWS * pC;
  float* hs = ms + DIB_SMALL_ROWS * pD;
  float* fs[DIB_ST_CHAIN_MAX_FF]; int pf[DIB_ST_CHAIN_MAX_FF];
  float* cur = hs + DIB_SMALL_ROWS * pD;
#pragma unroll
  for (int l = 0; l < DIB_ST_CHAIN_MAX_FF; ++l) {
    pf[l] = l < d.n_ff ? dib_small_pitch(d.ff_width[l]) : 0;
    fs[l] = cur; cur += DIB_SMALL_ROWS * pf[l];
  }
  float* xch = cur;
  const float slope = dib_neg_slope(d.act);
  dib_small_load_tile(a.ctx + r0 * d.HK, d.HK, d.HK, rows_valid, cs, pC);
  __syncthreads();
  // attention output projection (linear), then h = LN1(x + mha)
  dib_small_fwd(cs, pC, d.HK, d.HK, a.params + d.o_w, D, a.params + d.o_b, 1.f, ms, pD, nullptr, 0, rows_valid, xch);
  dib_st_ln_fwd_tile(a.x_in, r0, rows_valid, D, ms, pD, a.params + d.ln1_g, a.params + d.ln1_b, d.eps, hs, pD, a.h, a.xhat1, a.rstd1);
  __syncthreads();
  // feed-forward chain
#pragma unroll
  for (int l = 0; l < DIB_ST_CHAIN_MAX_FF; ++l) {
    if (l < d.n_ff) {
      const float* in = l == 0 ? hs : fs[l > 0 ? l - 1 : 0];
      const int K = l == 0 ? D : d.ff_width[l > 0 ? l - 1 : 0], pin = l == 0 ? pD : pf[l > 0 ? l - 1 : 0];
      dib_small_fwd(in, pin, K, K, a.params + d.ff_w[l], d.ff_width[l], a.params + d.ff_b[l], slope, fs[l], pf[l],
                    a.ff[l] + r0 * d.ff_width[l], d.ff_width[l], rows_valid, xch);
    }
  }
  // x' = LN2(h + f_last), both addends from their LDS tiles (the expressions of dib_st_ln_fwd_tile)
  const float* fl = d.n_ff == 1 ? fs[0] : (d.n_ff == 2 ? fs[1] : fs[2]);
  const int pl = d.n_ff == 1 ? pf[0] : (d.n_ff == 2 ? pf[1] : pf[2]);
  {
    const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
    const bool ok = row < rows_valid;
    float x[8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = l + 32 * c;
      x[c] = j < D ? hs[row * pD + j] + fl[row * pl + j] : 0.f;
      sum += x[c];
    }
    const float mean = dib_group_sum<32>(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float dlt = (l + 32 * c < D) ? x[c] - mean : 0.f;
      sq += dlt * dlt;
    }
    const float var = dib_group_sum<32>(sq) / (float)D;
    const float rs = 1.0f / sqrtf(var + d.eps);
    const float* gamma = a.params + d.ln2_g;
    const float* beta = a.params + d.ln2_b;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = l + 32 * c;
      if (j < D && ok) {
        const float xh = (x[c] - mean) * rs;
        a.xhat2[(r0 + row) * D + j] = xh;
        a.x_out[(r0 + row) * D + j] = xh * gamma[j] + beta[j];
      }
    }
    if (ok && l == 0) a.rstd2[r0 + row] = rs;
  }
}

__global__ void __launch_bounds__(DIB_SMALL_THREADS)
dib_st_chain_bwd_kernel(DibStChainBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ bool s_last;
  const DibStChainDesc& d = a.d;
  const int tile = blockIdx.x, tid = threadIdx.x;
  const long long r0 = (long long)tile * DIB_SMALL_ROWS;
  const int rows_valid = (int)min((long long)DIB_SMALL_ROWS, a.T - r0);
  const int D = d.D, pD = dib_small_pitch(D);
  const int nff = d.n_ff;
  // LDS: g (dL/dx') | g_a | g_h | v | LN reduction scratch [2][16][D] | per ff layer: g_z tile, f tile | exchange
  float* gs = lds;
  float* ga = gs + DIB_SMALL_ROWS * pD;
  float* gh = ga + DIB_SMALL_ROWS * pD;
  float* vs = gh + DIB_SMALL_ROWS * pD;
  float* red = vs + DIB_SMALL_ROWS * pD;
  float* cur = red + 2 * DIB_SMALL_ROWS * D;
  float* gz[DIB_ST_CHAIN_MAX_FF]; float* fs[DIB_ST_CHAIN_MAX_FF]; int pf[DIB_ST_CHAIN_MAX_FF];
#pragma unroll
  for (int l = 0; l < DIB_ST_CHAIN_MAX_FF; ++l) {
    pf[l] = l < nff ? dib_small_pitch(d.ff_width[l]) : 0;
    gz[l] = cur; cur += DIB_SMALL_ROWS * pf[l];
    fs[l] = cur; cur += DIB_SMALL_ROWS * pf[l];
  }
  float* xch = cur;
  const float slope = dib_neg_slope(d.act);
  float* part = a.ln_partial + (long long)tile * 4 * D;

  if (a.g_slabs <= 1) {
    dib_small_load_tile(a.g_out + r0 * D, D, D, rows_valid, gs, pD);
  } else {
    // the gradient arrives as partial slabs (round 6: [the next block's LN1-addend gradient | the split-K slabs of its
    // q / k / v input gradient]): summed here in slab order, 4 loads in flight per thread - instead of a reduce launch per block
    const int w4 = D >> 2, total = DIB_SMALL_ROWS * w4;
    for (int i = tid; i < total; i += DIB_SMALL_THREADS) {
      const int row = i / w4, c = (i - row * w4) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows_valid) {
        const float* src = a.g_out + (r0 + row) * D + c;
        int sl = 0;
        for (; sl + 4 <= a.g_slabs; sl += 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long long)(sl + u) * a.g_stride);
#pragma unroll
          for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; sl < a.g_slabs; ++sl) {
          const float4 v = *reinterpret_cast<const float4*>(src + (long long)sl * a.g_stride);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      *reinterpret_cast<float4*>(gs + row * pD + c) = acc;
    }
  }
#pragma unroll
  for (int l = 0; l < DIB_ST_CHAIN_MAX_FF; ++l)
    if (l < nff) dib_small_load_tile(a.ff[l] + r0 * d.ff_width[l], d.ff_width[l], d.ff_width[l], rows_valid, fs[l], pf[l]);
  __syncthreads();
  // x' = LN2(h + f_last): g_a = gradient of both addends; its parameter-gradient partials
  dib_st_ln_bwd_tile(gs, pD, a.xhat2, a.rstd2, r0, rows_valid, D, a.params + d.ln2_g, ga, pD, nullptr, red, part);
  // dL/d(pre-activation of the last feed-forward layer) = g_a (.) act'(f_last)
  {
    float* const gzl = nff == 1 ? gz[0] : (nff == 2 ? gz[1] : gz[2]);
    const float* const fl = nff == 1 ? fs[0] : (nff == 2 ? fs[1] : fs[2]);
    const int pl = nff == 1 ? pf[0] : (nff == 2 ? pf[1] : pf[2]);
    float* const gdst = nff == 1 ? a.g_ff[0] : (nff == 2 ? a.g_ff[1] : a.g_ff[2]);
    for (int i = tid; i < DIB_SMALL_ROWS * D; i += DIB_SMALL_THREADS) {
      const int row = i / D, j = i - row * D;
      const float v = ga[row * pD + j] * dib_small_act_grad(slope, fl[row * pl + j]);
      gzl[row * pl + j] = v;
      if (row < rows_valid) gdst[(r0 + row) * D + j] = v;
    }
    __syncthreads();
  }
  // dgrad chain through the feed-forward layers down to dL/dh
#pragma unroll
  for (int l = DIB_ST_CHAIN_MAX_FF - 1; l >= 1; --l) {
    if (l < nff)
      dib_small_bwd(gz[l], pf[l], d.ff_width[l], a.params + d.ff_w[l], d.ff_width[l - 1], fs[l - 1], pf[l - 1], slope, gz[l - 1],
                    pf[l - 1], a.g_ff[l - 1] + r0 * d.ff_width[l - 1], d.ff_width[l - 1], rows_valid, xch);
  }
  dib_small_bwd(gz[0], pf[0], d.ff_width[0], a.params + d.ff_w[0], D, nullptr, 0, 1.f, gh, pD, nullptr, 0, rows_valid, xch);
  // h = LN1(x + mha): dy = dL/dh from the feed-forward branch + the residual g_a
  for (int i = tid; i < DIB_SMALL_ROWS * D; i += DIB_SMALL_THREADS) {
    const int row = i / D, j = i - row * D;
    gh[row * pD + j] += ga[row * pD + j];
  }
  __syncthreads();
  dib_st_ln_bwd_tile(gh, pD, a.xhat1, a.rstd1, r0, rows_valid, D, a.params + d.ln1_g, vs, pD, a.g_in, red, part + 2 * D);
  // dL/d(attention context) = v @ W_o^T   (v = gradient of LN1's two addends: the block input's residual share and mha)
  dib_small_bwd(vs, pD, D, a.params + d.o_w, d.HK, nullptr, 0, 1.f, nullptr, 0, a.g_ctx + r0 * d.HK, d.HK, rows_valid, xch);
  // LayerNorm parameter gradients: the last workgroup to arrive sums the tiles' partials in tile order
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
  // 4 D columns x nch interleaved tile chunks (8 independent loads in flight per thread), chunks combined in order
  const int ncol = 4 * D, nch = max(1, DIB_SMALL_THREADS / ncol), ntl = (int)gridDim.x;
  float* comb = lds;   // [nch][4 D]: every tile of this workgroup is done with its LDS
  for (int i0 = 0; i0 < ncol; i0 += DIB_SMALL_THREADS) {
    const int i = i0 + (tid % min(ncol, DIB_SMALL_THREADS)), ch = tid / min(ncol, DIB_SMALL_THREADS);
    if (i < ncol && ch < nch) {
      const float* src = a.ln_partial + i;
      float t = 0.f;
      int tl = ch;
      for (; tl + 7 * nch < ntl; tl += 8 * nch) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long long)(tl + u * nch) * ncol];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
      for (; tl < ntl; tl += nch) t += src[(long long)tl * ncol];
      comb[ch * ncol + i] = t;
    }
    __syncthreads();
    if (i < ncol && ch == 0) {
      float t = 0.f;
      for (int c = 0; c < nch; ++c) t += comb[c * ncol + i];
      const int which = i / D, j = i - which * D;   // 0: dgamma2, 1: dbeta2, 2: dgamma1, 3: dbeta1
      const long long off = which == 0 ? d.ln2_g : (which == 1 ? d.ln2_b : (which == 2 ? d.ln1_g : d.ln1_b));
      a.grads[off + j] = t;
    }
    __syncthreads();
  }
}
