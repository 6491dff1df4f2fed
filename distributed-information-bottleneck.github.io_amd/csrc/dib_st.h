// dib_st.h - row-wise kernels of the per-particle Distributed-IB set transformer (BASELINE config 5;
// reference complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb, code
// cell 8: "Create the particle encoder and the set transformer" ... `train_step`).
//
// The matrix products of that model (shared particle encoder, q/k/v/output projections, per-(neighbourhood, head)
// Q K^T and P V, feed-forward, head) run on the grouped fp32-MFMA GEMM of dib_gemm.h through dib_gemm_grouped; this file
// holds what sits between them: softmax over the key axis, Add + LayerNormalization (Keras epsilon 1e-3), the mean over the
// particle axis, and small utilities.  All HBM-bound, one wave (or half-wave) per row, fixed-order reductions
// (deterministic), exact fp32.
#pragma once
#include "dib_common.h"

__device__ __forceinline__ float dib_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a group of W consecutive lanes (W = 32 or 64); all lanes of the group get the result
template <int W>
__device__ __forceinline__ float dib_group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- softmax over the key axis (Keras MultiHeadAttention: softmax(q k^T / sqrt(key_dim))) -------------------------
// S: [rows][ld], the first P entries of a row are scores; in place: S <- softmax(scale * S).  One wave per row; rows of up
// to 64 * RPL entries are held in registers between the passes (one HBM read + one write per element instead of three
// reads and two writes - at 4096 particles the score tensor is 805 MB per block and neighbourhood, so this kernel IS its
// HBM traffic); longer rows fall back to re-reading.
template <int RPL>   // register entries per lane; RPL = 0: rows longer than 64 * 64, streamed
__global__ void __launch_bounds__(256)
dib_softmax_rows_fwd_kernel(float* __restrict__ S, long long rows, int P, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float* s = S + row * ld;
    if (RPL > 0) {
      float v[RPL > 0 ? RPL : 1];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < RPL; ++c) {
        const int j = lane + 64 * c;
        v[c] = j < P ? s[j] : -INFINITY;
        m = fmaxf(m, v[c]);
      }
      m = dib_wave_max(m);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < RPL; ++c) {
        v[c] = expf(scale * (v[c] - m));   // exp(-inf) = 0 for the padding
        sum += v[c];
      }
      const float inv = 1.0f / dib_wave_sum(sum);
#pragma unroll
      for (int c = 0; c < RPL; ++c) {
        const int j = lane + 64 * c;
        if (j < P) s[j] = v[c] * inv;
      }
    } else {
      float m = -INFINITY;
      for (int j = lane; j < P; j += 64) m = fmaxf(m, s[j]);
      m = dib_wave_max(m);
      float sum = 0.f;
      for (int j = lane; j < P; j += 64) {
        const float e = expf(scale * (s[j] - m));
        s[j] = e;
        sum += e;
      }
      const float inv = 1.0f / dib_wave_sum(sum);
      for (int j = lane; j < P; j += 64) s[j] *= inv;
    }
  }
}

// backward, in place on dP: dS = scale * P * (dP - sum_j dP_j P_j)   (gradient w.r.t. the UNSCALED scores q k^T)
template <int RPL>
__global__ void __launch_bounds__(256)
dib_softmax_rows_bwd_kernel(const float* __restrict__ Pm, float* __restrict__ dP, long long rows, int P, int ld,
                            float scale) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float* p = Pm + row * ld;
    float* d = dP + row * ld;
    if (RPL > 0) {
      float pv[RPL > 0 ? RPL : 1], dv[RPL > 0 ? RPL : 1];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < RPL; ++c) {
        const int j = lane + 64 * c;
        pv[c] = j < P ? p[j] : 0.f;
        dv[c] = j < P ? d[j] : 0.f;
        dot += dv[c] * pv[c];
      }
      dot = dib_wave_sum(dot);
#pragma unroll
      for (int c = 0; c < RPL; ++c) {
        const int j = lane + 64 * c;
        if (j < P) d[j] = scale * pv[c] * (dv[c] - dot);
      }
    } else {
      float dot = 0.f;
      for (int j = lane; j < P; j += 64) dot += d[j] * p[j];
      dot = dib_wave_sum(dot);
      for (int j = lane; j < P; j += 64) d[j] = scale * p[j] * (d[j] - dot);
    }
  }
}

// ---- Add + LayerNormalization (notebook: tf.keras.layers.Add()([x, y]) -> LayerNormalization(), epsilon 1e-3) -------
// y = (s - mean) / sqrt(var + eps) * gamma + beta, s = a + b, statistics over the last axis (D <= 256).
// W = 32 lanes per row for D <= 32 (two rows per wave), else 64.  xhat and rstd are stashed for the backward.
template <int W>
__global__ void __launch_bounds__(256)
dib_add_layernorm_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bv, int b_slabs, long long b_stride,
                             long long T, int D, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                             float* __restrict__ Y, float* __restrict__ xhat, float* __restrict__ rstd) {
  constexpr int RPW = 64 / W;  // rows per wave
  const int lane = threadIdx.x & 63, sub = lane / W, l = lane % W;
  const long long rows_per_grid = (long long)gridDim.x * 4 * RPW;
  for (long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub; row < (T + RPW - 1) / RPW * RPW;
       row += rows_per_grid) {
    const bool ok = row < T;
    float x[256 / W];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 256 / W; ++c) {
      const int j = l + c * W;
      float bv = 0.f;   // the second addend may arrive as b_slabs split-K partials b_stride apart: summed here in slab order
      if (ok && j < D)
        for (int sl = 0; sl < b_slabs; ++sl) bv += Bv[sl * b_stride + row * D + j];
      x[c] = (ok && j < D) ? A[row * D + j] + bv : 0.f;
      sum += x[c];
    }
    const float mean = dib_group_sum<W>(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 256 / W; ++c) {
      const int j = l + c * W;
      const float dlt = (j < D) ? x[c] - mean : 0.f;
      sq += dlt * dlt;
    }
    const float var = dib_group_sum<W>(sq) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
    if (ok) {
#pragma unroll
      for (int c = 0; c < 256 / W; ++c) {
        const int j = l + c * W;
        if (j < D) {
          const float xh = (x[c] - mean) * rs;
          xhat[row * D + j] = xh;
          Y[row * D + j] = xh * gamma[j] + beta[j];
        }
      }
      if (l == 0) rstd[row] = rs;
    }
  }
}

// backward: dxhat = dy * gamma; ds = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) (flows to BOTH addends);
// per-slot partial sums of dgamma = sum_rows dy * xhat and dbeta = sum_rows dy: partial[slot][2][D], slot = row group of
// the launch (fixed geometry => fixed order), reduced by dib_colsum_partials_kernel.
// Optional fusions (round 3: the set transformer at 1600 tokens is a chain of ~5 us launches):
//   dY2      second gradient addend: dY := dY + dY2 (the residual branch's gradient arriving at LN1 - was dib_add_inplace)
//   act_src  post-activation tensor of the branch that fed the Add: dZ = dS * act'(act_src) (the feed-forward block's last
//            relu in front of LN2 - was dib_act_grad_mul)
template <int W>
__global__ void __launch_bounds__(256)
dib_add_layernorm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ dY2, const float* __restrict__ xhat,
                             const float* __restrict__ rstd, const float* __restrict__ gamma, long long T, int D,
                             float* __restrict__ dS, const float* __restrict__ act_src, int act, float* __restrict__ dZ,
                             float* __restrict__ partial) {
  constexpr int RPW = 64 / W;
  const int lane = threadIdx.x & 63, sub = lane / W, l = lane % W;
  const long long rows_per_grid = (long long)gridDim.x * 4 * RPW;
  float pg[256 / W], pb[256 / W];
#pragma unroll
  for (int c = 0; c < 256 / W; ++c) pg[c] = pb[c] = 0.f;
  for (long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub; row < (T + RPW - 1) / RPW * RPW;
       row += rows_per_grid) {
    const bool ok = row < T;
    float dxh[256 / W], xh[256 / W];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 256 / W; ++c) {
      const int j = l + c * W;
      const bool in = ok && j < D;
      const float dy = in ? (dY2 ? dY[row * D + j] + dY2[row * D + j] : dY[row * D + j]) : 0.f;
      xh[c] = in ? xhat[row * D + j] : 0.f;
      dxh[c] = in ? dy * gamma[j] : 0.f;
      pg[c] += dy * xh[c];
      pb[c] += dy;
      s1 += dxh[c];
      s2 += dxh[c] * xh[c];
    }
    const float m1 = dib_group_sum<W>(s1) / (float)D, m2 = dib_group_sum<W>(s2) / (float)D;
    if (ok) {
      const float rs = rstd[row];
#pragma unroll
      for (int c = 0; c < 256 / W; ++c) {
        const int j = l + c * W;
        if (j < D) {
          const float v = rs * (dxh[c] - m1 - xh[c] * m2);
          dS[row * D + j] = v;
          if (dZ) dZ[row * D + j] = v * dib_act_grad(act, act_src[row * D + j]);
        }
      }
    }
  }
  const long long slot = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
#pragma unroll
  for (int c = 0; c < 256 / W; ++c) {
    const int j = l + c * W;
    if (j < D) {
      partial[slot * 2 * D + j] = pg[c];
      partial[slot * 2 * D + D + j] = pb[c];
    }
  }
}

// ---- mean over the particle axis (notebook: x = tf.reduce_mean(x, axis=-2)) ------------------------------------------
__global__ void __launch_bounds__(256)
dib_mean_pool_fwd_kernel(const float* __restrict__ X, int B, int P, int D, float* __restrict__ out) {
  // one workgroup per neighbourhood; thread = (particle lane, column): 256 / D particle lanes stride over the particles,
  // fixed-order LDS reduction over the lanes (the first version walked all P particles in one thread per column: 0.93 ms
  // at 4096 particles; the second had one dependent load in flight per thread: 120 us with 4 workgroups on the chip -
  // now four independent partial sums, combined in a fixed order)
  __shared__ float red[256];
  const int b = blockIdx.x;
  if (D <= 256 && 256 % D == 0) {
    const int lanes = 256 / D, pl = threadIdx.x / D, d = threadIdx.x % D;
    const float* src = X + (long long)b * P * D + d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = pl;
    for (; p + 3 * lanes < P; p += 4 * lanes) {
      s0 += src[(long long)p * D];
      s1 += src[(long long)(p + lanes) * D];
      s2 += src[(long long)(p + 2 * lanes) * D];
      s3 += src[(long long)(p + 3 * lanes) * D];
    }
    for (; p < P; p += lanes) s0 += src[(long long)p * D];
    const float s = (s0 + s1) + (s2 + s3);
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0) {
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += red[l * D + d];
      out[(long long)b * D + d] = t / (float)P;
    }
  } else {
    for (int d = threadIdx.x; d < D; d += 256) {
      float s = 0.f;
      for (int p = 0; p < P; ++p) s += X[((long long)b * P + p) * D + d];
      out[(long long)b * D + d] = s / (float)P;
    }
  }
}
__global__ void __launch_bounds__(256)
dib_mean_pool_bwd_kernel(const float* __restrict__ G, int B, int P, int D, float* __restrict__ dX) {
  const long long total = (long long)B * P * D;
  const float inv = 1.0f / (float)P;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / ((long long)P * D);
    const int d = (int)(i % D);
    dX[i] = G[b * D + d] * inv;
  }
}

// dst += src
__global__ void __launch_bounds__(256)
dib_add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] += src[i];
}

// out = g * act'(y)  (y = post-activation values): the activation mask of a layer whose output feeds a non-GEMM consumer
// (the feed-forward block's last relu before Add + LayerNormalization)
__global__ void __launch_bounds__(256)
dib_act_grad_mul_kernel(const float* __restrict__ g, const float* __restrict__ y, int act, long long n,
                        float* __restrict__ out) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = g[i] * dib_act_grad(act, y[i]);
}
