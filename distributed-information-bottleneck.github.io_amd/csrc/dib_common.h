// dib_common.h - shared host/device helpers for the gfx950 Distributed-IB kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DIB_WAVE 64

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator + Box-Muller.  The reference samples eps with the stateful
// tf.random.normal (reference models.py:108); here eps is a pure function of
// (seed, step, global row, feature, dim) so forward, backward (which regenerates it instead of
// stashing 8 KB/sample), the CPU oracle (oracle/dib_oracle.py:philox_normal) and every
// data-parallel sharding agree.
// ---------------------------------------------------------------------------------------------
struct dib_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline dib_u4 dib_philox4x32_10(dib_u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per lane pair (v_mad_u64_u32) instead of separate mul_hi / mul_lo
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    dib_u4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// 4 standard normals for dims e = 4*blk .. 4*blk+3 of (row, feature) at `step`.
__host__ __device__ inline void dib_eps4(uint64_t seed, uint32_t step, uint32_t row, uint32_t feature,
                                         uint32_t blk, float out[4]) {
  dib_u4 c = {row, feature, blk, step};
  dib_u4 r = dib_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float s = 1.0f / 16777216.0f;
  const float u0 = ((float)(r.x >> 8) + 0.5f) * s;
  const float u1 = ((float)(r.y >> 8) + 0.5f) * s;
  const float u2 = ((float)(r.z >> 8) + 0.5f) * s;
  const float u3 = ((float)(r.w >> 8) + 0.5f) * s;
#if defined(__HIP_DEVICE_COMPILE__)
  // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions, so the
  // 2*pi*u range reduction is exact).  Measured against the fp64 oracle on 1 M normals: mean |diff| 1.1e-7, the same
  // as the libm path (1.05e-7), at a third of the instructions (fused fwd+bwd: -0.2 ms per step).
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  const float s0 = __builtin_amdgcn_sinf(u1), c0 = __builtin_amdgcn_cosf(u1);
  const float s1 = __builtin_amdgcn_sinf(u3), c1 = __builtin_amdgcn_cosf(u3);
#else
  const float r0 = sqrtf(-2.0f * logf(u0));
  const float r1 = sqrtf(-2.0f * logf(u2));
  const float s0 = sinf(6.283185307179586f * u1), c0 = cosf(6.283185307179586f * u1);
  const float s1 = sinf(6.283185307179586f * u3), c1 = cosf(6.283185307179586f * u3);
#endif
  out[0] = r0 * c0;
  out[1] = r0 * s0;
  out[2] = r1 * c1;
  out[3] = r1 * s1;
}

// ---------------------------------------------------------------------------------------------
// Activations.  ids match include/dib_hip.h.  Derivatives are expressed through the activation
// OUTPUT so that only post-activation tensors are stashed.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dib_act(int act, float z) {
  switch (act) {
    case 1: return fmaxf(z, 0.0f);
    case 2: return z > 0.0f ? z : 0.2f * z;
    case 3: return tanhf(z);
    case 4: return 1.0f / (1.0f + expf(-z));
    case 5: return z > 0.0f ? z : expm1f(z);
    case 6: return fmaxf(z, 0.0f) + log1pf(expf(-fabsf(z)));
    case 7: return z > 0.0f ? z : 0.1f * z;   // tf.keras.layers.LeakyReLU(0.1) of the set-transformer notebook
    default: return z;
  }
}

__device__ __forceinline__ float dib_act_grad(int act, float y) {
  switch (act) {
    case 1: return y > 0.0f ? 1.0f : 0.0f;
    case 2: return y > 0.0f ? 1.0f : 0.2f;
    case 3: return 1.0f - y * y;
    case 4: return y * (1.0f - y);
    case 5: return y > 0.0f ? 1.0f : y + 1.0f;
    case 6: return 1.0f - expf(-y);
    case 7: return y > 0.0f ? 1.0f : 0.1f;
    default: return 1.0f;
  }
}

// 64-lane wavefront sum (all lanes get the result)
__device__ __forceinline__ float dib_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0. `red` = 4 floats of LDS.
__device__ __forceinline__ float dib_block_sum_256(float v, float* red) {
  v = dib_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Phase marks of the small-batch kernels (diagnostic build -DDIB_SMALL_TIMING; tools/small_phase_timing.py): thread 0 of
// workgroup (0, 0) stores the 100 MHz wall clock at each phase boundary of the last launch.  Encoder forward marks at [0, 16),
// the integration kernel at [16, 40), the encoder backward at [40, 56), the one-launch InfoNCE kernel at [56, 64).
#ifdef DIB_SMALL_TIMING
__device__ long long dib_small_dbg[64];
#define DIB_ST(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) dib_small_dbg[i] = wall_clock64(); } while (0)
#else
#define DIB_ST(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Keras Adam on one element (reference train.py:128-129): m += (1-b1)(g-m); v += (1-b2)(g^2-v);
// theta -= lr_t m/(sqrt(v)+eps), lr_t = lr sqrt(1-b2^t)/(1-b1^t).  ONE definition with floating-point contraction OFF, used by
// dib_adam_kernel and by every segment of dib_step_tail_kernel: the same (theta, m, v, g) gives the same bits whichever launch
// applies the update (1-GPU fused tail, per-bucket data-parallel tails, the plain optimizer entry point).
// ---------------------------------------------------------------------------------------------
struct DibAdamCoef { float lr_t, c1, c2, eps, gscale; };   // c1 = 1 - beta_1, c2 = 1 - beta_2

__device__ __forceinline__ DibAdamCoef dib_adam_coef(float lr, long long t_applied, float b1, float b2, float eps, float gscale) {
#pragma clang fp contract(off)
  const float t = (float)(t_applied + 1);
  DibAdamCoef c;
  c.lr_t = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
  c.c1 = 1.f - b1; c.c2 = 1.f - b2; c.eps = eps; c.gscale = gscale;
  return c;
}

__device__ __forceinline__ void dib_adam_update(float& p, float& m, float& v, float g, const DibAdamCoef& c) {
#pragma clang fp contract(off)
  const float gg = g * c.gscale;
  const float d1 = gg - m, sq = gg * gg;
  const float m1 = c.c1 * d1;
  m = m + m1;
  const float d2 = sq - v;
  const float v1 = c.c2 * d2;
  v = v + v1;
  const float den = sqrtf(v) + c.eps;
  const float num = c.lr_t * m;
  p = p - num / den;
}
