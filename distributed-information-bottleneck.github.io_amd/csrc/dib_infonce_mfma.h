// dib_infonce_mfma.h - the InfoNCE similarity matrix and its embedding gradients on the gfx950 matrix cores.
//
// reference: train.py:203-215 (eval_batch_infonce: S = similarity(emb_x, emb_y) / T, loss = mean_i CE(i, S[i,:]) +
// mean_j CE(j, S[:,j])), utils.py:131-175 (get_scaled_similarity), utils.py:75-90 (pairwise squared distances as
// |a|^2 + |b|^2 - 2 a.b^T, clamped at 0).  Default similarity of the reference: 'l2' (train.py:57-58).
//
// For the three similarities built on the dot product - l2sq, l2, cosine - everything that is O(B^2 D) is a matrix product:
//   S-tile     ab = X Y^T                                [B, B]   K = D          dib_infonce_sim_mfma_kernel
//   g_x        C Y      with C_ij   = coefficient(S_ij)  [B, D]   K = B          dib_infonce_grad_mfma_kernel (side 0)
//   g_y        C^T X                                     [B, D]   K = B          dib_infonce_grad_mfma_kernel (side 1)
// (round 3 evaluated all three on the VALU: 0.38 ms of the 0.73 ms step at B = 2048, D = 64.)  l1 / linf are not bilinear
// and stay on the VALU kernels of dib_elementwise.h.
//
// With w_ij = dL/d(unscaled similarity)_ij = (softmax_row_i(S)_ij + softmax_col_j(S)_ij - 2 delta_ij) / (B T) the gradient wrt
// the row's own embedding a (partner b) is
//   l2sq   c = -2 w            (0 where the clamp max(d2, 0) is active)       g_a = sum_b c (a - b)  =  a R - C b,   R = sum_b c
//   l2     c = -w / r,  r = sqrt(d2 + 1e-9) = -S T  (0 where d2 = 0)          same form
//   cosine c = w / (|a||b|),  c2 = w sim / |a|^2                               g_a = C b - a R,               R = sum_b c2
// so each side is one product C . Other (MFMA) plus one row sum R (VALU, alongside the coefficient evaluation).
//
// Launches (all deterministic: fixed-order partials, the one atomic is an arrival counter):
//   sim    64 x 64 tile of S per workgroup (4 waves x one 32 x 32 MFMA tile, K chunks of 64 through LDS).  The row norms come
//          out of the staged tiles (sum of squares of what each thread stages, 16-lane DPP reduction); epilogue = norm /
//          -2ab / sqrt / 1/T, S written once; per (row, 32-column block) and per (column, 32-row block) partial
//          (max, sum exp) from the accumulator registers: along a column in-lane + one cross-half swizzle, along a row a
//          16-lane DPP butterfly + one swizzle (the first version's 162 ds_bpermute per tile were most of the kernel)
//   lse    combine the partials: lse_r[i], lse_c[j]; the LAST workgroup to finish (arrival counter) adds up the loss
//   grad   workgroup = 64 "self" rows x a slice of the partner tiles x side; per partner tile: read the S tile (side 1:
//          transposed through LDS), evaluate the coefficients, G += C . Other on the MFMAs, R += row sums.  With one slice
//          (B <= 256) the epilogue writes g = alpha self R + beta G itself; else partial G, R per slice and
//   final  g = alpha self R + beta G, slices summed in a fixed order
// = 3 launches at the reference's default batch of 128 (6 in the first version), 4 from B = 320.
#pragma once
#include "dib_common.h"
#include "dib_gemm.h"   // dib_f32x16, DIB_MFMA
#include "dib_fused.h"  // dib_f32x4, DIB_MFMA16

#define DIB_INCE_TS 64          // tile edge
#define DIB_INCE_KP 68          // LDS pitch of a k-contiguous operand tile (b128 fragment reads: conflict-free at BK + 4)
#define DIB_INCE_CP 65          // LDS pitch of the coefficient tile (odd: conflict-free for row-wise and transposed stores)

// ---- cross-lane helpers: DPP within a row of 16 lanes, ds_swizzle across the two rows of a 32-lane half ----
template <int CTRL>
__device__ __forceinline__ float dib_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float dib_swz_xor16(float v) {   // lane ^ 16 (bit-mask mode: and 0x1f, or 0, xor 0x10)
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
// reductions over the 32 lanes of a half-wave (all 32 get the result): quad_perm xor 1, xor 2, row_half_mirror, row_mirror
// (symmetric reductions: a mirror pairs what an xor would), then the other row of 16
__device__ __forceinline__ float dib_half_max(float v) {
  v = fmaxf(v, dib_dpp<0xB1>(v));
  v = fmaxf(v, dib_dpp<0x4E>(v));
  v = fmaxf(v, dib_dpp<0x141>(v));
  v = fmaxf(v, dib_dpp<0x140>(v));
  return fmaxf(v, dib_swz_xor16(v));
}
__device__ __forceinline__ float dib_half_sum(float v) {
  v += dib_dpp<0xB1>(v);
  v += dib_dpp<0x4E>(v);
  v += dib_dpp<0x141>(v);
  v += dib_dpp<0x140>(v);
  return v + dib_swz_xor16(v);
}
__device__ __forceinline__ float dib_row16_sum(float v) {   // over the 16 lanes of a DPP row
  v += dib_dpp<0xB1>(v);
  v += dib_dpp<0x4E>(v);
  v += dib_dpp<0x141>(v);
  return v + dib_dpp<0x140>(v);
}

// S_ij from the dot product.  KIND: 0 l2sq, 1 l2, 4 cosine.
template <int KIND>
__device__ __forceinline__ float dib_ince_similarity(float ab, float na, float nb, float inv_t) {
  float s;
  if (KIND == 4) {
    s = ab / (sqrtf(na) * sqrtf(nb));
  } else {  // utils.py:85-90: max(|a|^2 + |b|^2 - 2 a.b, 0)
    const float d2 = fmaxf(na + nb - 2.0f * ab, 0.f);
    s = (KIND == 0) ? -d2 : -sqrtf(d2 + 1e-9f);
  }
  return s * inv_t;
}

// grid (ceil(B/64) column tiles, ceil(B/64) row tiles), 256 threads.
// prow: [2][nb32][B] (plane 0 max, plane 1 sum of exp(s - max)) over the 32-column block cb of row i;  pcol: same for columns.
// norms [2B] (|x_i|^2, then |y_j|^2) is an OUTPUT: written by the first tile column / row, read by the gradient kernel.
// arrive: the log-sum-exp kernel's arrival counter, reset here (this launch precedes it on the stream).
template <int KIND>
__global__ void __launch_bounds__(256)
dib_infonce_sim_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Y, int B, int D, float inv_t,
                            float* __restrict__ norms, float* __restrict__ S, float* __restrict__ prow,
                            float* __restrict__ pcol, int nb32, unsigned* __restrict__ arrive) {
  __shared__ __attribute__((aligned(16))) float Xs[DIB_INCE_TS * DIB_INCE_KP];
  __shared__ __attribute__((aligned(16))) float Ys[DIB_INCE_TS * DIB_INCE_KP];
  __shared__ float Nx[DIB_INCE_TS], Ny[DIB_INCE_TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * DIB_INCE_TS, j0 = blockIdx.x * DIB_INCE_TS;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) arrive[0] = 0u;
  const bool vec = (D & 3) == 0 && ((((uintptr_t)X) | ((uintptr_t)Y)) & 15) == 0;
  dib_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lr = tid >> 4, lc = (tid & 15) * 4;     // this thread stages rows lr + 16 p, k columns lc .. lc + 3
  float nx[4] = {0.f, 0.f, 0.f, 0.f}, ny[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < D; k0 += 64) {
    if (k0) __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 16 * p, k = k0 + lc;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (i0 + r < B) {
        const float* src = X + (long long)(i0 + r) * D + k;
        if (vec && k + 3 < D) a = *reinterpret_cast<const float4*>(src);
        else { if (k < D) a.x = src[0]; if (k + 1 < D) a.y = src[1]; if (k + 2 < D) a.z = src[2]; if (k + 3 < D) a.w = src[3]; }
      }
      if (j0 + r < B) {
        const float* src = Y + (long long)(j0 + r) * D + k;
        if (vec && k + 3 < D) b = *reinterpret_cast<const float4*>(src);
        else { if (k < D) b.x = src[0]; if (k + 1 < D) b.y = src[1]; if (k + 2 < D) b.z = src[2]; if (k + 3 < D) b.w = src[3]; }
      }
      nx[p] += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      ny[p] += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
      *reinterpret_cast<float4*>(Xs + r * DIB_INCE_KP + lc) = a;
      *reinterpret_cast<float4*>(Ys + r * DIB_INCE_KP + lc) = b;
    }
    __syncthreads();
    const int kq = min(8, (D - k0 + 7) >> 3);          // 8-deep k blocks that hold data (the rest of the chunk is zero)
    for (int q = 0; q < kq; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(Xs + (wm * 32 + l31) * DIB_INCE_KP + q * 8 + h * 4);
      const float4 b = *reinterpret_cast<const float4*>(Ys + (wn * 32 + l31) * DIB_INCE_KP + q * 8 + h * 4);
      acc = DIB_MFMA(a.x, b.x, acc);
      acc = DIB_MFMA(a.y, b.y, acc);
      acc = DIB_MFMA(a.z, b.z, acc);
      acc = DIB_MFMA(a.w, b.w, acc);
    }
  }
  // row norms of both tiles: the 16 threads that staged a row are the 16 lanes of one DPP row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float sx = dib_row16_sum(nx[p]), sy = dib_row16_sum(ny[p]);
    if ((tid & 15) == 0) { Nx[lr + 16 * p] = sx; Ny[lr + 16 * p] = sy; }
  }
  __syncthreads();
  if (tid < DIB_INCE_TS) {
    if (blockIdx.x == 0 && i0 + tid < B) norms[i0 + tid] = Nx[tid];
    if (blockIdx.y == 0 && j0 + tid < B) norms[B + j0 + tid] = Ny[tid];
  }
  // ---- epilogue: C/D map of the 32 x 32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int j = j0 + wn * 32 + l31;
  const bool jok = j < B;
  const float nb = Ny[wn * 32 + l31];
  float s[16];
  float cmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int il = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, i = i0 + il;
    const bool ok = jok && i < B;
    const float v = dib_ince_similarity<KIND>(acc[r], Nx[il], nb, inv_t);
    if (ok) S[(long long)i * B + j] = v;
    s[r] = ok ? v : -INFINITY;
    cmax = fmaxf(cmax, s[r]);
  }
  // column partials: 16 rows in this lane + the other half-wave's 16
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
  float csum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) csum += (s[r] == -INFINITY) ? 0.f : __expf(s[r] - cmax);
  csum += __shfl_xor(csum, 32, 64);
  const int rb = (i0 + wm * 32) >> 5, cb = (j0 + wn * 32) >> 5;
  if (h == 0 && jok && rb < nb32) {
    pcol[(long long)rb * B + j] = cmax;
    pcol[(long long)(nb32 + rb) * B + j] = csum;
  }
  // row partials: the 32 columns of a row live in the 32 lanes of one half-wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m = dib_half_max(s[r]);
    const float e = dib_half_sum((s[r] == -INFINITY) ? 0.f : __expf(s[r] - m));
    const int i = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (l31 == 0 && i < B && cb < nb32) {
      prow[(long long)cb * B + i] = m;
      prow[(long long)(nb32 + cb) * B + i] = e;
    }
  }
}

// lse[0][i] = LSE_j S[i][j], lse[1][j] = LSE_i S[i][j] from the 32-wide block partials; the workgroup that arrives last (all
// lse values are then in memory) adds up loss = (1/B) sum_i (lse_r[i] + lse_c[i] - 2 S_ii) from one partial per workgroup
// (each workgroup sums lse - S_tt over its own 32 rows or columns).
// grid ceil(2B / 32): a workgroup owns 32 rows (or columns), thread (row = tid & 31, part = tid >> 5) merges every 8th block
// partial online - (m, s) <- (max(m, pm), s e^(m - m') + ps e^(pm - m')) - and the 8 parts of a row are merged in a fixed
// order through LDS.  (One thread per row walking all 2 x 64 partials of B = 2048 in two dependent passes took 47 us on 16
// workgroups - 40 % of the whole InfoNCE sequence, profiles/r04c_infonce_kernel_stats.csv.)
__global__ void __launch_bounds__(256)
dib_infonce_lse_loss_kernel(const float* __restrict__ prow, const float* __restrict__ pcol, const float* __restrict__ S, int B,
                            int nb32, float* __restrict__ lse, unsigned* __restrict__ arrive, float* __restrict__ lpart,
                            float* __restrict__ loss_out) {
  __shared__ float pm_s[8][32], ps_s[8][32];
  __shared__ float term[32];
  __shared__ float red[4];
  __shared__ bool last;
  const int r = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + r;                      // [0, B): rows, [B, 2B): columns (B is NOT assumed a multiple of 32:
  const bool ok = idx < 2 * B;                              //  a workgroup may straddle the two, idx decides per thread)
  const float* p = idx < B ? prow : pcol;
  const int t = idx < B ? idx : idx - B;
  float m = -INFINITY, sum = 0.f;
  if (ok) {
    for (int b0 = part; b0 < nb32; b0 += 32) {   // four partials per trip, their eight loads issued together (a load per
      float pm[4], ps[4];                        // dependent trip made this kernel 14-20 us at B = 2048)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 8 * u;
        pm[u] = b < nb32 ? p[(long long)b * B + t] : -INFINITY;
        ps[u] = b < nb32 ? p[(long long)(nb32 + b) * B + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (pm[u] == -INFINITY) continue;
        const float mn = fmaxf(m, pm[u]);
        sum = sum * expf(m - mn) + ps[u] * expf(pm[u] - mn);   // m = -inf on the first hit: sum = 0 * 0 + ...
        m = mn;
      }
    }
  }
  pm_s[part][r] = m;
  ps_s[part][r] = sum;
  __syncthreads();
  if (part == 0 && ok) {
    float mm = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) mm = fmaxf(mm, pm_s[q][r]);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += pm_s[q][r] == -INFINITY ? 0.f : ps_s[q][r] * expf(pm_s[q][r] - mm);
    const float l = mm + logf(tot);
    lse[idx] = l;
    term[r] = l - S[(long long)t * B + t];      // this row's (column's) share of the loss: lse - S_tt
  } else if (part == 0) {
    term[r] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // the workgroup's 32 terms in a fixed order, then the arrival
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += term[q];
    lpart[blockIdx.x] = sum;
    __threadfence();
    last = (atomicAdd(arrive, 1u) == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float sacc = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)   // the other workgroups' partials: bypass this CU's vector cache
    sacc += __builtin_nontemporal_load(lpart + i);
  const float tot = dib_block_sum_256(sacc, red);
  if (threadIdx.x == 0) loss_out[0] = tot / (float)B;
}

// grid (ceil(B/64) self blocks, nsplit partner slices, 2 sides), 256 threads, dynamic LDS (DIB_INCE_TS * (64 NACC + 4)) floats
// for the partner tile.  side 0: self = x rows, partners = y rows, coefficient tile read along rows of S; side 1: self = y
// rows, partners = x rows, the S tile is read along its rows (coalesced) and stored TRANSPOSED into the coefficient tile.
// nsplit == 1: GX / GY written directly; else Gp [2][nsplit][B][D], Rp [2][nsplit][B] for dib_infonce_grad_final_kernel.
template <int NACC, int KIND>
__global__ void __launch_bounds__(256)
dib_infonce_grad_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ S,
                             const float* __restrict__ lse, const float* __restrict__ norms, int B, int D, float inv_t,
                             float temperature, int nsplit, float* __restrict__ Gp, float* __restrict__ Rp,
                             float* __restrict__ GX, float* __restrict__ GY) {
  constexpr int DP = 64 * NACC, OP = DP + 4;
  extern __shared__ __attribute__((aligned(16))) float Os[];   // [64 partners][OP]
  __shared__ float Cs[DIB_INCE_TS * DIB_INCE_CP];              // [64 self][65]
  __shared__ float Rs[4][64];
  __shared__ float Rtot[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int side = blockIdx.z, split = blockIdx.y;
  const int s0 = blockIdx.x * DIB_INCE_TS;
  const float* Oth = side == 0 ? Y : X;
  const int ntiles = (B + DIB_INCE_TS - 1) / DIB_INCE_TS;
  const int tbeg = (int)(((long long)ntiles * split) / nsplit), tend = (int)(((long long)ntiles * (split + 1)) / nsplit);
  const float sc = inv_t / (float)B;
  const bool vecS = (B & 3) == 0 && (((uintptr_t)S) & 15) == 0;
  const bool vecO = (D & 3) == 0 && (((uintptr_t)Oth) & 15) == 0;
  const int lr = tid >> 4, lc = (tid & 15) * 4;   // S tile staging: tile rows lr + 16 p, tile columns lc .. lc + 3

  dib_f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};   // side 0: row sums of tile rows lr + 16 p;  side 1: of self columns lc + c

  for (int t = tbeg; t < tend; ++t) {
    const int o0 = t * DIB_INCE_TS;
    __syncthreads();   // the previous tile's MFMAs have read Cs / Os
    // ---- partner tile -> Os (zero beyond B and beyond D) ----
    for (int idx = tid; idx < DIB_INCE_TS * (DP / 4); idx += 256) {
      const int r = idx / (DP / 4), c = (idx - r * (DP / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o0 + r < B && c < D) {
        const float* src = Oth + (long long)(o0 + r) * D + c;
        if (vecO && c + 3 < D) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; if (c + 1 < D) v.y = src[1]; if (c + 2 < D) v.z = src[2]; if (c + 3 < D) v.w = src[3]; }
      }
      *reinterpret_cast<float4*>(Os + r * OP + c) = v;
    }
    // ---- S tile -> coefficients -> Cs[self][partner] ----
    const int rbase = side == 0 ? s0 : o0, cbase = side == 0 ? o0 : s0;   // S rows = x index, S columns = y index
    float lse_j[4], rn_j[4];                                              // column (y) log-sum-exp, 1 / |y_j|
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gj = min(cbase + lc + c, B - 1);
      lse_j[c] = lse[B + gj];
      rn_j[c] = KIND == 4 ? rsqrtf(norms[B + gj]) : 1.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int tr = lr + 16 * p, gi = rbase + tr;       // x index
      float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gi < B) {
        const float* src = S + (long long)gi * B + cbase + lc;
        if (vecS && cbase + lc + 3 < B) sv = *reinterpret_cast<const float4*>(src);
        else {
          if (cbase + lc < B) sv.x = src[0];
          if (cbase + lc + 1 < B) sv.y = src[1];
          if (cbase + lc + 2 < B) sv.z = src[2];
          if (cbase + lc + 3 < B) sv.w = src[3];
        }
      }
      const float svv[4] = {sv.x, sv.y, sv.z, sv.w};
      const float lse_i = lse[min(gi, B - 1)];                            // row (x) log-sum-exp
      const float rn_i = KIND == 4 ? rsqrtf(norms[min(gi, B - 1)]) : 1.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int tc = lc + c, gj = cbase + tc;      // y index
        float cf = 0.f, c2 = 0.f;
        if (gi < B && gj < B) {
          const float sij = svv[c];
          const float w = (__expf(sij - lse_i) + __expf(sij - lse_j[c]) - (gi == gj ? 2.0f : 0.f)) * sc;
          if (KIND == 0) cf = sij < 0.f ? -2.0f * w : 0.f;
          else if (KIND == 1) { const float rr = -sij * temperature; cf = (rr * rr > 1.0000005e-9f) ? -w * __builtin_amdgcn_rcpf(rr) : 0.f; }
          else {
            cf = w * rn_i * rn_j[c];
            c2 = w * (sij * temperature) * (side == 0 ? rn_i * rn_i : rn_j[c] * rn_j[c]);
          }
        }
        const float rterm = KIND == 4 ? c2 : cf;
        if (side == 0) { Cs[tr * DIB_INCE_CP + tc] = cf; rs[p] += rterm; }
        else           { Cs[tc * DIB_INCE_CP + tr] = cf; rs[c] += rterm; }
      }
    }
    __syncthreads();
    // ---- G[self 32 x (32 NACC)] += C[32 x 64] . Os[64 x (32 NACC)] ----
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float* ap = Cs + (wm * 32 + l31) * DIB_INCE_CP + q * 8 + h * 4;
      const float a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        const float* bp = Os + (q * 8 + h * 4) * OP + wn * 32 * NACC + n * 32 + l31;
        acc[n] = DIB_MFMA(a0, bp[0], acc[n]);
        acc[n] = DIB_MFMA(a1, bp[OP], acc[n]);
        acc[n] = DIB_MFMA(a2, bp[2 * OP], acc[n]);
        acc[n] = DIB_MFMA(a3, bp[3 * OP], acc[n]);
      }
    }
  }
  // ---- row sums of the coefficients: R[self] ----
  if (side == 0) {   // tile row lr + 16 p: the 16 threads of a row are the 16 lanes of one DPP row
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float v = dib_row16_sum(rs[p]);
      if ((tid & 15) == 0) Rtot[lr + 16 * p] = v;
    }
    __syncthreads();
  } else {           // self column lc + c: threads with equal tid & 15 - lanes 16 apart in a wave, then the 4 waves via LDS
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = rs[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) Rs[wave][lane * 4 + c] = v;
    }
    __syncthreads();
    if (tid < 64) Rtot[tid] = Rs[0][tid] + Rs[1][tid] + Rs[2][tid] + Rs[3][tid];
    __syncthreads();
  }
  // ---- outputs ----
  if (nsplit == 1) {   // g = alpha self R + beta G
    const float* Self = side == 0 ? X : Y;
    float* Gout = side == 0 ? GX : GY;
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      const int e = wn * 32 * NACC + n * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int il = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, i = s0 + il;
        if (i < B && e < D) {
          const float sr = Self[(long long)i * D + e] * Rtot[il];
          Gout[(long long)i * D + e] = KIND == 4 ? (acc[n][r] - sr) : (sr - acc[n][r]);
        }
      }
    }
    return;
  }
  float* G = Gp + ((long long)(side * nsplit + split) * B) * D;
#pragma unroll
  for (int n = 0; n < NACC; ++n) {
    const int e = wn * 32 * NACC + n * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = s0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (i < B && e < D) G[(long long)i * D + e] = acc[n][r];
    }
  }
  if (tid < 64 && s0 + tid < B) Rp[(long long)(side * nsplit + split) * B + s0 + tid] = Rtot[tid];
}

// g = alpha self R + beta G with the slices summed in a fixed order.  One thread per (side, row, coordinate).
__global__ void __launch_bounds__(256)
dib_infonce_grad_final_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ Gp,
                              const float* __restrict__ Rp, int B, int D, int kind, int nsplit, float* __restrict__ GX,
                              float* __restrict__ GY) {
  const long long per = (long long)B * D;
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= 2 * per) return;
  const int side = idx >= per ? 1 : 0;
  const long long o = idx - side * per;
  const int i = (int)(o / D);
  float g = 0.f, r = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    g += Gp[(long long)(side * nsplit + s) * per + o];
    r += Rp[(long long)(side * nsplit + s) * B + i];
  }
  const float self = (side == 0 ? X : Y)[o];
  (side == 0 ? GX : GY)[o] = kind == 4 ? (g - self * r) : (self * r - g);
}

// =====================================================================================================================
// One launch for the reference's DEFAULT batch (train.py:34: 128 rows, shared space 64, train.py:60): B <= 128, D <= 64.
// The three launches above spend 39 us on 2 MFLOP at that size (profiles/r05i_config2_loop_kernel_stats_b128.csv: similarity
// 12.3, log-sum-exp + loss 6.4, gradients 19.9 us - each a dependent launch of a handful of workgroups).  Here every
// workgroup keeps BOTH embedding sets and the whole [128][128] similarity matrix in its CU's LDS (136 KB of the 160) and
// recomputes S and the 2 B log-sum-exps itself - 8192 MFMA cycles per SIMD, cheaper than any exchange between workgroups -
// then evaluates the gradient of ITS 64 self rows of one side: grid (ceil(B/64) self blocks, 2 sides); without gradients (the
// validation pass) one workgroup.  512 threads = 2 waves per SIMD: the square roots / exponentials of one wave run under the
// MFMAs of the other (with 4 waves the phases ran back to back: S 11.6 us of which 3.7 are MFMAs, profiles/r05r_*).  The
// coefficient matrix overwrites S in place (side 1 reads it transposed: the LDS pitch is odd, so row-wise and column-wise
// accesses are both conflict-free).  Same expressions as the kernels above except v_sqrt_f32 / v_rsq_f32 (1 ulp) for the
// IEEE sqrtf / division sequences; every sum in a fixed order.  Workgroup (0, 0) writes the loss.
// =====================================================================================================================
#define DIB_INCE1_MAXB 128
#define DIB_INCE1_THREADS 512
#define DIB_INCE1_SP 129          // pitch of the similarity / coefficient matrix
#define DIB_INCE1_LDS_FLOATS (2 * DIB_INCE1_MAXB * DIB_INCE_KP + DIB_INCE1_MAXB * DIB_INCE1_SP + 4 * DIB_INCE1_MAXB + 4 * 256 + 8 * 64 + 64 + 8)

template <int KIND>
__device__ __forceinline__ float dib_ince1_similarity(float ab, float na, float nb, float inv_t) {
  if (KIND == 4) return ab * __builtin_amdgcn_rsqf(na) * __builtin_amdgcn_rsqf(nb) * inv_t;
  const float d2 = fmaxf(na + nb - 2.0f * ab, 0.f);   // utils.py:85-90
  return ((KIND == 0) ? -d2 : -__builtin_amdgcn_sqrtf(d2 + 1e-9f)) * inv_t;
}

template <int KIND>
__global__ void __launch_bounds__(DIB_INCE1_THREADS)
dib_infonce_small_kernel(const float* __restrict__ X, const float* __restrict__ Y, int B, int D, float inv_t, float temperature,
                         float* __restrict__ GX, float* __restrict__ GY, float* __restrict__ loss_out) {
  extern __shared__ __attribute__((aligned(16))) float lds1[];
  float* Xs = lds1;                                         // [128][68]  rows >= B and columns >= D are zero
  float* Ys = Xs + DIB_INCE1_MAXB * DIB_INCE_KP;
  float* Sm = Ys + DIB_INCE1_MAXB * DIB_INCE_KP;            // [128][129]
  float* Nx = Sm + DIB_INCE1_MAXB * DIB_INCE1_SP;           // |x_i|^2
  float* Ny = Nx + DIB_INCE1_MAXB;
  float* lse_r = Ny + DIB_INCE1_MAXB;
  float* lse_c = lse_r + DIB_INCE1_MAXB;
  float* mpart = lse_c + DIB_INCE1_MAXB;                    // [2 halves][256 rows | columns]
  float* spart = mpart + 2 * 256;
  float* Rpart = spart + 2 * 256;                           // [8][64]
  float* Rtot = Rpart + 8 * 64;                             // [64]
  float* red = Rtot + 64;                                   // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int nrow32 = (B + 31) >> 5;                         // 32-row blocks that hold data

  DIB_ST(56);
  // ---- phase 1: both embedding sets -> LDS (all loads of a thread in flight together), squared row norms ----
  {
    const bool vec = (D & 3) == 0 && ((((uintptr_t)X) | ((uintptr_t)Y)) & 15) == 0;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    float4 xa[4], ya[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 32 * p;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (r < B) {
        const float* sx = X + (long long)r * D + lc;
        const float* sy = Y + (long long)r * D + lc;
        if (vec && lc + 3 < D) { a = *reinterpret_cast<const float4*>(sx); b = *reinterpret_cast<const float4*>(sy); }
        else {
          if (lc < D) { a.x = sx[0]; b.x = sy[0]; }
          if (lc + 1 < D) { a.y = sx[1]; b.y = sy[1]; }
          if (lc + 2 < D) { a.z = sx[2]; b.z = sy[2]; }
          if (lc + 3 < D) { a.w = sx[3]; b.w = sy[3]; }
        }
      }
      xa[p] = a; ya[p] = b;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 32 * p;
      *reinterpret_cast<float4*>(Xs + r * DIB_INCE_KP + lc) = xa[p];
      *reinterpret_cast<float4*>(Ys + r * DIB_INCE_KP + lc) = ya[p];
      const float sx = dib_row16_sum(xa[p].x * xa[p].x + xa[p].y * xa[p].y + xa[p].z * xa[p].z + xa[p].w * xa[p].w);
      const float sy = dib_row16_sum(ya[p].x * ya[p].x + ya[p].y * ya[p].y + ya[p].z * ya[p].z + ya[p].w * ya[p].w);
      if ((tid & 15) == 0) { Nx[r] = sx; Ny[r] = sy; }
    }
  }
  __syncthreads();
  DIB_ST(57);

  // ---- phase 2: S = similarity(X Y^T): wave w owns rows [32 (w & 3), + 32) x column blocks 2 (w >> 2), 2 (w >> 2) + 1 ----
  {
    const int rb = wave & 3, kq = (D + 7) >> 3;
    const float* arow = Xs + (rb * 32 + l31) * DIB_INCE_KP + h * 4;
#pragma unroll 1
    for (int nn = 0; nn < 2; ++nn) {
      const int n = 2 * (wave >> 2) + nn;
      if (rb >= nrow32 || n >= nrow32) continue;   // wave-uniform
      dib_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* brow = Ys + (n * 32 + l31) * DIB_INCE_KP + h * 4;
#pragma unroll 2
      for (int q = 0; q < kq; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(arow + q * 8);
        const float4 b = *reinterpret_cast<const float4*>(brow + q * 8);
        acc = DIB_MFMA(a.x, b.x, acc);
        acc = DIB_MFMA(a.y, b.y, acc);
        acc = DIB_MFMA(a.z, b.z, acc);
        acc = DIB_MFMA(a.w, b.w, acc);
      }
      const int j = n * 32 + l31;
      const float nb = Ny[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Sm[i * DIB_INCE1_SP + j] = (i < B && j < B) ? dib_ince1_similarity<KIND>(acc[r], Nx[i], nb, inv_t) : -INFINITY;
      }
    }
  }
  __syncthreads();
  DIB_ST(58);

  // ---- phase 3: log-sum-exp of every row (t < 128) and column (t >= 128), t = tid & 255; the two halves of the workgroup
  // take entries [0, 64) and [64, 128) and are merged in that order ----
  {
    const int t = tid & 255, idx = t & 127, half = tid >> 8;
    const bool col = t >= 128;
    const int stride = col ? DIB_INCE1_SP : 1;
    const float* base = Sm + (col ? idx : idx * DIB_INCE1_SP);
    const int ubeg = 64 * half, uend = min(B, ubeg + 64);
    float m = -INFINITY, ssum = 0.f;
    if (idx < B && ubeg < uend) {
      // batches of 16 LDS reads in flight (a read per dependent trip would be one LDS latency per entry)
#pragma unroll 1
      for (int u0 = ubeg; u0 < uend; u0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = base[min(u0 + k, uend - 1) * stride];   // duplicates of the last entry: harmless for a max
#pragma unroll
        for (int k = 0; k < 16; ++k) m = fmaxf(m, v[k]);
      }
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four interleaved partial sums: a fixed order
#pragma unroll 1
      for (int u0 = ubeg; u0 < uend; u0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = base[min(u0 + k, uend - 1) * stride];
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          s0 += u0 + k < uend ? __expf(v[k] - m) : 0.f;
          s1 += u0 + k + 1 < uend ? __expf(v[k + 1] - m) : 0.f;
          s2 += u0 + k + 2 < uend ? __expf(v[k + 2] - m) : 0.f;
          s3 += u0 + k + 3 < uend ? __expf(v[k + 3] - m) : 0.f;
        }
      }
      ssum = (s0 + s1) + (s2 + s3);
    }
    mpart[half * 256 + t] = m;
    spart[half * 256 + t] = ssum;
  }
  __syncthreads();
  if (tid < 256) {
    const int idx = tid & 127;
    const float m0 = mpart[tid], m1 = mpart[256 + tid];
    const float mm = fmaxf(m0, m1);
    float l = 0.f;
    if (idx < B) {   // m0 is finite for idx < B (entry 0 exists); m1 = -inf when the second half is empty
      const float tot = spart[tid] * __expf(m0 - mm) + (m1 == -INFINITY ? 0.f : spart[256 + tid] * __expf(m1 - mm));
      l = mm + logf(tot);
    }
    (tid >= 128 ? lse_c : lse_r)[idx] = l;
  }
  __syncthreads();
  DIB_ST(59);
  if (blockIdx.x == 0 && blockIdx.y == 0) {   // loss = (1/B) sum_i (lse_r[i] + lse_c[i] - 2 S_ii)   (block-uniform branch)
    float term = 0.f;
    if (tid < B) term = (lse_r[tid] - Sm[tid * DIB_INCE1_SP + tid]) + (lse_c[tid] - Sm[tid * DIB_INCE1_SP + tid]);
    term = dib_wave_sum(term);
    if (lane == 0) red[wave] = term;
    __syncthreads();
    if (tid == 0) loss_out[0] = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) / (float)B;
  }
  DIB_ST(60);
  if (GX == nullptr || GY == nullptr) return;

  // ---- phase 4: coefficients of this workgroup's 64 self rows, in place; R[self] = their row sums ----
  const int side = blockIdx.y, s0 = blockIdx.x * 64;
  const float sc = inv_t / (float)B;
  {
    const int sl = tid & 63, part = tid >> 6, self = s0 + sl;   // part = wave: 16 partners each
    float rs = 0.f;
    const float lse_self = side == 0 ? lse_r[self] : lse_c[self];
    const float rn_self = KIND == 4 ? __builtin_amdgcn_rsqf(side == 0 ? Nx[self] : Ny[self]) : 1.f;
    const bool self_ok = self < B;
    const int estride = side == 0 ? 1 : DIB_INCE1_SP;                 // along the partners
    float* ebase = Sm + (side == 0 ? self * DIB_INCE1_SP : self);
    const float* lse_p = side == 0 ? lse_c : lse_r;
    const float* n_p = side == 0 ? Ny : Nx;
    if (part * 16 < 32 * nrow32) {                                    // partner blocks beyond the batch were never written
#pragma unroll 1
      for (int u0 = 0; u0 < 16; u0 += 8) {                            // 8 reads in flight, then 8 evaluations + stores
        float sv[8], lp[8], np_[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int p = part * 16 + u0 + k;
          sv[k] = ebase[p * estride];
          lp[k] = lse_p[p];
          np_[k] = KIND == 4 ? n_p[p] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int p = part * 16 + u0 + k;
          float cf = 0.f, c2 = 0.f;
          if (self_ok && p < B) {
            const float sij = sv[k];
            const float w = (__expf(sij - lse_self) + __expf(sij - lp[k]) - (p == self ? 2.0f : 0.f)) * sc;
            if (KIND == 0) cf = sij < 0.f ? -2.0f * w : 0.f;
            else if (KIND == 1) { const float rr = -sij * temperature; cf = (rr * rr > 1.0000005e-9f) ? -w * __builtin_amdgcn_rcpf(rr) : 0.f; }
            else {
              const float rn_p = __builtin_amdgcn_rsqf(np_[k]);
              cf = w * rn_self * rn_p;
              c2 = w * (sij * temperature) * (rn_self * rn_self);
            }
          }
          ebase[p * estride] = cf;
          rs += KIND == 4 ? c2 : cf;
        }
      }
    }
    Rpart[part * 64 + sl] = rs;
  }
  __syncthreads();
  if (tid < 64)
    Rtot[tid] = ((Rpart[tid] + Rpart[64 + tid]) + (Rpart[128 + tid] + Rpart[192 + tid])) +
                ((Rpart[256 + tid] + Rpart[320 + tid]) + (Rpart[384 + tid] + Rpart[448 + tid]));
  __syncthreads();
  DIB_ST(61);

  // ---- phase 5: G[64 self][64] = C . Other as 16 tiles of 16 x 16 (v_mfma_f32_16x16x4_f32), two per wave: self rows
  // [16 (w & 3), + 16) x columns [32 (w >> 2), + 32); g = alpha self R + beta G ----
  {
    const int ti = wave & 3, tj0 = 2 * (wave >> 2), j = lane & 15, q4 = lane >> 4;
    const float* Oth = side == 0 ? Ys : Xs;
    const float* Self = side == 0 ? Xs : Ys;
    float* Gout = side == 0 ? GX : GY;
    dib_f32x4 acc0 = dib_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int srow = s0 + ti * 16 + j;                         // A: lane (i = j, k = q4)
    const int astride = side == 0 ? 1 : DIB_INCE1_SP;
    const float* ap = Sm + (side == 0 ? srow * DIB_INCE1_SP : srow) + q4 * astride;
    const float* bp = Oth + q4 * DIB_INCE_KP + tj0 * 16 + j;   // B: lane (column j, k = q4)
    const int ksteps = 8 * nrow32;                             // partner index in steps of 4
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const float a = ap[4 * ks * astride];
      const float b0 = bp[4 * ks * DIB_INCE_KP], b1 = bp[4 * ks * DIB_INCE_KP + 16];
      acc0 = DIB_MFMA16(a, b0, acc0);
      acc1 = DIB_MFMA16(a, b1, acc1);
    }
    DIB_ST(62);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = ti * 16 + 4 * q4 + r, i = s0 + il;
      if (i < B) {
        const float rt = Rtot[il];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int e = (tj0 + c) * 16 + j;
          if (e < D) {
            const float sr = Self[i * DIB_INCE_KP + e] * rt;
            const float g = c == 0 ? acc0[r] : acc1[r];
            Gout[(long long)i * D + e] = KIND == 4 ? (g - sr) : (sr - g);
          }
        }
      }
    }
    DIB_ST(63);
  }
}
