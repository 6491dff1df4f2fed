// dib_gemm.h - grouped fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This is the general (any shape, ragged groups) building block of the Distributed-IB path: the
// bank of per-feature encoder MLPs (reference models.py:73-78,106 - F independent Dense chains
// unrolled by a Python loop) runs as ONE grouped launch per layer (blockIdx.z = feature), and the
// integration network (models.py:81-84,122) uses the same kernel with one group.
//
//   MODE 0 (fwd)   C[M,N] = act( A[M,K] @ B[K,N] + bias[N] )            A row-major, B = Keras kernel [in,out]
//   MODE 1 (dgrad) C[M,N] = ( A[M,K] @ B[N,K]^T ) * act'(aux[M,N])      B = Keras kernel [in=N, out=K]
//   MODE 2 (wgrad) C[M,N] = A[Kb,M]^T @ B[Kb,N]  (+ column sums of B)   contraction over batch rows Kb,
//                  split over blockIdx.x into partial buffers (deterministic second-stage reduce)
//
// Arithmetic is exact fp32 (the reference is fp32 end to end; gfx950 has no TF32/xf32): the
// f32-input MFMA is bitwise an fmaf chain, peak 157.3 TFLOP/s = 64 FLOP/clk/SIMD.
//
// Tiling: 128x128x32 block tile, 256 threads = 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32
// (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS with the next tile's
// global loads in flight during the MFMAs of the current one.  Two LDS images:
//   KC ("k-contiguous", source rows run along k): T[128][32+4]; a lane fetches 4 consecutive k
//       with one ds_read_b128 and feeds 4 MFMAs (pitch 36 floats is conflict-free for b128).
//   MC ("mn-contiguous", source rows run along m/n): T[32][128+4]; one ds_read_b32 per MFMA,
//       32 consecutive floats per half-wave (conflict-free).
// Within an 8-deep k block q, MFMA step t contracts k = 8q + 4*(lane>>5) + t for BOTH operands
// (any consistent permutation of k is legal), which is what makes the b128 fetch possible.
#pragma once
#include "dib_common.h"

typedef float dib_f32x16 __attribute__((ext_vector_type(16)));

struct DibGemmGroup {
  long long a_off, b_off, c_off, bias_off, aux_off;  // element offsets into the base pointers
  int M, N, K;                                       // -1 => "batch" (runtime kernel argument)
  int lda, ldb, ldc, ldaux;
  int flags;                                         // bit0: A rows 16B-vectorisable, bit1: B rows
};

#define DIB_BM 128
#define DIB_BN 128
#define DIB_BK 32
#define DIB_KC_PITCH 36
#define DIB_MC_PITCH 132
#define DIB_TILE_FLOATS 4608  // max(128*36, 32*132)

// ---- global -> register staging ---------------------------------------------------------------
// KC: element (mn,k) = base[(mn0+mn)*ld + k0+k]   thread: mn = (tid>>3)+32p, k = 4*(tid&7)
// MC: element (k,mn) = base[(k0+k)*ld + mn0+mn]   thread: k = (tid>>5)+8p,  mn = 4*(tid&31)
template <bool KC>
__device__ __forceinline__ void dib_gload(float4 (&r)[4], const float* __restrict__ base, long long ld,
                                          int mn0, int mn_max, int k0, int k_max, bool vec, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int R = KC ? (mn0 + (tid >> 3) + 32 * p) : (k0 + (tid >> 5) + 8 * p);
    const int Cc = KC ? (k0 + (tid & 7) * 4) : (mn0 + (tid & 31) * 4);
    const int Rmax = KC ? mn_max : k_max;
    const int Cmax = KC ? k_max : mn_max;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (R < Rmax) {
      const float* src = base + (long long)R * ld + Cc;
      if (vec && Cc + 3 < Cmax) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        if (Cc + 0 < Cmax) v.x = src[0];
        if (Cc + 1 < Cmax) v.y = src[1];
        if (Cc + 2 < Cmax) v.z = src[2];
        if (Cc + 3 < Cmax) v.w = src[3];
      }
    }
    r[p] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void dib_lstore(float* __restrict__ T, const float4 (&r)[4], int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int off = KC ? (((tid >> 3) + 32 * p) * DIB_KC_PITCH + (tid & 7) * 4)
                       : (((tid >> 5) + 8 * p) * DIB_MC_PITCH + (tid & 31) * 4);
    *reinterpret_cast<float4*>(T + off) = r[p];
  }
}

// fetch the 4 operand values (MFMA steps t=0..3 of k-block q) for the 32-wide sub-tile at mn_base
template <bool KC>
__device__ __forceinline__ float4 dib_frag(const float* __restrict__ T, int mn_base, int q, int l31, int h) {
  if (KC) {
    return *reinterpret_cast<const float4*>(T + (mn_base + l31) * DIB_KC_PITCH + q * 8 + h * 4);
  } else {
    const float* p = T + (q * 8 + h * 4) * DIB_MC_PITCH + mn_base + l31;
    return make_float4(p[0], p[DIB_MC_PITCH], p[2 * DIB_MC_PITCH], p[3 * DIB_MC_PITCH]);
  }
}

#define DIB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(256)
dib_gemm_kernel(const DibGemmGroup* __restrict__ groups, const float* __restrict__ Abase,
                const float* __restrict__ Bbase, float* __restrict__ Cbase, const float* __restrict__ bias,
                const float* __restrict__ aux, float* __restrict__ bias_out, int batch, int act, int tiles_n,
                int rows_per_split, long long split_stride) {
  constexpr bool A_KC = (MODE != 2);
  constexpr bool B_KC = (MODE == 1);
  __shared__ __attribute__((aligned(16))) float smem[2 * DIB_TILE_FLOATS];
  float* As = smem;
  float* Bs = smem + DIB_TILE_FLOATS;

  const DibGemmGroup g = groups[blockIdx.z];
  const int M = g.M < 0 ? batch : g.M;
  const int N = g.N < 0 ? batch : g.N;
  const int K = g.K < 0 ? batch : g.K;
  int tm, tn, kbeg, kend;
  if (MODE == 2) {
    tn = blockIdx.y % tiles_n;
    tm = blockIdx.y / tiles_n;
    kbeg = blockIdx.x * rows_per_split;
    kend = min(K, kbeg + rows_per_split);
  } else {
    tm = blockIdx.x;
    tn = blockIdx.y;
    kbeg = 0;
    kend = K;
  }
  const int m0 = tm * DIB_BM, n0 = tn * DIB_BN;
  if (m0 >= M || n0 >= N) return;  // block-uniform

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const bool vecA = (g.flags & 1) != 0, vecB = (g.flags & 2) != 0;
  const float* Ag = Abase + g.a_off;
  const float* Bg = Bbase + g.b_off;

  // 32-wide sub-tiles of this wave that contain real output
  const int ni = min(2, max(0, (M - (m0 + wm * 64) + 31) >> 5));
  const int nj = min(2, max(0, (N - (n0 + wn * 64) + 31) >> 5));

  dib_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;  // MODE 2: column sums of B (bias gradient), only for tm == 0
  const bool do_bias = (MODE == 2) && (bias_out != nullptr) && (g.bias_off >= 0) && (tm == 0);

  float4 ra[4], rb[4];
  if (kbeg < kend) {
    dib_gload<A_KC>(ra, Ag, g.lda, m0, M, kbeg, kend, vecA, tid);
    dib_gload<B_KC>(rb, Bg, g.ldb, n0, N, kbeg, kend, vecB, tid);
  }
  for (int k0 = kbeg; k0 < kend; k0 += DIB_BK) {
    dib_lstore<A_KC>(As, ra, tid);
    dib_lstore<B_KC>(Bs, rb, tid);
    __syncthreads();
    if (k0 + DIB_BK < kend) {  // next tile's global loads fly during this tile's MFMAs
      dib_gload<A_KC>(ra, Ag, g.lda, m0, M, k0 + DIB_BK, kend, vecA, tid);
      dib_gload<B_KC>(rb, Bg, g.ldb, n0, N, k0 + DIB_BK, kend, vecB, tid);
    }
    if (do_bias) {
      const int col = tid & 127, half = tid >> 7;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += Bs[(half * 16 + r) * DIB_MC_PITCH + col];
    }
    const int nq = min(4, (kend - k0 + 7) >> 3);
    if (ni > 0 && nj > 0) {
      for (int q = 0; q < nq; ++q) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = dib_frag<A_KC>(As, wm * 64 + i * 32, q, l31, h);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = dib_frag<B_KC>(Bs, wn * 64 + j * 32, q, l31, h);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i < ni) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j < nj) {
                acc[i][j] = DIB_MFMA(a[i].x, b[j].x, acc[i][j]);
                acc[i][j] = DIB_MFMA(a[i].y, b[j].y, acc[i][j]);
                acc[i][j] = DIB_MFMA(a[i].z, b[j].z, acc[i][j]);
                acc[i][j] = DIB_MFMA(a[i].w, b[j].w, acc[i][j]);
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  float* Cg = Cbase + g.c_off + (MODE == 2 ? (long long)blockIdx.x * split_stride : 0ll);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (i < ni && j < nj) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col < N) {
          float bv = 0.f;
          if (MODE == 0 && bias != nullptr && g.bias_off >= 0) bv = bias[g.bias_off + col];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) {
              float v = acc[i][j][r];
              if (MODE == 0) {
                v = dib_act(act, v + bv);
              } else if (MODE == 1) {
                if (aux != nullptr && act != 0) v *= dib_act_grad(act, aux[g.aux_off + (long long)row * g.ldaux + col]);
              }
              Cg[(long long)row * g.ldc + col] = v;
            }
          }
        }
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    smem[tid] = bsum;
    __syncthreads();
    if (tid < 128 && n0 + tid < N)
      bias_out[g.bias_off + (long long)blockIdx.x * split_stride + n0 + tid] = smem[tid] + smem[tid + 128];
  }
}
