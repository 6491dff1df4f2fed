// dib_gemm_bf16x6.h - EXPERIMENTAL (not on the default path, not what bench.py measures): an fp32 GEMM whose products run
// on the bf16 matrix pipe.  Every fp32 operand x is split exactly into three bf16 pieces x = hi + mid + lo (8 + 8 + 8
// significand bits); a product a*b is the sum of the six piece products whose magnitude is >= 2^-16 |a b|
//   hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi          (the three dropped terms are below fp32 resolution)
// each of which is exact in fp32, accumulated in the fp32 MFMA accumulator (v_mfma_f32_32x32x16_bf16).  Measured
// accuracy = plain fp32 (tools/split_bf16_accuracy.py: 1.7e-7 vs 3.8e-7 of max|C|); measured pipe rate 284 TFLOP/s
// fp32-equivalent vs 154.5 for v_mfma_f32_32x32x2_f32 (tools/mfma_peak.hip).  This file is the first real kernel of that
// kind: C[M,N] = act(A[M,K] @ W[K,N] + bias) for the integration network's forward (reference models.py:81-84,122).
//
//   W is prepared once per optimizer step by dib_split_weights_kernel: three bf16 planes, TRANSPOSED to [N][Kp] so that both
//   operands are k-contiguous; A (activations) is split on the way into LDS.
//   Tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave 2 x 2 MFMA tiles; per 16-deep k step a wave issues
//   12 ds_read_b128 (3 pieces x 2 tiles x {A,B}) and 24 MFMAs.  LDS rows are 32 bf16 with an XOR chunk swizzle (see below).  The k assignment inside a 16-block is whatever the hardware uses for "8 elements per lane": A and B are
//   fetched identically, so the contraction is correct for any such assignment.
#pragma once
#include "dib_gemm.h"

typedef __bf16 dib_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dib_bf16x4 __attribute__((ext_vector_type(4)));

// fp32 [K][N] (Keras kernel) -> planes[3][N][Kp] bf16 (hi, mid, lo), Kp = K rounded up to 32, zero padded
__global__ void __launch_bounds__(256)
dib_split_weights_kernel(const float* __restrict__ W, int K, int N, int Kp, __bf16* __restrict__ planes) {
  const long long total = (long long)N * Kp;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / Kp), k = (int)(i - (long long)n * Kp);
    const float x = (k < K) ? W[(long long)k * N + n] : 0.f;
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    planes[i] = hi;
    planes[total + i] = mid;
    planes[2 * total + i] = lo;
  }
}

#define DIB_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__global__ void __launch_bounds__(256, 2)
dib_gemm_bf16x6_kernel(const float* __restrict__ A, int lda, const __bf16* __restrict__ Wp /*[3][N][Kp]*/, int Kp,
                       float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K, int act) {
  // LDS rows are 32 bf16 = four 16-byte chunks, unpadded; chunk c of row r lives at physical chunk c ^ ((r >> 2) & 3).
  // With this swizzle the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X_MICROARCH section LDS) hit 16
  // distinct 4-bank slots, and the row-major stores (two adjacent rows per 8- / 16-lane phase) fall on disjoint bank
  // halves.  (The first cut padded rows to 80 bytes: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 1.11.)
  constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK;
  auto swz = [](int row, int chunk) { return row * PITCH + ((chunk ^ ((row >> 2) & 3)) << 3); };
  __shared__ __attribute__((aligned(16))) __bf16 sA[3][BM * PITCH];
  __shared__ __attribute__((aligned(16))) __bf16 sB[3][BN * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (as dib_gemm_kernel): workgroup ids go round-robin over the 8 XCDs; all n-tiles of an m-tile run
  // back to back on one XCD so that the activation tile comes from that XCD's L2 after its first read
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;  // block-uniform
  const int m0 = tm * BM, n0 = tn * BN;
  const long long plane = (long long)N * Kp;

  dib_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging maps: A tile 128 x 32 fp32 = 1024 float4 (4 per thread: row = tid/8 + 32p, k = 4*(tid%8));
  //               B tile per plane 128 x 32 bf16 = 512 x 16 B (2 per thread: row = tid/4 + 64p, k = 8*(tid%4))
  const int ar = tid >> 3, ak = (tid & 7) * 4, br = tid >> 2, bk = (tid & 3) * 8;
  const bool avec = ((lda & 3) == 0) && ((reinterpret_cast<unsigned long long>(A) & 15) == 0);
  float4 ra0[4], ra1[4];   // activations are prefetched TWO K-tiles ahead (HBM latency ~2 us vs ~0.7 us of MFMAs per tile)
  uint4 rb[3][2];          // weight planes (L2-resident) one tile ahead
  const bool rows_in = m0 + BM <= M, cols_in = n0 + BN <= N;
  auto gload_a = [&](float4 (&ra)[4], int k0) {
    if (avec && rows_in && k0 + BK <= K) {  // interior tile: four unconditional 16-byte loads
#pragma unroll
      for (int p = 0; p < 4; ++p)
      {  // streaming operand: non-temporal, so that it does not evict the (re-used) weight planes from L2
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(A + (long long)(m0 + ar + 32 * p) * lda + k0 + ak));
        ra[p] = make_float4(v.x, v.y, v.z, v.w);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = m0 + ar + 32 * p, k = k0 + ak;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < M) {
        const float* src = A + (long long)row * lda + k;
        if (avec && k + 3 < K) v = *reinterpret_cast<const float4*>(src);
        else {
          if (k + 0 < K) v.x = src[0];
          if (k + 1 < K) v.y = src[1];
          if (k + 2 < K) v.z = src[2];
          if (k + 3 < K) v.w = src[3];
        }
      }
      ra[p] = v;
    }
  };
  auto gload_b = [&](int k0) {
    if (cols_in) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          rb[pl][p] = *reinterpret_cast<const uint4*>(Wp + pl * plane + (long long)(n0 + br + 64 * p) * Kp + k0 + bk);
      return;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = n0 + br + 64 * p;
        rb[pl][p] = (n < N) ? *reinterpret_cast<const uint4*>(Wp + pl * plane + (long long)n * Kp + k0 + bk) : make_uint4(0, 0, 0, 0);
      }
  };
  auto lstore = [&](const float4 (&ra)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float x[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w};
      dib_bf16x4 hi, mid, lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        hi[c] = (__bf16)x[c];
        const float r1 = x[c] - (float)hi[c];
        mid[c] = (__bf16)r1;
        lo[c] = (__bf16)(r1 - (float)mid[c]);
      }
      const int off = swz(ar + 32 * p, ak >> 3) + (ak & 4);
      *reinterpret_cast<dib_bf16x4*>(&sA[0][off]) = hi;
      *reinterpret_cast<dib_bf16x4*>(&sA[1][off]) = mid;
      *reinterpret_cast<dib_bf16x4*>(&sA[2][off]) = lo;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(&sB[pl][swz(br + 64 * p, bk >> 3)]) = rb[pl][p];
  };

  const int nkt = (K + BK - 1) / BK;
  auto mfma_tile = [&]() {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      dib_bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[pl][t] = *reinterpret_cast<const dib_bf16x8*>(&sA[pl][swz(wm * 64 + t * 32 + l31, ks * 2 + h)]);
          b[pl][t] = *reinterpret_cast<const dib_bf16x8*>(&sB[pl][swz(wn * 64 + t * 32 + l31, ks * 2 + h)]);
        }
      // six products per output tile, smallest first; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        const int pa = (term == 0) ? 2 : (term == 1) ? 0 : (term == 2) ? 1 : (term == 3) ? 1 : (term == 4) ? 0 : 0;
        const int pb = (term == 0) ? 0 : (term == 1) ? 2 : (term == 2) ? 1 : (term == 3) ? 0 : (term == 4) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = DIB_MFMA_BF16(a[pa][i], b[pb][j], acc[i][j]);
      }
    }
  };
  if (nkt > 0) { gload_a(ra0, 0); gload_b(0); }
  if (nkt > 1) gload_a(ra1, BK);
  for (int kt = 0; kt < nkt; kt += 2) {   // unrolled by two so that the prefetch ring is addressed statically
    lstore(ra0);
    __syncthreads();
    if (kt + 2 < nkt) gload_a(ra0, (kt + 2) * BK);
    if (kt + 1 < nkt) gload_b((kt + 1) * BK);
    mfma_tile();
    __syncthreads();
    if (kt + 1 < nkt) {
      lstore(ra1);
      __syncthreads();
      if (kt + 3 < nkt) gload_a(ra1, (kt + 3) * BK);
      if (kt + 2 < nkt) gload_b((kt + 2) * BK);
      mfma_tile();
      __syncthreads();
    }
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) C[(long long)row * ldc + col] = dib_act(act, acc[i][j][r] + bv);
      }
    }
}
