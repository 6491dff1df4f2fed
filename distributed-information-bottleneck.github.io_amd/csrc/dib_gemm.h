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
//                  split over blockIdx.x into partial slabs (deterministic second-stage reduce)
//
// Arithmetic is exact fp32 (the reference is fp32 end to end; gfx950 has no TF32/xf32): the
// f32-input MFMA is bitwise an fmaf chain, peak 157.3 TFLOP/s = 64 FLOP/clk/SIMD.
//
// Tiling: (64*NI) x (64*NJ) x BK block tile, 256 threads = 4 waves as 2x2, each wave NI x NJ MFMA tiles
// of 32x32.  Operands are staged global -> registers -> LDS with the next tile's global loads in
// flight during the MFMAs of the current one; BK = 64 for the 128x128 tile (a 32-deep MFMA phase is
// shorter than the ~2 us HBM latency of the prefetch - measured +18 %), 32 for the narrow tiles.
// Two LDS images:
//   KC ("k-contiguous", source rows run along k): T[ext][BK+4]; a lane fetches 4 consecutive k
//       with one ds_read_b128 and feeds 4 MFMAs (pitch BK+4 floats is conflict-free for b128).
//   MC ("mn-contiguous", source rows run along m/n): T[BK][ext+4]; one ds_read_b32 per MFMA,
//       32 consecutive floats per half-wave (conflict-free).
// Forward / dgrad 128x128 tiles stage C through the idle operand LDS so global stores (and the dgrad's
// activation-mask loads) are full-row float4 accesses.
// Within an 8-deep k block q, MFMA step t contracts k = 8q + 4*(lane>>5) + t for BOTH operands
// (any consistent permutation of k is legal), which is what makes the b128 fetch possible.
// (Round 2 tried the analogous trick on the mn axis for the MC operands - sub-tile t of a wave takes rows 2*lane + t, so
// that the two sub-tiles' values of one k come from ONE ds_read_b64 instead of two ds_read_b32; parity-green, but the big
// wgrad went 1.89 -> 1.98 ms and the forward 0.74 -> 0.76 ms: reverted.)
//
// Activation matrices are FEATURE-MAJOR in HBM ([F][B][width]: each group's operand is a dense
// row-major matrix; group offsets therefore scale with the batch: off = x_off + x_boff * batch),
// so tile rows are contiguous 128..512-byte runs instead of the 32 KB-strided slices a
// sample-major [B, F*width] layout (the reference's tf.split view, models.py:101) would give.
#pragma once
#include "dib_common.h"

typedef float dib_f32x16 __attribute__((ext_vector_type(16)));

struct DibGemmGroup {
  long long a_off, b_off, c_off, bias_off, aux_off;  // element offsets into the base pointers
  long long a_boff, b_boff, c_boff, aux_boff;        // + boff * batch (feature-major activations)
  int M, N, K;                                       // -1 => "batch" (runtime kernel argument)
  int lda, ldb, ldc, ldaux;
  int flags;                                         // reserved
};

template <bool KC, int EXT, int BK>
struct DibStage {
  static constexpr int NP = EXT * BK / 1024;       // float4 per thread per tile (256 threads)
  static constexpr int KC_PITCH = BK + 4;
  static constexpr int MC_PITCH = EXT + 4;
  static constexpr int FLOATS = KC ? EXT * KC_PITCH : BK * MC_PITCH;
  // KC: element (mn,k) = base[(mn0+mn)*ld + k0+k]   thread: mn = tid/(BK/4) + (1024/BK)p, k = 4*(tid%(BK/4))
  // MC: element (k,mn) = base[(k0+k)*ld + mn0+mn]   thread: k = tid/(EXT/4) + (1024/EXT)p, mn = 4*(tid%(EXT/4))
  static __device__ __forceinline__ int row(int tid, int p) {
    return KC ? (tid / (BK / 4) + (1024 / BK) * p) : (tid / (EXT / 4) + (1024 / EXT) * p);
  }
  static __device__ __forceinline__ int col(int tid) { return KC ? ((tid % (BK / 4)) * 4) : ((tid % (EXT / 4)) * 4); }

  // loads float4 p in [P0, P1) of the tile (the whole tile by default; the weight gradients issue it in pieces)
  // nt: non-temporal loads (a streamed operand that nobody re-reads soon: keeps it out of the way of what IS re-read)
  template <int P0 = 0, int P1 = NP>
  static __device__ __forceinline__ void gload(float4 (&r)[NP], const float* __restrict__ base, long long ld,
                                               int mn0, int mn_max, int k0, int k_max, bool vec, int tid, bool nt = false) {
    const int r0 = KC ? mn0 : k0, c0 = KC ? k0 : mn0;
    const int Rmax = KC ? mn_max : k_max, Cmax = KC ? k_max : mn_max;
    const int rext = KC ? EXT : BK, cext = KC ? BK : EXT;
    if (vec && r0 + rext <= Rmax && c0 + cext <= Cmax) {  // interior tile: unconditional 16 B loads
      if (nt) {   // block-uniform
        typedef float f4nt __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int p = P0; p < P1; ++p) {
          const f4nt v = __builtin_nontemporal_load(reinterpret_cast<const f4nt*>(base + (long long)(r0 + row(tid, p)) * ld + c0 + col(tid)));
          r[p] = make_float4(v.x, v.y, v.z, v.w);
        }
      } else {
#pragma unroll
        for (int p = P0; p < P1; ++p)
          r[p] = *reinterpret_cast<const float4*>(base + (long long)(r0 + row(tid, p)) * ld + c0 + col(tid));
      }
    } else {
#pragma unroll
      for (int p = P0; p < P1; ++p) {
        const int R = r0 + row(tid, p), Cc = c0 + col(tid);
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
  }
  static __device__ __forceinline__ void lstore(float* __restrict__ T, const float4 (&r)[NP], int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      *reinterpret_cast<float4*>(T + row(tid, p) * (KC ? KC_PITCH : MC_PITCH) + col(tid)) = r[p];
  }
  // the 4 operand values (MFMA steps t=0..3 of k-block q) for the 32-wide sub-tile at mn_base
  static __device__ __forceinline__ float4 frag(const float* __restrict__ T, int mn_base, int q, int l31, int h) {
    if (KC) {
      return *reinterpret_cast<const float4*>(T + (mn_base + l31) * KC_PITCH + q * 8 + h * 4);
    } else {
      const float* p = T + (q * 8 + h * 4) * MC_PITCH + mn_base + l31;
      return make_float4(p[0], p[MC_PITCH], p[2 * MC_PITCH], p[3 * MC_PITCH]);
    }
  }
};

#define DIB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// FLAT (round 6, weight gradients of a 32-row operand only - the set transformer's q / k / v kernels [32, heads x key_dim]): the
// four waves side by side, a 32 x 256 tile of 32 x 64 wave tiles.  On the 64 x 128 tile two of the four waves owned rows that
// do not exist (M = 32): 18 such groups per step at the notebook's size ran at 41 TFLOP/s.
template <int MODE, int NI, int NJ, int BK, bool FLAT = false>
__global__ void __launch_bounds__(256, 2)
dib_gemm_kernel(const DibGemmGroup* __restrict__ groups, const float* __restrict__ Abase,
                const float* __restrict__ Bbase, float* __restrict__ Cbase, const float* __restrict__ bias,
                const float* __restrict__ aux, float* __restrict__ bias_out, int batch, int act, int tiles_m,
                int tiles_n, int rows_per_split, long long split_stride, int stream_flags = 0) {
  static_assert(!FLAT || (MODE == 2 && NI == 1 && NJ == 2), "the flat wave layout is a weight-gradient tile of 32 x (4 x 64)");
  constexpr bool A_KC = (MODE != 2);
  constexpr bool B_KC = (MODE == 1);
  constexpr int BM = FLAT ? 32 : 64 * NI, BN = FLAT ? 256 : 64 * NJ;
  using SA = DibStage<A_KC, BM, BK>;
  using SB = DibStage<B_KC, BN, BK>;
  __shared__ __attribute__((aligned(16))) float smem[SA::FLOATS + SB::FLOATS];
  float* As = smem;
  float* Bs = smem + SA::FLOATS;

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
    // XCD-aware tile order: the dispatcher places consecutive workgroup ids on consecutive XCDs
    // (id % 8); give each XCD its own m-tiles and walk all n-tiles of an m-tile back to back on that
    // XCD so the A tile is served from its L2 for every n-tile after the first.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tn = slot % tiles_n;
    tm = (slot / tiles_n) * 8 + xcd;
    kbeg = 0;
    kend = K;
  }
  int m0 = tm * BM, n0 = tn * BN;
  if (tm >= tiles_m || m0 >= M || n0 >= N) return;  // block-uniform
  if (MODE == 2) {
    // A ragged LAST tile of a weight gradient is shifted back to end at the matrix edge (it overlaps its neighbour; both
    // write bit-identical values: the same k order per element): every operand tile is then an interior tile - unconditional
    // 16-byte loads.  With the half-empty 13th m-tile of BASELINE config 4's 1600 x 256 integration layer on the bounds-
    // checked load path, the 38 workgroups that own it were the stragglers of a single-round launch: 0.590 ms against
    // 0.511 ms for the LARGER 1664 x 256 problem on the same 494 workgroups (profiles/r05g_config4_wgrad_m_pitch_split_sweep.txt).
    if (M >= BM && (M & 3) == 0 && m0 + BM > M) m0 = M - BM;
    if (N >= BN && (N & 3) == 0 && n0 + BN > N) n0 = N - BN;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = FLAT ? 0 : wave >> 1, wn = FLAT ? wave : wave & 1;
  const long long aoff = g.a_off + g.a_boff * batch, boff = g.b_off + g.b_boff * batch;
  const bool vecA = ((aoff | (long long)g.lda) & 3) == 0, vecB = ((boff | (long long)g.ldb) & 3) == 0;
  const float* Ag = Abase + aoff;
  const float* Bg = Bbase + boff;
  const bool active = (m0 + wm * 32 * NI < M) && (n0 + wn * 32 * NJ < N);  // wave has real output

  dib_f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;  // MODE 2: column sums of B (bias gradient), only for tm == 0
  const bool do_bias = (MODE == 2) && (bias_out != nullptr) && (g.bias_off >= 0) && (tm == 0);

  float4 ra[SA::NP], rb[SB::NP];
  if (kbeg < kend) {
    SA::gload(ra, Ag, g.lda, m0, M, kbeg, kend, vecA, tid);
    SB::gload(rb, Bg, g.ldb, n0, N, kbeg, kend, vecB, tid);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    SA::lstore(As, ra, tid);
    SB::lstore(Bs, rb, tid);
    __syncthreads();
    // The next K-tile's global loads are issued in PIECES spread over this tile's MFMA phase instead of one burst after the
    // barrier: a workgroup's HBM request stream becomes even, and the tiled kernels' read rate moves from 4.45 TB/s towards
    // what the streaming kernels reach.  Measured per shape (same-box A/B of 2 / 4 pieces, weight gradients only / every mode,
    // profiles/r03r_gemm_prefetch_pieces_ab.txt): 64-deep forward / dgrad tiles want 4 pieces (dgrad 0.79 -> 0.73 ms), the
    // weight gradients and the 32-deep tiles 2 (layer-3 wgrad 0.70 -> 0.65 ms; 4 pieces there: 0.71).
    constexpr int kPieces = (MODE != 2 && BK == 64) ? 4 : 2;   // (1 = rounds 1-2: the whole prefetch at the top of the MFMA phase)
    // stream_flags bit 0 (set by the host for LARGE streamed operands, launch_gemm_t): the streamed operands - both of a
    // weight gradient, the activation matrix of a forward / dgrad - are loaded non-temporally; bit 1: the forward / dgrad
    // output is stored non-temporally.  Same-box A/Bs at B = 65536 (profiles/r03v_gemm_cache_policy_ab.txt): between -3 % and
    // nothing on the step depending on the box, never slower; small shapes re-read their operands from L2 and lose 1-2 % with
    // it, so the host leaves the flags off below 8192 streamed rows.
    const bool kNtA = (stream_flags & 1) != 0, kNtB = (stream_flags & 1) != 0 && MODE == 2;
    constexpr int QN = BK / 8;   // MFMA sub-phases of a K-tile
    const bool have_next = k0 + BK < kend;
    // piece i of the next tile's global loads (kPieces == 2: A | B; 4: A lo | A hi | B lo | B hi) is issued at sub-phase
    // i * QN / kPieces of this tile's MFMAs
    auto prefetch_piece = [&](auto piece_c) {
      constexpr int piece = decltype(piece_c)::value;
      if (!have_next) return;
      if constexpr (kPieces == 1) {
        SA::gload(ra, Ag, g.lda, m0, M, k0 + BK, kend, vecA, tid, kNtA);
        SB::gload(rb, Bg, g.ldb, n0, N, k0 + BK, kend, vecB, tid, kNtB);
      } else if constexpr (kPieces == 2) {
        if constexpr (piece == 0) SA::gload(ra, Ag, g.lda, m0, M, k0 + BK, kend, vecA, tid, kNtA);
        else SB::gload(rb, Bg, g.ldb, n0, N, k0 + BK, kend, vecB, tid, kNtB);
      } else {
        constexpr int HA = (SA::NP + 1) / 2, HB = (SB::NP + 1) / 2;
        if constexpr (piece == 0) SA::template gload<0, HA>(ra, Ag, g.lda, m0, M, k0 + BK, kend, vecA, tid, kNtA);
        else if constexpr (piece == 1) SA::template gload<HA, SA::NP>(ra, Ag, g.lda, m0, M, k0 + BK, kend, vecA, tid, kNtA);
        else if constexpr (piece == 2) SB::template gload<0, HB>(rb, Bg, g.ldb, n0, N, k0 + BK, kend, vecB, tid, kNtB);
        else SB::template gload<HB, SB::NP>(rb, Bg, g.ldb, n0, N, k0 + BK, kend, vecB, tid, kNtB);
      }
    };
    prefetch_piece(std::integral_constant<int, 0>{});
    if (do_bias) {
      constexpr int PARTS = 256 / BN, RPP = BK / PARTS;
      const int colb = tid % BN, part = tid / BN;
#pragma unroll
      for (int r = 0; r < RPP; ++r) bsum += Bs[(part * RPP + r) * SB::MC_PITCH + colb];
    }
    if (active) {  // rows/cols/k beyond the matrix edge are zero-filled in LDS, so all BK/2 k-steps always run
      // (Round 4 tried issuing the LDS fragment reads of k-block q + 1 before the 16 MFMAs of k-block q - an explicit one-block
      // software pipeline, sched_barrier-pinned, +24 VGPRs, still 2 workgroups per CU: every GEMM shape 1-3 % SLOWER on a
      // same-box A/B, profiles/r04d_gemm_frag_prefetch_ab.txt.  The compiler's own placement - reads hoisted between the MFMAs
      // of the previous block where registers allow - plus the second wave on the SIMD already cover that latency.)
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        if constexpr (kPieces > 1) {
          if (q > 0 && (q * kPieces) % QN == 0) {   // q = i * QN / kPieces, i = 1 .. kPieces - 1 (compile-time after unrolling)
            __builtin_amdgcn_sched_barrier(0);
            const int piece = q * kPieces / QN;
            if (piece == 1) prefetch_piece(std::integral_constant<int, 1>{});
            else if (piece == 2) prefetch_piece(std::integral_constant<int, 2>{});
            else if (piece == 3) prefetch_piece(std::integral_constant<int, 3>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        float4 a[NI], b[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = SA::frag(As, wm * 32 * NI + i * 32, q, l31, h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = SB::frag(Bs, wn * 32 * NJ + j * 32, q, l31, h);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc[i][j] = DIB_MFMA(a[i].x, b[j].x, acc[i][j]);
            acc[i][j] = DIB_MFMA(a[i].y, b[j].y, acc[i][j]);
            acc[i][j] = DIB_MFMA(a[i].z, b[j].z, acc[i][j]);
            acc[i][j] = DIB_MFMA(a[i].w, b[j].w, acc[i][j]);
          }
      }
    } else if constexpr (kPieces > 1) {   // a wave without output still stages its share of the next tile
      prefetch_piece(std::integral_constant<int, 1>{});
      if constexpr (kPieces == 4) {
        prefetch_piece(std::integral_constant<int, 2>{});
        prefetch_piece(std::integral_constant<int, 3>{});
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  constexpr bool LDS_EPILOGUE = (MODE != 2) && (NI == 2) && (NJ == 2) && (SA::FLOATS + SB::FLOATS >= 128 * 132);
  if (LDS_EPILOGUE) {
    // Big forward / dgrad tiles: stage the 128x128 C tile through the (now idle) operand LDS so that global stores -
    // and the dgrad's activation-mask loads - are 16-byte accesses covering full 512-byte rows, instead of 64 scalar
    // 4-byte stores per lane (measured: the 537 MB integration dgrad output ran at 0.7 TB/s with scalar stores).
    constexpr int CP = 132;
    if (active) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            smem[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CP + wn * 64 + j * 32 + l31] = acc[i][j][r];
    }
    __syncthreads();
    float* Cg = Cbase + g.c_off + g.c_boff * batch;
    const float* auxg = (MODE == 1 && aux != nullptr && act != 0) ? aux + g.aux_off + g.aux_boff * batch : nullptr;
    const int c4 = (tid & 31) * 4, colc = n0 + c4;
    const bool vecC = (((g.c_off + g.c_boff * batch) | (long long)g.ldc) & 3) == 0 && colc + 3 < N;
    const bool vecX = auxg != nullptr && (((g.aux_off + g.aux_boff * batch) | (long long)g.ldaux) & 3) == 0;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0 && bias != nullptr && g.bias_off >= 0) {
      if (colc + 0 < N) bv.x = bias[g.bias_off + colc + 0];
      if (colc + 1 < N) bv.y = bias[g.bias_off + colc + 1];
      if (colc + 2 < N) bv.z = bias[g.bias_off + colc + 2];
      if (colc + 3 < N) bv.w = bias[g.bias_off + colc + 3];
    }
#pragma unroll 4
    for (int pss = 0; pss < 16; ++pss) {
      const int rloc = (tid >> 5) + 8 * pss, rowc = m0 + rloc;
      if (rowc >= M || colc >= N) continue;
      float4 v = *reinterpret_cast<const float4*>(smem + rloc * CP + c4);
      if (MODE == 0) {
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (act != 0) { v.x = dib_act(act, v.x); v.y = dib_act(act, v.y); v.z = dib_act(act, v.z); v.w = dib_act(act, v.w); }
      } else if (auxg != nullptr) {
        const float* ap = auxg + (long long)rowc * g.ldaux + colc;
        float4 x;
        if (vecX && colc + 3 < N) x = *reinterpret_cast<const float4*>(ap);
        else {
          x.x = ap[0];
          x.y = colc + 1 < N ? ap[1] : 0.f;
          x.z = colc + 2 < N ? ap[2] : 0.f;
          x.w = colc + 3 < N ? ap[3] : 0.f;
        }
        v.x *= dib_act_grad(act, x.x); v.y *= dib_act_grad(act, x.y); v.z *= dib_act_grad(act, x.z); v.w *= dib_act_grad(act, x.w);
      }
      float* cp = Cg + (long long)rowc * g.ldc + colc;
      if (vecC) {
        if (stream_flags & 2) {
          typedef float f4nt __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(f4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f4nt*>(cp));
        } else {
          *reinterpret_cast<float4*>(cp) = v;
        }
      }
      else {
        cp[0] = v.x;
        if (colc + 1 < N) cp[1] = v.y;
        if (colc + 2 < N) cp[2] = v.z;
        if (colc + 3 < N) cp[3] = v.w;
      }
    }
  } else
  if (active) {
    float* Cg = Cbase + g.c_off + g.c_boff * batch + (MODE == 2 ? (long long)blockIdx.x * split_stride : 0ll);
    const float* auxg = (MODE == 1 && aux != nullptr && act != 0) ? aux + g.aux_off + g.aux_boff * batch : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int colc = n0 + wn * 32 * NJ + j * 32 + l31;
        if (colc < N) {
          float bv = 0.f;
          if (MODE == 0 && bias != nullptr && g.bias_off >= 0) bv = bias[g.bias_off + colc];
          const int rbase = m0 + wm * 32 * NI + i * 32 + 4 * h;
          if (MODE == 0 && act == 1) {  // ReLU fast path (the reference default, train.py:37)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rowc = rbase + (r & 3) + 8 * (r >> 2);
              if (rowc < M) Cg[(long long)rowc * g.ldc + colc] = fmaxf(acc[i][j][r] + bv, 0.f);
            }
          } else if (MODE == 1 && act == 1 && auxg != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rowc = rbase + (r & 3) + 8 * (r >> 2);
              if (rowc < M)
                Cg[(long long)rowc * g.ldc + colc] =
                    auxg[(long long)rowc * g.ldaux + colc] > 0.f ? acc[i][j][r] : 0.f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rowc = rbase + (r & 3) + 8 * (r >> 2);
              if (rowc < M) {
                float v = acc[i][j][r];
                if (MODE == 0) v = dib_act(act, v + bv);
                else if (MODE == 1 && auxg != nullptr) v *= dib_act_grad(act, auxg[(long long)rowc * g.ldaux + colc]);
                Cg[(long long)rowc * g.ldc + colc] = v;
              }
            }
          }
        }
      }
    }
  }
  if (do_bias) {
    constexpr int PARTS = 256 / BN;
    __syncthreads();
    smem[tid] = bsum;
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float s = smem[tid];
#pragma unroll
      for (int p = 1; p < PARTS; ++p) s += smem[tid + p * BN];
      bias_out[g.bias_off + (long long)blockIdx.x * split_stride + n0 + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny-K streaming GEMM (K <= 32): C[M,N] = A[M,K] @ B (+ bias) for the projections of the set transformer whose
// contraction is the 32-wide residual stream and whose output is hundreds of MB (q / k / v at 16 384 tokens: 302 MB; the
// gradient of the attention context: 100 MB).  Such a launch is bound by its output stores; in the tiled kernel above a
// workgroup is a load -> 32-deep product -> 64 KB store sequence with two workgroups per CU to overlap them (1.9 TB/s).
// Here a WAVE is the unit: its 32 columns of B live in registers for the whole launch, A rows are fetched straight into
// MFMA operand layout (4 consecutive k per lane: one 16-byte load), a 64 x 32 output block is 32 MFMAs and 32 store
// instructions of two full 128-byte lines each; no LDS, no barrier, 16 waves per CU in flight.
//   MODE 0: B = [K, N] (Keras kernel, Dense forward), bias added;   MODE 1: B = [N, K] (Dense dgrad: C = A @ B^T).
// grid (row chunks, ceil(N / 128), groups); wave w of a workgroup owns columns n0 + 32w ..; a chunk is `tiles` 64-row tiles.
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
dib_gemm_skinnyk_kernel(const DibGemmGroup* __restrict__ groups, const float* __restrict__ Abase,
                        const float* __restrict__ Bbase, float* __restrict__ Cbase, const float* __restrict__ bias,
                        int M, int N, int K, int tiles, int nt_store) {
  const DibGemmGroup g = groups[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 128 + wave * 32;
  if (n0 >= N) return;   // whole wave (N is a multiple of 32)
  const float* A = Abase + g.a_off;
  const float* B = Bbase + g.b_off;
  float* C = Cbase + g.c_off;
  const int col = n0 + l31;
  // 16-byte accesses along k need aligned rows (block-uniform; the set transformer's buffers are)
  const bool vec_a = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool vec_b = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  // this lane's column of B: bf[q] = B[k = 8q + 4h + t][col], t = x..w (zero beyond K)
  float4 bf[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = 8 * q + 4 * h;
    bf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) {   // K % 4 == 0
      if (MODE == 1) {
        const float* p = B + (long long)col * g.ldb + k;
        bf[q] = vec_b ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
      } else {
        const float* p = B + (long long)k * g.ldb + col;
        bf[q] = make_float4(p[0], p[g.ldb], p[2 * (long long)g.ldb], p[3 * (long long)g.ldb]);
      }
    }
  }
  const float bv = (MODE == 0 && bias != nullptr && g.bias_off >= 0) ? bias[g.bias_off + col] : 0.f;
  const int row_first = blockIdx.x * tiles * 64;
  const unsigned lda = (unsigned)g.lda;
  // A rows of one 64-row tile in operand layout: af[s][q] = A[row0 + 32s + l31][8q + 4h ..+3] (rows clamped, k >= K -> 0)
  auto load_tile = [&](float4 (&af)[2][4], int row0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* src = A + (long long)min(row0 + 32 * s + l31, M - 1) * lda + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[s][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (8 * q + 4 * h < K) {
          const float* p = src + 8 * q;
          af[s][q] = vec_a ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
        }
      }
    }
  };
  float4 cur[2][4], nxt[2][4];
  load_tile(cur, row_first);
  for (int t = 0; t < tiles; ++t) {
    const int row0 = row_first + t * 64;
    if (row0 >= M) break;
    if (t + 1 < tiles) load_tile(nxt, min(row0 + 64, M - 1));   // in flight during this tile's products and stores
    dib_f32x16 acc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = bv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // the two row sub-tiles alternate: neighbouring MFMAs are independent
      acc[0] = DIB_MFMA(cur[0][q].x, bf[q].x, acc[0]);
      acc[1] = DIB_MFMA(cur[1][q].x, bf[q].x, acc[1]);
      acc[0] = DIB_MFMA(cur[0][q].y, bf[q].y, acc[0]);
      acc[1] = DIB_MFMA(cur[1][q].y, bf[q].y, acc[1]);
      acc[0] = DIB_MFMA(cur[0][q].z, bf[q].z, acc[0]);
      acc[1] = DIB_MFMA(cur[1][q].z, bf[q].z, acc[1]);
      acc[0] = DIB_MFMA(cur[0][q].w, bf[q].w, acc[0]);
      acc[1] = DIB_MFMA(cur[1][q].w, bf[q].w, acc[1]);
    }
    // acc[s][r] = C[row0 + 32s + (r&3) + 8(r>>2) + 4h][col]: per store instruction two rows x 32 consecutive columns.
    // Interior tiles (workgroup-uniform test) store unconditionally; nt_store (uniform): an output too large to stay in the
    // infinity cache for its consumer.
    auto store_tile = [&](auto interior_c, auto nt_c) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float* dst = C + (long long)(row0 + 32 * s + 4 * h) * g.ldc + col;
        const int rlim = M - (row0 + 32 * s + 4 * h);   // rows of this lane's group that exist
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (decltype(interior_c)::value || dr < rlim) {
            if (decltype(nt_c)::value) __builtin_nontemporal_store(acc[s][r], dst + (long long)dr * g.ldc);
            else dst[(long long)dr * g.ldc] = acc[s][r];
          }
        }
      }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    if (row0 + 64 <= M) {
      if (nt_store) store_tile(T_{}, T_{}); else store_tile(T_{}, F_{});
    } else {
      if (nt_store) store_tile(F_{}, T_{}); else store_tile(F_{}, F_{});
    }
    if (t + 1 < tiles) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[s][q] = nxt[s][q];
    }
  }
}
