// dib_attn.h - flash-style self-attention over the particle axis for the per-particle Distributed-IB set transformer
// (reference notebook ...per_particle_measurements_and_set_transformer.ipynb, cell 8:
// `tf.keras.layers.MultiHeadAttention(number_heads_per_mha, key_dim)(x, x, x)`, 12 heads x key_dim 128; BASELINE config 5
// asks for 4096 particles per neighbourhood).  Exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Keys / values stream through LDS in tiles of 32, softmax is computed online; the [P, P] PROBABILITIES never exist in HBM.
// The backward pass needs the scores again.  Two modes (the caller chooses by passing a stash buffer or NULL):
//   recompute  S = (scale Q) K^T is evaluated a second time in the backward from q, k and the per-query log-sum-exp: no
//              extra memory, 5 tile products per (key tile, query tile) pair for 4 algorithmic ones;
//   stash      the forward writes the raw score tiles (before the softmax) to HBM, tile-major [b][h][key tile][query tile]
//              [32 queries][32 keys], and the backward reads them back: 4 products.  At 4 x 4096 particles x 12 heads that is
//              3.2 GB per attention block (19 GB for the notebook's six - MI355X has 288 GB), written once with
//              16-byte stores and read once with fully coalesced (non-temporal) dword loads, both in the shadow of the MFMAs;
//              it takes 64 of the 320 MFMAs, the Q / K operand fetches of the S product (32 ds_read_b128) out of every
//              backward tile.  Same-box A/B: profiles/r03c_attention_stash_ab.txt.
//
//   forward   dib_attn_fwd_kernel : one wave = 32 queries (workgroup = 128 queries of one (neighbourhood, head)).
//             S^T = K Q^T is evaluated TRANSPOSED (rows = keys, columns = queries): lane (j, h) then holds 16 keys of ONE
//             query j, so the row maximum / sum are register reductions plus one cross-half shuffle, and the probability
//             tile P^T is - register for register - the B operand of O^T += V^T P^T (contraction index = key
//             (r&3) + 8(r>>2) + 4h = the C-fragment row of register r): probabilities never touch LDS.
//   backward  dib_attn_bwd_kernel (one wave = 32 keys, one wave per SIMD with the whole 512-register file, loops over query tiles; S = Q K^T evaluated UNtransposed so that
//             P and dS are the B operands of dV^T += dO^T P and dK^T += Q^T dS, contraction index = query; the dQ
//             contribution of the workgroup's 128 keys goes through an LDS transpose of dS into a per-key-block partial
//             buffer) + dib_attn_dq_reduce_kernel: no atomics, one writer per element, fixed summation order.
//   delta     dib_attn_delta_kernel : delta[q] = sum_d dO[q][d] O[q][d]
//
// Layout: q, k, v, o and their gradients are [tokens, ld] row-major with head h at columns [h*128, (h+1)*128)
// (ld = heads * 128), token = neighbourhood * P + particle; lse / delta are [neighbourhood][head][P].
// Operand fetch follows dib_gemm.h: "KC" = a [rows][128+4] LDS image read along k with one ds_read_b128 per 4 MFMAs, "MC"
// = the same image read along rows with ds_read_b32; MFMA step t of k-block q contracts k = 8q + 4*(lane>>5) + t.
#pragma once
#include <type_traits>
#include "dib_common.h"
#include "dib_gemm.h"

// Order pins.  MFMAs are pure register instructions: neither __builtin_amdgcn_sched_barrier nor program order keeps the
// instruction selector's list scheduler from floating them across a whole phase (it once sank all 64 S products of the
// backward behind the 64 dP products, with 128 registers of operand fragments live).  Passing an accumulator through an
// empty volatile asm makes it opaque there: its producers stay above, its consumers below, and volatile asms keep their
// own order.  "a" = accumulator (AGPR) operands (the 512-register backward), "v" = values kept in VGPRs.
#define DIB_PIN_ACC_A(x) asm volatile("" : "+a"(x))
#define DIB_PIN_ACC_V(x) asm volatile("" : "+v"(x))

typedef float dib_nt4a __attribute__((ext_vector_type(4)));  // native vector type for non-temporal 16-byte stores
constexpr int kAttnD = 128;          // key_dim (= value dim) of the notebook's MultiHeadAttention
constexpr int kAttnPitch = kAttnD + 4;
constexpr int kAttnTile = 32;        // keys (fwd, dq) / queries (dkv) per LDS tile

struct DibAttnArgs {
  const float* q; const float* k; const float* v;   // [T, ld]
  float* o;                                         // fwd out [T, ld]
  float* lse;                                       // [B][H][P]  fwd out / bwd in
  const float* d_o;                                 // bwd: dL/do [T, ld]
  const float* delta;                               // bwd: [B][H][P]
  float* dq; float* dk; float* dv;                  // bwd out [T, ld]
  float* s_stash;                                   // fwd out / bwd in (NULL: recompute), see dib_attn_stash_tile
  int P, H; long long ld; float scale;
  // round 6, dib_attn_small_fwd_kernel<true> only: q, k, v are OUTPUTS - the head's slices of the three input projections
  // x [T, 32] @ W_i [32, H * 128] + b_i are computed in the kernel's prologue (and written for the backward)
  const float* px; long long pldx; const float* pparams; long long pw[3], pb[3]; float* pq; float* pk; float* pv;
  // round 6, dib_attn_small_bwd8_kernel<true> only: the head's share of the projections' INPUT gradient,
  // dq_h W_q[:, head]^T + dk_h W_k[:, head]^T + dv_h W_v[:, head]^T [P, 32], written to slab (1 + head) of pdx (pdx_stride floats apart)
  float* pdx; long long pdx_stride;
};

// first element of the 32 x 32 score tile (key tile kt, query tile qt) of (neighbourhood b, head): row-major [query][key]
__device__ __forceinline__ long long dib_attn_stash_tile(int b, int H, int head, int n_tiles, int kt, int qt) {
  return ((((long long)b * H + head) * n_tiles + kt) * n_tiles + qt) * (long long)(kAttnTile * kAttnTile);
}

// 32 rows x 128 floats of a [T, ld] matrix (rows row0.. clamped to row_max) -> registers (4 float4 per thread, 256 threads).
// Passed BY VALUE as a struct of four named float4: as `float4 (&)[4]` one of the two tiles of the backward ended up as a
// 64-byte stack object - every global load was followed by `s_waitcnt vmcnt; scratch_store` a few instructions after issue.
// Row offsets are 32-bit (the host entry checks P * ld < 2^30 elements): one v_mul_lo_u32 per row instead of a 64-bit multiply.
struct DibAttnTile { float4 r0, r1, r2, r3; };
__device__ __forceinline__ DibAttnTile dib_attn_gload(const float* __restrict__ base, long long ld, int row0, int row_max,
                                                      int tid) {
  const unsigned ldu = (unsigned)ld, col = (unsigned)(tid & 31) * 4u;
  const int r = row0 + (tid >> 5);
  DibAttnTile t;
  t.r0 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r, row_max) * ldu + col));
  t.r1 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r + 8, row_max) * ldu + col));
  t.r2 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r + 16, row_max) * ldu + col));
  t.r3 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r + 24, row_max) * ldu + col));
  return t;
}
__device__ __forceinline__ void dib_attn_lstore(float* __restrict__ T, const DibAttnTile t, int tid, float mul = 1.0f) {
  float* dst = T + (tid >> 5) * kAttnPitch + (tid & 31) * 4;
  *reinterpret_cast<float4*>(dst) = make_float4(t.r0.x * mul, t.r0.y * mul, t.r0.z * mul, t.r0.w * mul);
  *reinterpret_cast<float4*>(dst + 8 * kAttnPitch) = make_float4(t.r1.x * mul, t.r1.y * mul, t.r1.z * mul, t.r1.w * mul);
  *reinterpret_cast<float4*>(dst + 16 * kAttnPitch) = make_float4(t.r2.x * mul, t.r2.y * mul, t.r2.z * mul, t.r2.w * mul);
  *reinterpret_cast<float4*>(dst + 24 * kAttnPitch) = make_float4(t.r3.x * mul, t.r3.y * mul, t.r3.z * mul, t.r3.w * mul);
}
// KC fragment: 4 consecutive k of row (l31) for k-block q   |   MC fragment: rows 8q+4h+t (t = 0..3), column c
__device__ __forceinline__ float4 dib_attn_kc(const float* __restrict__ T, int q, int l31, int h) {
  return *reinterpret_cast<const float4*>(T + l31 * kAttnPitch + q * 8 + h * 4);
}
__device__ __forceinline__ float4 dib_attn_mc(const float* __restrict__ T, int q, int c, int h) {
  const float* p = T + (q * 8 + h * 4) * kAttnPitch + c;
  return make_float4(p[0], p[kAttnPitch], p[2 * kAttnPitch], p[3 * kAttnPitch]);
}
// the 32 x 128 row block of one token range held as the B operand of a transposed product: lane (j, h) keeps
// X[row j][8q + 4h + t] in f[q] (t = x, y, z, w)
__device__ __forceinline__ void dib_attn_rowfrag(float4 (&f)[16], const float* __restrict__ base, long long ld, int row,
                                                 int h, float mul) {
  const float* src = base + (long long)row * ld + 4 * h;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float4 v = *reinterpret_cast<const float4*>(src + 8 * q);
    f[q] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
  }
}
// store a transposed accumulator (lane = row j of the output, registers = 128 columns) as row-major [row][128]
__device__ __forceinline__ void dib_attn_store_rows(float* __restrict__ base, long long ld, int row, bool ok, int h,
                                                    const dib_f32x16 (&acc)[4], float mul) {
  if (!ok) return;
  float* dst = base + (long long)row * ld + 4 * h;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + 32 * dt + 8 * g) =
          make_float4(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(P / 128), H, B), 256 threads
// ---------------------------------------------------------------------------------------------------------------------
// 2 waves / SIMD: measured (tools/attn_bench.py, 4 x 4096 x 12 heads, same box) 3.16-3.19 ms = 130 TFLOP/s = 0.83 of peak with
// 206 registers and no spills; 3 waves / SIMD (168 registers, 36 spilled) 3.78-3.80 ms.  (Before the by-value tile staging the
// order was the other way round, 4.98 vs 4.26 ms: the stack-object traffic of the prefetched tile hurt the 2-wave build more.)
// 8-wave forward (round 6): 256 queries share ONE staged K / V tile - half the L2 -> LDS traffic and half the LDS
// stores per query of two 4-wave workgroups; 512 threads stage a 32 x 128 tile as 2 float4 each
struct DibAttnTile2 { float4 r0, r1; };
__device__ __forceinline__ DibAttnTile2 dib_attn_gload2(const float* __restrict__ base, long long ld, int row0, int row_max, int tid) {
  const unsigned ldu = (unsigned)ld, col = (unsigned)(tid & 31) * 4u;
  const int r = row0 + (tid >> 5);   // 0 .. 15
  DibAttnTile2 t;
  t.r0 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r, row_max) * ldu + col));
  t.r1 = *reinterpret_cast<const float4*>(base + ((unsigned)min(r + 16, row_max) * ldu + col));
  return t;
}
__device__ __forceinline__ void dib_attn_lstore2(float* __restrict__ T, const DibAttnTile2 t, int tid) {
  float* dst = T + (tid >> 5) * kAttnPitch + (tid & 31) * 4;
  *reinterpret_cast<float4*>(dst) = t.r0;
  *reinterpret_cast<float4*>(dst + 16 * kAttnPitch) = t.r1;
}
template <int WAVES>   // 4: 128 queries per workgroup, two workgroups per CU; 8: 256 queries, one workgroup per CU (the same 2 waves per SIMD)
__device__ __forceinline__ void dib_attn_fwd_body(const DibAttnArgs& a) {
  __shared__ __attribute__((aligned(16))) float Ks[kAttnTile * kAttnPitch];
  __shared__ __attribute__((aligned(16))) float Vs[kAttnTile * kAttnPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, P = a.P;
  const long long tok0 = (long long)b * P;
  const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
  const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
  const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
  const int qrow = blockIdx.x * (32 * WAVES) + wave * 32 + l31;          // this lane's query
  const bool q_ok = qrow < P;
  const bool wave_ok = blockIdx.x * (32 * WAVES) + wave * 32 < P;        // wave has at least one real query

  float4 qf[16];
  dib_attn_rowfrag(qf, Qb, a.ld, min(qrow, P - 1), h, a.scale);
  dib_f32x16 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (P + kAttnTile - 1) / kAttnTile;
  using Tile = typename std::conditional<WAVES == 8, DibAttnTile2, DibAttnTile>::type;
  auto gload = [&](const float* base, int row0) -> Tile {
    if constexpr (WAVES == 8) return dib_attn_gload2(base, a.ld, row0, P - 1, tid);
    else return dib_attn_gload(base, a.ld, row0, P - 1, tid);
  };
  Tile rk = gload(Kb, 0);
  Tile rv = gload(Vb, 0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    if constexpr (WAVES == 8) { dib_attn_lstore2(Ks, rk, tid); dib_attn_lstore2(Vs, rv, tid); }
    else { dib_attn_lstore(Ks, rk, tid); dib_attn_lstore(Vs, rv, tid); }
    __syncthreads();
    if (kt + 1 < n_tiles) {   // (issuing the V tile's loads after the S product instead - the GEMM's prefetch-piece trick - measured
      rk = gload(Kb, (kt + 1) * kAttnTile);   // no different here: profiles/r03s_*)
      rv = gload(Vb, (kt + 1) * kAttnTile);
    }
    if (wave_ok) {
      // S^T[key][query] = sum_d K[key][d] (scale Q[query][d]); the K fragment of step q + 1 is fetched before the MFMAs of
      // step q are issued (one LDS latency per tile instead of one per 4 MFMAs)
      dib_f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      float4 kk = dib_attn_kc(Ks, 0, l31, h);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 kn = dib_attn_kc(Ks, q < 15 ? q + 1 : 15, l31, h);
        s = DIB_MFMA(kk.x, qf[q].x, s);
        s = DIB_MFMA(kk.y, qf[q].y, s);
        s = DIB_MFMA(kk.z, qf[q].z, s);
        s = DIB_MFMA(kk.w, qf[q].w, s);
        kk = kn;
      }
      if (a.s_stash != nullptr) {
        // raw scores for the backward (stash mode): lane (query j, h) owns keys 8g + 4h + {0..3} of its query - one 16-byte
        // store per register group (plain: the two 16-byte halves of a 32-byte sector come from two lanes and merge in L2;
        // non-temporal stores measured 1.7 % slower, profiles/r03g_attention_stash_cache_policy_ab.txt); values of keys /
        // queries beyond P are finite and multiplied by 0 there
        float* sp = a.s_stash + dib_attn_stash_tile(b, a.H, head, n_tiles, kt, blockIdx.x * WAVES + wave) + l31 * kAttnTile + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<dib_nt4a*>(sp + 8 * g) = dib_nt4a{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
      }
      // online softmax over this tile's keys (register r <-> key kt*32 + (r&3) + 8(r>>2) + 4h)
      if (kt == n_tiles - 1) {   // only the last tile can hold keys beyond P
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * kAttnTile + (r & 3) + 8 * (r >> 2) + 4 * h;
          s[r] = key < P ? s[r] : -INFINITY;
        }
      }
      float mloc = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // Lazy rescale: the running reference m_run only moves when the tile maximum exceeds it by more than kLazy (then
      // exp(s - m_run) <= e^kLazy, harmless in fp32).  The 64 accumulator registers live in AGPRs, so a rescale is 64 x
      // (read, multiply, write back) in front of the P V MFMAs that depend on them - with the eager form that was paid on
      // every key tile; now on the first tile and the (rare) tiles where some query's maximum jumps.  lse stays exact.
      constexpr float kLazy = 6.0f;
      const bool moved = mloc > m_run + kLazy;         // m_run = -inf on the first tile: always true
      const float m_new = moved ? mloc : m_run;
      if (__any(moved)) {
        const float alpha = __expf(m_run - m_new);     // 1 for the lanes that did not move, 0 on the first tile
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
      }
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __expf(s[r] - m_new);
        psum += s[r];
      }
      l_run += psum;
      // O^T[d][query] += sum_key V[key][d] P^T[key][query]: the four d-tile fragments of the NEXT key block are in flight
      // while the 16 MFMAs of the current one (four independent accumulators) issue
      float4 vv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vv[dt] = dib_attn_mc(Vs, 0, 32 * dt + l31, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 vn[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vn[dt] = dib_attn_mc(Vs, q < 3 ? q + 1 : 3, 32 * dt + l31, h);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          acc[dt] = DIB_MFMA(vv[dt].x, s[4 * q + 0], acc[dt]);
          acc[dt] = DIB_MFMA(vv[dt].y, s[4 * q + 1], acc[dt]);
          acc[dt] = DIB_MFMA(vv[dt].z, s[4 * q + 2], acc[dt]);
          acc[dt] = DIB_MFMA(vv[dt].w, s[4 * q + 3], acc[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vv[dt] = vn[dt];
      }
    }
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  dib_attn_store_rows(a.o + tok0 * a.ld + head * kAttnD, a.ld, qrow, q_ok && wave_ok, h, acc, inv);
  if (q_ok && wave_ok && h == 0) a.lse[((long long)b * a.H + head) * P + qrow] = m_run + __logf(l_tot);
}


__global__ void __launch_bounds__(256, 2)   // workgroups per CU = waves per SIMD (2: <= 256 registers)
dib_attn_fwd_kernel(DibAttnArgs a) { dib_attn_fwd_body<4>(a); }
// P >= 256 (dib_set_tuning "attn_fwd_waves" = 8, the default): bit-identical outputs, 3.34 -> 3.20 - 3.28 ms at 4 x 4096 x 12 heads,
// BASELINE config 5's step 69.9 -> 69.5 ms (profiles/r06ab_attention_forward_8_waves_ab.txt)
__global__ void __launch_bounds__(512, 1)
dib_attn_fwd8_kernel(DibAttnArgs a) { dib_attn_fwd_body<8>(a); }

// delta[b][h][q] = sum_d dO[q][d] * O[q][d] : grid (ceil(T*H / 4)), one wave per (token, head)
__global__ void __launch_bounds__(256)
dib_attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, long long ld, int B, int P, int H,
                      float* __restrict__ delta) {
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)B * P * H;
  if (item >= total) return;
  const int lane = threadIdx.x & 63;
  const long long tok = item / H;
  const int head = (int)(item % H);
  const float* po = o + tok * ld + head * kAttnD;
  const float* pd = d_o + tok * ld + head * kAttnD;
  float s = po[lane] * pd[lane] + po[lane + 64] * pd[lane + 64];
  s = dib_wave_sum(s);
  if (lane == 0) {
    const long long b = tok / P, p = tok % P;
    delta[(b * H + head) * P + p] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: ONE kernel, grid (ceil(P / 128), H, B), 256 threads, dynamic LDS (DibAttnBwdLds floats).
// A workgroup owns 128 keys (one wave = 32 keys, lane = key); query tiles of 32 stream through LDS.  Per query tile:
//   S[query][key] = (scale Q) K^T, P = exp(S - lse[query]), dP = dO V^T, dS = P (dP - delta[query])      (lane = key)
//   dV^T[d][key] += sum_query dO[query][d] P[query][key],  dK^T[d][key] += sum_query (scale Q)[query][d] dS[query][key]
//   dQ contribution of these 128 keys: every wave drops its dS tile TRANSPOSED into an LDS patch; after a barrier wave w
//   computes the d-tile w of  dQ^T[d][query] = sum_{128 keys} K^T[d][key] dS^T[key][query]  (K block resident in LDS) and
//   writes it to the per-key-block partial buffer  part[b][h][key block][query][128].
// dib_attn_dq_reduce_kernel then sums the key-block partials in a fixed order (x scale).  Every gradient element has one
// writer and a fixed summation order (deterministic), S and dP are computed ONCE per tile pair: 5 tile products where the
// separate dQ and dK/dV kernels of the first version needed 7.
// ---------------------------------------------------------------------------------------------------------------------
// Phase timing of the backward (a diagnostic build: -DDIB_ATTN_TIMING; tools/attn_phase_timing.py): wave 0 of workgroup
// (1, 0, 0) accumulates s_memtime deltas per phase of the query-tile loop into dib_attn_dbg.
#ifdef DIB_ATTN_TIMING
__device__ long long dib_attn_dbg[16];
#define DIB_T(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); tacc_[i] += now_ - tprev_; tprev_ = now_; \
                      __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIB_T(i) do { } while (0)
#endif
constexpr int kAttnPatch = 32 * 36;
// dQ partial stores: a wave turns its dQ^T accumulators (lane = query, registers = d) through a private 32 x 36 LDS patch
// into row-major pieces, so that ONE store instruction writes whole 128-byte lines (8 lanes per query row) instead of 32-byte
// pieces of 32 rows: the non-temporal partial stores are then full-line writes (WRITE_SIZE of the launch 6.7 -> 3.4 GB)
constexpr int DibAttnBwdLds = 2 * kAttnTile * kAttnPitch + 128 * kAttnPitch + 4 * kAttnPatch + 2 * kAttnTile +
                              4 * kAttnPatch;

template <bool STASH>   // STASH: scores read back from the forward's stash (4 tile products); else S recomputed (5)
__global__ void __launch_bounds__(256, 1)
dib_attn_bwd_kernel(DibAttnArgs a, float* __restrict__ dq_part, int n_key_blocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                                   // [32][132] scaled Q tile
  float* Gs = Qs + kAttnTile * kAttnPitch;           // [32][132] dO tile
  float* Kblk = Gs + kAttnTile * kAttnPitch;         // [128][132] this workgroup's keys (A operand of the dQ product)
  float* patches = Kblk + 128 * kAttnPitch;          // [4 waves][32 queries][36]: dS^T tiles
  float* Ls = patches + 4 * kAttnPatch;              // lse / delta of the tile's queries
  float* Ds = Ls + kAttnTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  float* my_dqs = Ds + kAttnTile + wave * kAttnPatch;   // this wave's [32 queries][36] transposition patch (nobody else touches it)
  const int head = blockIdx.y, b = blockIdx.z, P = a.P;
  const long long tok0 = (long long)b * P;
  const float* Qb = a.q + tok0 * a.ld + head * kAttnD;
  const float* Kb = a.k + tok0 * a.ld + head * kAttnD;
  const float* Vb = a.v + tok0 * a.ld + head * kAttnD;
  const float* dOb = a.d_o + tok0 * a.ld + head * kAttnD;
  const float* lse_b = a.lse + ((long long)b * a.H + head) * P;
  const float* dlt_b = a.delta + ((long long)b * a.H + head) * P;
  const int krow = blockIdx.x * 128 + wave * 32 + l31;          // this lane's key
  const bool k_ok = krow < P;
  const bool wave_ok = blockIdx.x * 128 + wave * 32 < P;
  const int kc = min(krow, P - 1);
  // V row fragment resident in registers (B operand of dP = dO V^T); the K row fragment (B operand of S = Q K^T) is read
  // from the workgroup's K block in LDS instead - it is there for the dQ product anyway, and 64 registers fewer end the
  // spilling (48 -> 0) and leave room to fetch every LDS fragment one step ahead of its MFMAs (one wave per SIMD: nobody
  // else hides that latency)
  float4 vf[16];
  dib_attn_rowfrag(vf, Vb, a.ld, kc, h, 1.0f);
  // the workgroup's 128 key rows -> LDS (rows beyond P are clamped: their dS is 0)
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int rl = (tid >> 5) + 8 * p;
    const int row = min(blockIdx.x * 128 + rl, P - 1);
    *reinterpret_cast<float4*>(Kblk + rl * kAttnPitch + (tid & 31) * 4) =
        *reinterpret_cast<const float4*>(Kb + (long long)row * a.ld + (tid & 31) * 4);
  }
  dib_f32x16 dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
  float* my_patch = patches + wave * kAttnPatch;
  // dQ output of this workgroup: direct (one key block: P <= 128) or partial buffer slice
  float* dq_out = (n_key_blocks > 1)
                      ? dq_part + ((((long long)b * a.H + head) * n_key_blocks + blockIdx.x) * P) * kAttnD
                      : a.dq + tok0 * a.ld + head * kAttnD;
  const long long dq_ld = (n_key_blocks > 1) ? kAttnD : a.ld;
  const float dq_mul = (n_key_blocks > 1) ? 1.0f : a.scale;

  const int n_tiles = (P + kAttnTile - 1) / kAttnTile;
  float rl_ = 0.f, rd_ = 0.f;
  DibAttnTile rq = dib_attn_gload(Qb, a.ld, 0, P - 1, tid);
  DibAttnTile rg = dib_attn_gload(dOb, a.ld, 0, P - 1, tid);
  if (tid < kAttnTile) { rl_ = tid < P ? lse_b[tid] : INFINITY; rd_ = dlt_b[min(tid, P - 1)]; }
  // a lane whose key lies beyond P contributes nothing: its probabilities are multiplied by 0 (lane-constant factor); a
  // query beyond P carries lse = +inf in the LDS copy, so exp(s - lse) = 0 without a per-element test.  The loop body has no
  // divergent control flow: the loop-carried dV / dK accumulators are only ever touched by MFMAs and stay in AGPRs (with
  // an `if (wave has keys)` around the products the compiler kept them in VGPRs and copied all 128 registers into AGPRs
  // and back on every query tile)
  const float kmul = k_ok ? 1.0f : 0.0f;
  // stage one query tile: registers (global) -> LDS.  The NEXT tile's global loads are issued after the second barrier of
  // a tile and land during the dQ product; they go to LDS right after it (Qs / Gs are free by then).  Holding them in
  // registers across the whole tile instead (the first version) cost 32 registers through the two big phases: the compiler
  // parked them in AGPRs and spilled four float4 to scratch, each spill waiting for its load a few instructions after issue
#define DIB_ATTN_STAGE_TILE()                                   \
  do {                                                          \
    dib_attn_lstore(Qs, rq, tid, a.scale);                      \
    dib_attn_lstore(Gs, rg, tid);                               \
    if (tid < kAttnTile) { Ls[tid] = rl_; Ds[tid] = rd_; }      \
  } while (0)
  DIB_ATTN_STAGE_TILE();
  // stash mode: this wave's row of score tiles [key tile][query tile 0 ..] - lane (key l31, h) reads rows (queries) 4h + ...
  const float* stash_w = nullptr;
  if constexpr (STASH) {
    const int ktw = min((int)blockIdx.x * 4 + wave, n_tiles - 1);   // a wave whose keys all lie beyond P re-reads a real tile (x 0)
    stash_w = a.s_stash + dib_attn_stash_tile(b, a.H, head, n_tiles, ktw, 0) + 4 * h * kAttnTile + l31;
  }
#ifdef DIB_ATTN_TIMING
  long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev_ = clock64();
  const long long tstart_ = tprev_;
  const long long wstart_ = wall_clock64();   // constant 100 MHz
#endif
  for (int qt = 0; qt < n_tiles; ++qt) {
    __syncthreads();
    DIB_T(0);   // barrier A
    // lse / delta of this lane's 16 queries (register r <-> query (r&3) + 8(r>>2) + 4h): 4 + 4 ds_read_b128, in flight
    // during the S / dP products
    float4 lq[4], dq4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      lq[g] = *reinterpret_cast<const float4*>(Ls + 8 * g + 4 * h);
      dq4[g] = *reinterpret_cast<const float4*>(Ds + 8 * g + 4 * h);
    }
    float sv[16];   // S[query r][this lane's key]
    dib_f32x16 dp;
    if constexpr (STASH) {
      // scores from the forward's stash: 16 dword loads, each 2 x 128 contiguous bytes per wave (rows = queries
      // (r&3) + 8(r>>2) + 4h, column = this lane's key); they land during the dP product
      // (fetching tile qt + 1 one tile ahead, during the dQ product, measured 1 % slower: profiles/r03c_attention_stash_ab.txt)
      const float* sp = stash_w + (long long)qt * (kAttnTile * kAttnTile);
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = __builtin_nontemporal_load(sp + ((r & 3) + 8 * (r >> 2)) * kAttnTile);
      // dP[query][key] = dO V^T: A = dO tile rows (KC), B = this lane's V row (registers).  Two accumulators (even / odd
      // k-blocks) so that the LDS fetch of the next step never sits between two MFMAs on the same accumulator
      dib_f32x16 dp1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dp[r] = 0.f; dp1[r] = 0.f; }
      float4 ga = dib_attn_kc(Gs, 0, l31, h), gb = dib_attn_kc(Gs, 1, l31, h);
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        const int qn_ = q < 14 ? q + 2 : 14;
        const float4 gan = dib_attn_kc(Gs, qn_, l31, h), gbn = dib_attn_kc(Gs, qn_ + 1, l31, h);
        __builtin_amdgcn_sched_barrier(0);
        dp = DIB_MFMA(ga.x, vf[q].x, dp);
        dp1 = DIB_MFMA(gb.x, vf[q + 1].x, dp1);
        dp = DIB_MFMA(ga.y, vf[q].y, dp);
        dp1 = DIB_MFMA(gb.y, vf[q + 1].y, dp1);
        dp = DIB_MFMA(ga.z, vf[q].z, dp);
        dp1 = DIB_MFMA(gb.z, vf[q + 1].z, dp1);
        dp = DIB_MFMA(ga.w, vf[q].w, dp);
        dp1 = DIB_MFMA(gb.w, vf[q + 1].w, dp1);
        DIB_PIN_ACC_A(dp);
        DIB_PIN_ACC_A(dp1);
        __builtin_amdgcn_sched_barrier(0);
        ga = gan; gb = gbn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] += dp1[r];
    } else {
      dib_f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      // S[query][key] and dP[query][key]: A = query-tile rows (KC), B = this lane's key row (K from the LDS block, V from
      // registers).  The three LDS fragments of step q + 1 are issued BEFORE the 8 MFMAs of step q (sched_barrier: left to
      // itself the scheduler sinks them to just in front of their use and the wave - alone on its SIMD - eats one LDS
      // latency per step)
      const float* Kw = Kblk + wave * 32 * kAttnPitch;
      float4 qq = dib_attn_kc(Qs, 0, l31, h), gg = dib_attn_kc(Gs, 0, l31, h), kk = dib_attn_kc(Kw, 0, l31, h);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int qn_ = q < 15 ? q + 1 : 15;
        const float4 qn = dib_attn_kc(Qs, qn_, l31, h), gn = dib_attn_kc(Gs, qn_, l31, h), kn = dib_attn_kc(Kw, qn_, l31, h);
        __builtin_amdgcn_sched_barrier(0);
        s = DIB_MFMA(qq.x, kk.x, s);
        dp = DIB_MFMA(gg.x, vf[q].x, dp);
        s = DIB_MFMA(qq.y, kk.y, s);
        dp = DIB_MFMA(gg.y, vf[q].y, dp);
        s = DIB_MFMA(qq.z, kk.z, s);
        dp = DIB_MFMA(gg.z, vf[q].z, dp);
        s = DIB_MFMA(qq.w, kk.w, s);
        dp = DIB_MFMA(gg.w, vf[q].w, dp);
        DIB_PIN_ACC_A(s);
        DIB_PIN_ACC_A(dp);
        __builtin_amdgcn_sched_barrier(0);
        qq = qn; gg = gn; kk = kn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = s[r];
    }
    DIB_T(1);   // S / dP products issued
    // first operand fragments of the dV / dK products: in flight during the exponentials
    float4 gv[4], qv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) gv[dt] = dib_attn_mc(Gs, 0, 32 * dt + l31, h);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) qv[dt] = dib_attn_mc(Qs, 0, 32 * dt + l31, h);
    __builtin_amdgcn_sched_barrier(0);
    // P and dS of query group g (register r = 4g + t <-> query qt*32 + t + 8g + 4h; lane <-> key).  Group 0 is computed here,
    // group g + 1 inside the dV step of group g - VALU in the shadow of 16 MFMAs instead of 64 exposed exponentials per tile.
    // The accumulators are made opaque at the top of every step (or the exponentials would be hoisted back up here) and the
    // results pinned at its end (or they would sink to their first use).
    float pv[16], dsv[16];
    auto softmax_group = [&](int g) {
      const float lv[4] = {lq[g].x, lq[g].y, lq[g].z, lq[g].w};
      const float dl[4] = {dq4[g].x, dq4[g].y, dq4[g].z, dq4[g].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 4 * g + t;
        const float p = __expf(sv[r] - lv[t]) * kmul;
        pv[r] = p;                                       // P
        dsv[r] = p * (dp[r] - dl[t]);                    // dS
      }
    };
    softmax_group(0);
    DIB_T(2);   // exponentials of the first query group
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 gvn[4], qvn[4];
      if (q < 3) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) gvn[dt] = dib_attn_mc(Gs, q + 1, 32 * dt + l31, h);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) DIB_PIN_ACC_V(sv[r]);
      DIB_PIN_ACC_A(dp);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dv[dt] = DIB_MFMA(gv[dt].x, pv[4 * q + 0], dv[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dv[dt] = DIB_MFMA(gv[dt].y, pv[4 * q + 1], dv[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dv[dt] = DIB_MFMA(gv[dt].z, pv[4 * q + 2], dv[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dv[dt] = DIB_MFMA(gv[dt].w, pv[4 * q + 3], dv[dt]);
      if (q < 3) softmax_group(q + 1);
      // dS^T of this group into the wave's patch: patch[query][key]
#pragma unroll
      for (int t = 0; t < 4; ++t) my_patch[(t + 8 * q + 4 * h) * 36 + l31] = dsv[4 * q + t];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) DIB_PIN_ACC_A(dv[dt]);
      if (q < 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { DIB_PIN_ACC_V(pv[4 * q + 4 + t]); DIB_PIN_ACC_V(dsv[4 * q + 4 + t]); }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q < 3) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) qvn[dt] = dib_attn_mc(Qs, q + 1, 32 * dt + l31, h);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dk[dt] = DIB_MFMA(qv[dt].x, dsv[4 * q + 0], dk[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dk[dt] = DIB_MFMA(qv[dt].y, dsv[4 * q + 1], dk[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dk[dt] = DIB_MFMA(qv[dt].z, dsv[4 * q + 2], dk[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dk[dt] = DIB_MFMA(qv[dt].w, dsv[4 * q + 3], dk[dt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) DIB_PIN_ACC_A(dk[dt]);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 3) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { gv[dt] = gvn[dt]; qv[dt] = qvn[dt]; }
      }
    }
    DIB_T(3);   // dV / dK products issued
    // first K fragments of the dQ product: the K block is static, so they can be in flight across the barrier
    const float4 ka0 = dib_attn_mc(Kblk, 0, 32 * wave + l31, h), kb0 = dib_attn_mc(Kblk, 1, 32 * wave + l31, h);
    __syncthreads();   // all four dS^T patches are in LDS; nobody reads Qs / Gs / Ls / Ds any more
    DIB_T(4);   // barrier B
    {
      // next query tile (the last iteration re-loads its own tile: no branch in the loop body).  Issuing these loads at the
      // TOP of the tile and storing them to LDS here, in front of the dQ product (32 registers live through the dP / dV / dK
      // phases), measured slower: 7.75 -> 8.12 ms in stash mode (profiles/r03f_attention_early_prefetch_ab.txt)
      const int qnext = min(qt + 1, n_tiles - 1) * kAttnTile;
      rq = dib_attn_gload(Qb, a.ld, qnext, P - 1, tid);
      rg = dib_attn_gload(dOb, a.ld, qnext, P - 1, tid);
      if (tid < kAttnTile) {
        rl_ = qnext + tid < P ? lse_b[qnext + tid] : INFINITY;
        rd_ = dlt_b[min(qnext + tid, P - 1)];
      }
    }
    {
      // dQ^T[d = 32*wave + .][query] over the workgroup's 128 keys: A = K block (MC), B = dS^T patches (b128 along keys).
      // Two accumulators (even / odd 8-key steps), MFMAs alternating between them: the LDS reads and waits between the
      // MFMAs then never sit between two products on the same accumulator
      dib_f32x16 dq, dq1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dq[r] = 0.f; dq1[r] = 0.f; }
      auto ds_frag = [&](int st) {   // st = 4 * key wave + g
        return *reinterpret_cast<const float4*>(patches + (st >> 2) * kAttnPatch + l31 * 36 + 8 * (st & 3) + 4 * h);
      };
      auto k_frag = [&](int st) { return dib_attn_mc(Kblk + (st >> 2) * 32 * kAttnPitch, st & 3, 32 * wave + l31, h); };
      float4 dsa = ds_frag(0), dsb = ds_frag(1), ka = ka0, kb = kb0;
#pragma unroll
      for (int st = 0; st < 16; st += 2) {
        const int sn = st < 14 ? st + 2 : 14;
        const float4 dsan = ds_frag(sn), dsbn = ds_frag(sn + 1), kan = k_frag(sn), kbn = k_frag(sn + 1);
        __builtin_amdgcn_sched_barrier(0);
        dq = DIB_MFMA(ka.x, dsa.x, dq);
        dq1 = DIB_MFMA(kb.x, dsb.x, dq1);
        dq = DIB_MFMA(ka.y, dsa.y, dq);
        dq1 = DIB_MFMA(kb.y, dsb.y, dq1);
        dq = DIB_MFMA(ka.z, dsa.z, dq);
        dq1 = DIB_MFMA(kb.z, dsb.z, dq1);
        dq = DIB_MFMA(ka.w, dsa.w, dq);
        dq1 = DIB_MFMA(kb.w, dsb.w, dq1);
        DIB_PIN_ACC_A(dq);
        DIB_PIN_ACC_A(dq1);
        __builtin_amdgcn_sched_barrier(0);
        dsa = dsan; dsb = dsbn; ka = kan; kb = kbn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] += dq1[r];
      // dq[r] = dQ^T[d = 32*wave + (r&3) + 8(r>>2) + 4h][query l31]
      // non-temporal: the 3.4 GB of partials are read once, by the reduce kernel, after the whole launch (same-box A/B:
      // backward 7.73 -> 7.66 ms, profiles/r03u_attention_dq_partial_nt_ab.txt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(my_dqs + l31 * 36 + 8 * g + 4 * h) =
            make_float4(dq[4 * g] * dq_mul, dq[4 * g + 1] * dq_mul, dq[4 * g + 2] * dq_mul, dq[4 * g + 3] * dq_mul);
      __builtin_amdgcn_wave_barrier();   // wave-private patch: LDS operations of one wave complete in order
      // store instruction j: query rows 8j .. 8j + 7, eight lanes x 16 bytes per row
      float4 rv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const float4*>(my_dqs + ((lane >> 3) + 8 * j) * 36 + (lane & 7) * 4);
      float* dst = dq_out + (long long)(qt * kAttnTile + (lane >> 3)) * dq_ld + 32 * wave + (lane & 7) * 4;
      if ((qt + 1) * kAttnTile <= P) {   // workgroup-uniform: every tile but a ragged last one
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_nontemporal_store(dib_nt4a{rv[j].x, rv[j].y, rv[j].z, rv[j].w}, reinterpret_cast<dib_nt4a*>(dst + 8 * j * dq_ld));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (qt * kAttnTile + (lane >> 3) + 8 * j < P)
            __builtin_nontemporal_store(dib_nt4a{rv[j].x, rv[j].y, rv[j].z, rv[j].w}, reinterpret_cast<dib_nt4a*>(dst + 8 * j * dq_ld));
      }
      __builtin_amdgcn_wave_barrier();
    }
    DIB_T(5);   // next-tile loads issued, dQ product, dQ store
    DIB_ATTN_STAGE_TILE();
    DIB_T(6);   // next tile -> LDS
  }
#ifdef DIB_ATTN_TIMING
  if (blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
    for (int i = 0; i < 7; ++i) dib_attn_dbg[i] = tacc_[i];
    dib_attn_dbg[7] = clock64() - tstart_;
    dib_attn_dbg[8] = n_tiles;
    dib_attn_dbg[9] = wall_clock64() - wstart_;
  }
#endif
#undef DIB_ATTN_STAGE_TILE
  dib_attn_store_rows(a.dv + tok0 * a.ld + head * kAttnD, a.ld, krow, k_ok && wave_ok, h, dv, 1.0f);
  dib_attn_store_rows(a.dk + tok0 * a.ld + head * kAttnD, a.ld, krow, k_ok && wave_ok, h, dk, 1.0f);  // Q tile was pre-scaled
}

// dq[token][head cols] = scale * sum_{key blocks, fixed order} part[b][h][kb][query][128]
__global__ void __launch_bounds__(256)
dib_attn_dq_reduce_kernel(const float* __restrict__ part, int B, int P, int H, int n_key_blocks, long long ld, float scale,
                          float* __restrict__ dq) {
  const long long total4 = (long long)B * H * P * (kAttnD / 4);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % (kAttnD / 4));
    const long long row = i / (kAttnD / 4);            // (b*H + head)*P + query
    const int qy = (int)(row % P);
    const long long bh = row / P;
    const float4* src = reinterpret_cast<const float4*>(part + ((bh * n_key_blocks) * P + qy) * kAttnD) + c4;
    // (a version with eight non-temporal partial loads in flight per thread measured SLOWER: 726 vs 601 us at 4 x 4096 x 12,
    // profiles/r03h_config5_kernel_stats.csv vs r03b - the plain loop already streams at 5.5 TB/s)
    float4 sacc = src[0];
    for (int kb = 1; kb < n_key_blocks; ++kb) {
      const float4 v = src[(long long)kb * P * (kAttnD / 4)];
      sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
    }
    const long long b = bh / H;
    const int head = (int)(bh % H);
    *reinterpret_cast<float4*>(dq + (b * P + qy) * ld + head * kAttnD + 4 * c4) =
        make_float4(sacc.x * scale, sacc.y * scale, sacc.z * scale, sacc.w * scale);
  }
}
