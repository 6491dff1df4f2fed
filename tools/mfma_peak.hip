// mfma_peak.hip - what fp32-MFMA rate does an MI355X actually sustain?  (tools only; not part of the product)
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o exp/mfma_peak && exp/mfma_peak        (on the GPU box)
//
// Variants (all: 256 threads = 4 waves per workgroup, 2 workgroups per CU -> 2 waves per SIMD, random operands):
//   chain1  one accumulator per wave: every MFMA depends on the previous one (what the fused encoder kernels do
//           inside an output tile)
//   chain4  four independent accumulators, round-robin (what the GEMM tile does)
//   + one extra instruction (group) per 16 MFMAs of chain4: a streaming 16-byte store / non-temporal store / store to one
//     line, a streaming 16-byte load, a ds_write_b128, six ds_read_b128, one VALU FMA, eight s_nop; and 8 VALU FMAs per MFMA
// Prints TFLOP/s (2*32*32*2 FLOP per MFMA) and the implied clock if the pipe were 100 % busy (64 cycles per MFMA per SIMD).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float nt4 __attribute__((ext_vector_type(4)));
enum { M_NONE, M_STORE, M_STORE_NT, M_STORE_SAME, M_LOAD, M_LDSW, M_LDSR6, M_VALU8, M_VALU1, M_SALU8, M_LDSR16_B32, M_LDSR4, M_LDSR16_B128 };
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, int MODE>
__global__ void __launch_bounds__(256, 2) mfma_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 8 + i) & 0xFFFFF];
    b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF];
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  __shared__ float lds[8192];
  if (MODE == M_LDSR6 || MODE == M_LDSW || MODE >= M_LDSR16_B32) { for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f; __syncthreads(); }
  float4 ld = make_float4(0.f, 0.f, 0.f, 0.f);
  float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3];
  float4* st = reinterpret_cast<float4*>(out) + (size_t)blockIdx.x * 256 * 64 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16 / NACC; ++k) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        acc[n] = MFMA(a[(k + n) & 3], b[(k * 3 + n) & 3], acc[n]);
        if (MODE == M_VALU8) {
          v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 0.75f);
          v0 = fmaf(v0, 0.9999f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.25f); v2 = fmaf(v2, 0.9998f, 0.125f); v3 = fmaf(v3, 1.0002f, 0.75f);
        }
      }
    }
    if (MODE == M_VALU1) { v0 = fmaf(v0, 1.0001f, 0.5f); }   // 1 VALU per 16 MFMAs (see M_VALU16 for per-MFMA)
    if (MODE == M_STORE) st[(size_t)(it & 63) * 256] = make_float4(acc[0][0], acc[0][1], v0, v1);   // streams 1 KB/wave
    if (MODE == M_STORE_NT)
      __builtin_nontemporal_store(nt4{acc[0][0], acc[0][1], v0, v1}, reinterpret_cast<nt4*>(st + (size_t)(it & 63) * 256));
    if (MODE == M_STORE_SAME) st[0] = make_float4(acc[0][0], acc[0][1], v0, v1);                   // same line: L2 only
    if (MODE == M_LOAD) {  // streaming 16-byte loads (1 KB/wave), consumed one iteration later
      const float4 q = reinterpret_cast<const float4*>(out)[(size_t)blockIdx.x * 256 * 64 + threadIdx.x + (size_t)(it & 63) * 256];
      v0 += ld.x; v1 += ld.y;
      ld = q;
    }
    if (MODE == M_LDSW) { *reinterpret_cast<float4*>(lds + threadIdx.x * 4 + (it & 7) * 1024) = make_float4(v0, v1, v2, v3); }
    if (MODE == M_LDSR6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + u * 1024 + (it & 1) * 512) & 8191));
        a[u & 3] += q.x * 1e-30f;
      }
    }
    if (MODE == M_LDSR16_B32) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u & 3] += lds[(threadIdx.x + u * 260 + (it & 1) * 128) & 8191] * 1e-30f;
    }
    if (MODE == M_LDSR4 || MODE == M_LDSR16_B128) {
#pragma unroll
      for (int u = 0; u < (MODE == M_LDSR4 ? 4 : 16); ++u) {
        const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + u * 1024 + (it & 1) * 512) & 8191));
        a[u & 3] += q.x * 1e-30f;
      }
    }
    if (MODE == M_SALU8) { asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0"); }
  }
  float s = v0 + v1 + v2 + v3 + ld.x + ld.y + a[0] + a[1] + a[2] + a[3];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 12345.678f) out[tid] = s;  // keep the results alive
}

// ---- split-bf16 emulation of an fp32 product on the bf16 matrix pipe (planning numbers for a later round) ----
// One "fp32-equivalent" 32x32x16 block = NPROD bf16 MFMAs (32x32x16) on hi / mid / lo pieces of the operands:
//   NPROD = 1 plain bf16; 3: hi*hi + hi*lo + lo*hi (~2^-16 relative); 6: three-way split, all terms >= 2^-16 (~fp32).
// SPLIT: the pieces are produced from fp32 registers with VALU ops inside the loop (what a kernel that receives fp32
// activations has to do once per element); otherwise they are loop-invariant (weights split once).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 to_bf16(const f32x8 v) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
  return r;
}
__device__ __forceinline__ f32x8 to_f32(const bf16x8 v) {
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}

template <int NPROD, bool SPLIT>
__global__ void __launch_bounds__(256, 2) bf16_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f32x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(tid * 16 + i) & 0xFFFFF];
    b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF];
  }
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  bf16x8 ah = to_bf16(a), bh = to_bf16(b);
  bf16x8 am = to_bf16(a - to_f32(ah)), bm = to_bf16(b - to_f32(bh));
  bf16x8 al = to_bf16(a - to_f32(ah) - to_f32(am)), bl = to_bf16(b - to_f32(bh) - to_f32(bm));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (SPLIT) {  // the B operand arrives as fp32 (activations): split it now; perturb so it is not hoisted
        b[n] += 1e-7f;
        bh = to_bf16(b);
        if (NPROD >= 3) bm = to_bf16(b - to_f32(bh));
        if (NPROD >= 6) bl = to_bf16(b - to_f32(bh) - to_f32(bm));
      }
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
      if (NPROD >= 3) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[n], 0, 0, 0);
      }
      if (NPROD >= 6) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
      }
    }
  }
  float s = b[0];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 12345.678f) out[tid] = s;
}

template <int NPROD, bool SPLIT>
static void run_bf16(const char* name, const float* in, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((bf16_kernel<NPROD, SPLIT>), dim3(blocks), dim3(256), 0, 0, in, out, iters / 8);
  hipDeviceSynchronize();
  float sum = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bf16_kernel<NPROD, SPLIT>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
  }
  // fp32-equivalent FLOPs: one 32x32x16 block per NPROD MFMAs
  const double blocks16 = (double)blocks * 4 * iters * 4;
  printf("%-22s %8.3f ms  fp32-equivalent %7.1f TFLOP/s  (matrix-pipe %7.1f TFLOP/s)\n", name, sum / 5,
         blocks16 * 32768.0 / (sum / 5 * 1e-3) / 1e12, blocks16 * NPROD * 32768.0 / (sum / 5 * 1e-3) / 1e12);
}

template <int NACC, int MODE>
static void run(const char* name, const float* in, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_kernel<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, iters / 8);  // warm-up
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_kernel<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double mfmas = (double)blocks * 4 /*waves*/ * iters * 16;
  const double tf_best = mfmas * 4096.0 / (best * 1e-3) / 1e12, tf_avg = mfmas * 4096.0 / (sum / 5 * 1e-3) / 1e12;
  // 1024 SIMDs, 64 cycles per MFMA: clock needed if the matrix pipes were 100 % busy
  printf("%-12s %8.3f ms  best %6.1f TFLOP/s  avg %6.1f TFLOP/s  (= 100%% busy at %.2f GHz)\n", name, best, tf_best, tf_avg,
         mfmas * 64.0 / 1024.0 / (sum / 5 * 1e-3) / 1e9);
}

int main() {
  const int blocks = 512 * 4, iters = 6000;  // 8192 waves = 8 per SIMD in 4 rounds of 2 resident
  std::vector<float> h(1 << 20);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
  }
  float *in, *out;
  hipMalloc(&in, h.size() * 4);
  hipMalloc(&out, (size_t)blocks * 256 * 64 * 16);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<1, M_NONE>("chain1", in, out, blocks, iters);
  run<4, M_NONE>("chain4", in, out, blocks, iters);
  run<4, M_STORE>("+store16B", in, out, blocks, iters);
  run<4, M_STORE_NT>("+store nt", in, out, blocks, iters);
  run<4, M_STORE_SAME>("+store same", in, out, blocks, iters);
  run<4, M_LOAD>("+load16B", in, out, blocks, iters);
  run<4, M_LDSW>("+ds_write", in, out, blocks, iters);
  run<4, M_LDSR6>("+6 ds_read", in, out, blocks, iters);
  run<4, M_LDSR4>("+4 ds_r128", in, out, blocks, iters);
  run<4, M_LDSR16_B32>("+16 ds_r32", in, out, blocks, iters);
  run<4, M_LDSR16_B128>("+16 ds_r128", in, out, blocks, iters);
  run<4, M_VALU1>("+1 valu/16", in, out, blocks, iters);
  run<4, M_VALU8>("+8 valu/1", in, out, blocks, iters);
  run<1, M_VALU8>("c1 +8valu/1", in, out, blocks, iters);
  run<4, M_SALU8>("+8 s_nop/16", in, out, blocks, iters);
  run<4, M_NONE>("chain4 again", in, out, blocks, iters);
  run_bf16<1, false>("bf16 x1", in, out, blocks, iters * 2);
  run_bf16<3, false>("bf16 x3 (presplit)", in, out, blocks, iters);
  run_bf16<3, true>("bf16 x3 (+split B)", in, out, blocks, iters);
  run_bf16<6, false>("bf16 x6 (presplit)", in, out, blocks, iters);
  run_bf16<6, true>("bf16 x6 (+split B)", in, out, blocks, iters);
  hipFree(in);
  hipFree(out);
  return 0;
}
