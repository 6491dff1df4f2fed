// icache_probe: what does a kernel that runs ONCE pay for the instructions it executes?  One workgroup of 256 threads runs
// the same 16 384 independent v_fma_f32 (8 bytes each) either as straight-line code (128 KB fetched once) or as a loop over a
// smaller body (the body stays in the instruction cache after its first trip).  Between timed launches a different 128 KB
// kernel runs on every CU (evicts the instruction caches), as the other kernels of a training step do.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o exp/icache_probe  &&  exp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define F8 asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
                        "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n" \
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z));
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
#define F2048 F512 F512 F512 F512

template <int BODY /* units of 512 instructions */, int TRIPS>
__global__ void __launch_bounds__(256) probe(float* out, float y, float z) {
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
    if constexpr (BODY >= 1) { F512 }
    if constexpr (BODY >= 2) { F512 }
    if constexpr (BODY >= 4) { F512 F512 }
    if constexpr (BODY >= 8) { F2048 }
    if constexpr (BODY >= 16) { F2048 F2048 }
    if constexpr (BODY >= 32) { F2048 F2048 F2048 F2048 }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// the evictor: another 128 KB of straight-line code on every CU
__global__ void __launch_bounds__(256) evict(float* out, float y, float z) {
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
  F2048 F2048 F2048 F2048 F2048 F2048 F2048 F2048
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <typename K>
static float timed(K k, int grid, float* buf, bool cold) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ms;
  for (int i = 0; i < 15; ++i) {
    if (cold) hipLaunchKernelGGL(evict, dim3(1024), dim3(256), 0, 0, buf + (1 << 20), 1.0001f, 0.5f);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, 1.0001f, 0.5f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b);
    ms.push_back(t * 1e3f);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  float* buf; hipMalloc(&buf, (2 << 20) * sizeof(float));
  printf("16384 v_fma_f32 per thread, 256 threads per workgroup; median us of 15 launches (hipEvent pair around ONE launch)\n");
  printf("%-44s %10s %10s %10s %10s\n", "code shape", "1wg cold", "1wg warm", "256wg cold", "256wg warm");
#define ROW(BODY, TRIPS, NAME)                                                                      \
  printf("%-44s %10.1f %10.1f %10.1f %10.1f\n", NAME, timed(probe<BODY, TRIPS>, 1, buf, true),       \
         timed(probe<BODY, TRIPS>, 1, buf, false), timed(probe<BODY, TRIPS>, 256, buf, true),       \
         timed(probe<BODY, TRIPS>, 256, buf, false));
  ROW(32, 1, "straight line: 128 KB x 1 trip")
  ROW(16, 2, "64 KB body x 2 trips")
  ROW(8, 4, "32 KB body x 4 trips")
  ROW(4, 8, "16 KB body x 8 trips")
  ROW(2, 16, "8 KB body x 16 trips")
  ROW(1, 32, "4 KB body x 32 trips")
  // the cost of the launch itself: an (almost) empty body
  printf("%-44s %10.1f %10.1f\n", "4 KB body x 1 trip (launch floor + 512 fma)", timed(probe<1, 1>, 1, buf, true),
         timed(probe<1, 1>, 1, buf, false));
  printf("%-44s %10.1f %10.1f\n", "16 KB body x 1 trip", timed(probe<4, 1>, 1, buf, true), timed(probe<4, 1>, 1, buf, false));
  printf("%-44s %10.1f %10.1f\n", "32 KB body x 1 trip", timed(probe<8, 1>, 1, buf, true), timed(probe<8, 1>, 1, buf, false));
  printf("%-44s %10.1f %10.1f\n", "64 KB body x 1 trip", timed(probe<16, 1>, 1, buf, true), timed(probe<16, 1>, 1, buf, false));
  return 0;
}
