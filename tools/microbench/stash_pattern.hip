// stash_pattern.hip - HBM write throughput of the fused kernels' stash pattern vs fully contiguous wave stores.
// A wave owns 16 KB chunks (32 rows x 512 B of one feature's [B][128] stash).  Pattern 0 ("row segments", what dib_store_tile
// does): 16 store instructions per chunk, each covering 8 rows x 128 B (lanes 8r..8r+7 write one 128-byte segment of row r;
// rows of one instruction are 512 B x {1, 4, 16, 20} apart).  Pattern 1 ("native"): 16 instructions x 1 KiB contiguous.
// 256 workgroups x 8 waves stream `gb` GB; reports event time and TB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stash_pattern.hip -o exp/stash_pattern && exp/stash_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float nt4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ void __launch_bounds__(512) stash(float* dst, int chunks_per_wave, int work) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v = (float)threadIdx.x;
  for (int c = 0; c < chunks_per_wave; ++c) {
    // chunk order as in the kernel: workgroup-major tiles, 8 waves = 8 consecutive chunks of a 256-row tile
    float* chunk = dst + ((size_t)(c * gridDim.x + blockIdx.x) * 8 + wave) * 4096;
    for (int k = 0; k < work; ++k) v = fmaf(v, 1.0000001f, 0.5f);
    nt4 x = {v, v, v, v};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float* p;
      if (PATTERN == 0) {
        const int jo = i >> 2, pss = i & 3;                       // unit tile, row pass
        const int row = (lane >> 3) + 8 * 0 + 4 * (pss & 1) + 16 * (pss >> 1) + ((lane >> 3) >= 4 ? 4 : 0);  // 8 rows per pass
        p = chunk + row * 128 + jo * 32 + (lane & 7) * 4;
      } else {
        p = chunk + i * 256 + lane * 4;
      }
      __builtin_nontemporal_store(x, reinterpret_cast<nt4*>(p));
    }
  }
}

int main() {
  const size_t bytes = 4ull << 30;
  float* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int cpw = (int)(bytes / 16384 / (256 * 8));
  printf("%8s %5s | %10s %8s\n", "pattern", "work", "event us", "TB/s");
  for (int work : {0, 2000, 4000}) for (int pat = 0; pat < 2; ++pat) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      if (pat) hipLaunchKernelGGL(stash<1>, dim3(256), dim3(512), 0, 0, buf, cpw, work);
      else hipLaunchKernelGGL(stash<0>, dim3(256), dim3(512), 0, 0, buf, cpw, work);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%8s %5d | %10.1f %8.2f\n", pat ? "native" : "rowseg", work, best * 1e3, (double)cpw * 256 * 8 * 16384 / (best * 1e-3) / 1e12);
  }
  return 0;
}
