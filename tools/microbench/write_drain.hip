// write_drain.hip - how long after a store-heavy kernel's LAST wave has ended does the NEXT kernel's first wave start?
// (MI355X: the fused encoder forward writes 6 GB of stashes per launch and the next kernel starts ~370 us after its
// last workgroup is done.)  fill<NT>: 256 persistent workgroups x 512 threads stream `bytes` to HBM in 16-byte stores
// (plain or non-temporal), optionally with `work` dependent FMAs between stores to pace the store stream; probe: records its
// entry time.  Time base: wall_clock64() = s_memrealtime, 100 MHz, device-global.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/write_drain.hip -o exp/write_drain && exp/write_drain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float nt4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long marks[4];   // [0] max end of fill, [1] max(~entry of probe), [2] max(~entry of fill)

template <int NT>
__global__ void __launch_bounds__(512) fill(float* dst, size_t floats_per_wg, int work) {
  const long long t_in = wall_clock64();
  float* p = dst + (size_t)blockIdx.x * floats_per_wg;
  float v = (float)threadIdx.x;
  for (size_t i = (size_t)threadIdx.x * 4; i < floats_per_wg; i += 512 * 4) {
    for (int k = 0; k < work; ++k) v = fmaf(v, 1.0000001f, 0.5f);
    nt4 x = {v, v, v, v};
    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<nt4*>(p + i));
    else *reinterpret_cast<nt4*>(p + i) = x;
  }
  if (threadIdx.x == 0) {
    atomicMax(&marks[0], (unsigned long long)wall_clock64());
    atomicMax(&marks[2], ~(unsigned long long)t_in);
  }
}
__global__ void probe() {
  if (threadIdx.x == 0) atomicMax(&marks[1], ~(unsigned long long)wall_clock64());
}

int main() {
  const size_t max_bytes = 6ull << 30;
  float* buf;
  if (hipMalloc(&buf, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%8s %3s %5s | %10s %10s %10s | %8s\n", "GB", "nt", "work", "waves us", "drain us", "event us", "TB/s(ev)");
  for (int work : {0, 64}) for (int nt = 0; nt < 2; ++nt) for (double gb : {0.25, 1.0, 2.0, 4.0, 6.0}) {
    const size_t bytes = (size_t)(gb * (1ull << 30));
    const size_t fpw = bytes / 4 / 256 / 2048 * 2048;
    float best_ev = 1e30f; double w = 0, d = 0;
    for (int rep = 0; rep < 3; ++rep) {
      unsigned long long z[4] = {0, 0, 0, 0};
      hipMemcpyToSymbol(HIP_SYMBOL(marks), z, sizeof(z));
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (nt) hipLaunchKernelGGL(fill<1>, dim3(256), dim3(512), 0, 0, buf, fpw, work);
      else hipLaunchKernelGGL(fill<0>, dim3(256), dim3(512), 0, 0, buf, fpw, work);
      hipEventRecord(e1);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long m[4];
      hipMemcpyFromSymbol(m, HIP_SYMBOL(marks), sizeof(m));
      const double end = (double)m[0], probe_in = (double)(~m[1]), fill_in = (double)(~m[2]);
      if (ms < best_ev) { best_ev = ms; w = (end - fill_in) / 100.0; d = (probe_in - end) / 100.0; }
    }
    printf("%8.2f %3d %5d | %10.1f %10.1f %10.1f | %8.2f\n", gb, nt, work, w, d, best_ev * 1e3, bytes / (best_ev * 1e-3) / 1e12);
  }
  return 0;
}
