/* c_abi_step.c - one Distributed-IB training step driven through the C ABI alone (no Python, no torch):
 * what a C / Go (cgo) / Rust (FFI) host does with libdib_hip.so.  Deterministic inputs, prints the per-feature KL
 * (nats), the task loss and a gradient checksum; tests/test_gpu_parity.py runs it and compares with the Python engine.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ examples/c_abi_step.c -Iinclude -I/opt/rocm/include -L<pkg> -L/opt/rocm/lib \
 *       -ldib_hip -lamdhip64 -lm -Wl,-rpath,<pkg> -Wl,-rpath,/opt/rocm/lib -o examples/c_abi_step
 *   (the HIP runtime is used only for hipMalloc / hipMemcpy: the host owns every buffer, as the ABI prescribes)
 *   examples/c_abi_step [batch]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dib_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_DIB(x) do { int e_ = (x); if (e_ != DIB_OK) { fprintf(stderr, "%s: %s\n", #x, dib_error_string(e_)); return 3; } } while (0)

/* the same closed-form test pattern the Python side generates (tests/test_gpu_parity.py) */
static float pattern(long long i, float scale) { return scale * sinf(0.37f * (float)(i % 1009) + 0.001f * (float)(i % 7919)); }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 300;
  enum { F = 4, E = 32, OUT = 1 };
  const int dims[F] = {1, 1, 1, 1}, enc[2] = {32, 32}, integ[1] = {64};
  dib_layout* l = NULL;
  CHECK_DIB(dib_layout_create(F, dims, 2, enc, E, 1, integ, OUT, /*posenc*/ 1, /*n_freq*/ 5, DIB_ACT_RELU, DIB_ACT_LINEAR, &l));
  const int64_t n = dib_layout_param_count(l);

  void *tables, *ws;
  float *params, *grads, *x, *y, *beta;
  CHECK_HIP(hipMalloc(&tables, (size_t)dib_layout_table_bytes(l)));
  CHECK_DIB(dib_layout_upload_tables(l, tables, NULL));
  const int64_t ws_bytes = dib_workspace_bytes(l, B);
  CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
  CHECK_DIB(dib_workspace_init(l, B, ws, 0)); /* include/dib_hip.h contract: once per (workspace, batch size) */
  CHECK_HIP(hipMalloc((void**)&params, (size_t)n * 4));
  CHECK_HIP(hipMalloc((void**)&grads, (size_t)n * 4));
  CHECK_HIP(hipMemset(grads, 0, (size_t)n * 4));
  CHECK_HIP(hipMalloc((void**)&x, (size_t)B * F * 4));
  CHECK_HIP(hipMalloc((void**)&y, (size_t)B * 4));
  CHECK_HIP(hipMalloc((void**)&beta, 4));

  float* h = (float*)malloc((size_t)(n > (int64_t)B * F ? n : (int64_t)B * F) * 4);
  for (int64_t i = 0; i < n; ++i) h[i] = pattern(i, 0.2f);
  CHECK_HIP(hipMemcpy(params, h, (size_t)n * 4, hipMemcpyHostToDevice));
  for (int64_t i = 0; i < (int64_t)B * F; ++i) h[i] = pattern(i + 12345, 1.5f);
  CHECK_HIP(hipMemcpy(x, h, (size_t)B * F * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < B; ++i) h[i] = (pattern(i + 777, 1.0f) > 0.f) ? 1.f : 0.f;
  CHECK_HIP(hipMemcpy(y, h, (size_t)B * 4, hipMemcpyHostToDevice));
  const float beta_h = 0.25f;
  CHECK_HIP(hipMemcpy(beta, &beta_h, 4, hipMemcpyHostToDevice));

  const uint64_t seed = 42;
  const uint32_t step = 3;
  const float inv_b = 1.0f / (float)B;
  CHECK_DIB(dib_encoder_bank_fwd(l, x, F, NULL, 0, B, params, seed, step, 0, ws, NULL));
  CHECK_DIB(dib_integration_fwd(l, B, params, ws, NULL));
  CHECK_DIB(dib_loss_fwd_bwd(l, DIB_LOSS_BCE_LOGITS, y, 1, NULL, 0, B, inv_b, 0, ws, NULL));
  CHECK_DIB(dib_integration_bwd(l, B, params, grads, ws, NULL));
  CHECK_DIB(dib_encoder_bank_bwd(l, B, params, grads, beta, inv_b, ws, NULL));
  CHECK_DIB(dib_grads_finalize(l, B, grads, ws, NULL));
  CHECK_HIP(hipDeviceSynchronize());

  float so[F + 3];
  CHECK_HIP(hipMemcpy(so, (char*)ws + dib_workspace_offset(l, B, DIB_WS_STEP_OUT), sizeof(so), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h, grads, (size_t)n * 4, hipMemcpyDeviceToHost));
  double gsum = 0.0, gabs = 0.0;
  for (int64_t i = 0; i < n; ++i) { gsum += h[i]; gabs += fabs(h[i]); }
  printf("%s\n", dib_version());
  printf("params %lld workspace_bytes %lld\n", (long long)n, (long long)ws_bytes);
  printf("KL %.9g %.9g %.9g %.9g\n", so[0] / B, so[1] / B, so[2] / B, so[3] / B);
  printf("task_loss %.9g correct %.0f rows %.0f\n", so[F] / B, so[F + 1], so[F + 2]);
  printf("grad_sum %.9g grad_abs %.9g\n", gsum, gabs);
  free(h);
  dib_layout_destroy(l);
  hipFree(tables); hipFree(ws); hipFree(params); hipFree(grads); hipFree(x); hipFree(y); hipFree(beta);
  return 0;
}
