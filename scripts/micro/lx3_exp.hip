// lx3_exp.hip — where does brl_linear_x3p's 128 x 128 kernel spend its time?  Timing-only builds of csrc/mlp_linear_x3p.hpp with parts of
// the K loop removed (-DLX3_EXP=: 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no barrier; sums allowed), 8192 x 1024 x 1024, planes out.
//   hipcc --offload-arch=gfx950 -O3 -I brl_amd/csrc -DLX3_EXP=0 -o scripts/micro/lx3_exp_0 scripts/micro/lx3_exp.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_linear_x3p.hpp"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100;
  const int M = 8192, N = 1024, K = 1024;
  uint16_t *xp, *wp, *yp; float *bias, *y;
  CK(hipMalloc(&xp, (size_t)3 * M * K * 2)); CK(hipMalloc(&wp, (size_t)3 * N * K * 2)); CK(hipMalloc(&yp, (size_t)3 * M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&y, (size_t)M * N * 4));
  CK(hipMemset(xp, 0x3c, (size_t)3 * M * K * 2)); CK(hipMemset(wp, 0x3c, (size_t)3 * N * K * 2)); CK(hipMemset(bias, 0, N * 4));
  lx3::Args G{};
  G.x = xp; G.ldx = K; G.sx = (int64_t)M * K; G.w = wp; G.ldw = K; G.sw = (int64_t)N * K; G.bias = bias;
  G.y = nullptr; G.ldy = N; G.yp = yp; G.ldyp = N; G.syp = (int64_t)M * N; G.M = M; G.N = N; G.K = K; G.relu = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(lx3::k_linear_x3p<3>, dim3(512), dim3(lx3::THREADS), 0, 0, G);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("LX3_EXP=%d  %.2f us per launch\n", LX3_EXP, ms * 1e3 / iters);
  }
  return 0;
}
