// lx3_exp.hip — where does brl_linear_x3p's 128 x 128 kernel spend its time?  (-DLX3_TIMING: also the shader clock it runs at — s_memtime
// against the 100 MHz s_memrealtime around the K loop — with constant and with random operands: argv[2] = 1 -> random bits in the planes)  Timing-only builds of csrc/mlp_linear_x3p.hpp with parts of
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
  const bool random_bits = argc > 2 && atoi(argv[2]) == 1;
  if (random_bits) {      // finite bf16 values with random sign / mantissa, exponents around 1 (hi), 2^-8 (mid), 2^-16 (lo): what a split leaves
    auto fill = [&](uint16_t *dst, size_t rows_k) -> int {
      uint16_t *h = (uint16_t *)malloc(rows_k * 3 * 2);
      unsigned s = 777u;
      for (int pl = 0; pl < 3; pl++)
        for (size_t i = 0; i < rows_k; i++) {
          s = s * 1664525u + 1013904223u;
          const unsigned e = 127u - 8u * pl - ((s >> 8) & 3u);
          h[pl * rows_k + i] = (uint16_t)(((s >> 31) << 15) | (e << 7) | ((s >> 16) & 0x7fu));
        }
      hipError_t e_ = hipMemcpy(dst, h, rows_k * 3 * 2, hipMemcpyHostToDevice);
      free(h);
      return e_ == hipSuccess ? 0 : 1;
    };
    if (fill(xp, (size_t)M * K) || fill(wp, (size_t)N * K)) return 1;
  }
  lx3::Args G{};
  G.x = xp; G.ldx = K; G.sx = (int64_t)M * K; G.w = wp; G.ldw = K; G.sw = (int64_t)N * K; G.bias = bias;
  G.y = nullptr; G.ldy = N; G.yp = yp; G.ldyp = N; G.syp = (int64_t)M * N; G.M = M; G.N = N; G.K = K; G.relu = 1;
#ifdef LX3_TIMING
  unsigned long long *dbg;
  CK(hipMalloc(&dbg, 512 * 4 * 8)); CK(hipMemset(dbg, 0, 512 * 4 * 8));
  G.dbg = dbg;
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(lx3::k_linear_x3p<3>, dim3(512), dim3(lx3::THREADS), 0, 0, G);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("LX3_EXP=%d  %s operands  %.2f us per launch\n", LX3_EXP, random_bits ? "random" : "constant", ms * 1e3 / iters);
  }
#ifdef LX3_TIMING
  {
    unsigned long long h[512 * 4];
    CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    double mhz = 0, loop_us = 0;
    for (int b = 0; b < 512; b++) {
      const double dc = (double)(h[4 * b + 2] - h[4 * b]), dr = (double)(h[4 * b + 3] - h[4 * b + 1]);
      mhz += dc / dr * 100.0;
      loop_us += dr / 100.0;
    }
    printf("  K loop of the last launch: %.1f us per workgroup, shader clock %.0f MHz (mean over 512 workgroups; s_memtime / s_memrealtime)\n", loop_us / 512, mhz / 512);
  }
#endif
  return 0;
}
