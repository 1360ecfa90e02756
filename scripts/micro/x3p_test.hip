// x3p_test.hip — harness of scripts/micro/mlp_gemm_x3p_exp.hpp: split both operands of C = A B^T into bf16 planes, run the planes kernel,
// check against a float64 product on the host, time it.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/x3p_test scripts/micro/x3p_test.hip && scripts/micro/x3p_test [M N K] [iters]
#include "mlp_gemm_x3p_exp.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 1024, N = argc > 3 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int iters = argc > 4 ? atoi(argv[4]) : (argc == 2 ? atoi(argv[1]) : 200);
  if (M % 64 || N % 64 || K % 64) { printf("M, N, K multiples of 64\n"); return 1; }
  std::vector<float> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd();
  float *dA, *dB, *dC;
  uint16_t *pa, *pb;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
  CK(hipMalloc(&pa, A.size() * 6)); CK(hipMalloc(&pb, B.size() * 6));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, C.size() * 4));
  hipLaunchKernelGGL(x3p::k_split_planes, dim3((unsigned)((A.size() + 255) / 256)), dim3(256), 0, 0, dA, pa, pa + A.size(), pa + 2 * A.size(), (int64_t)A.size());
  hipLaunchKernelGGL(x3p::k_split_planes, dim3((unsigned)((B.size() + 255) / 256)), dim3(256), 0, 0, dB, pb, pb + B.size(), pb + 2 * B.size(), (int64_t)B.size());
  x3p::Args G{};
  for (int p = 0; p < 3; p++) { G.a[p] = pa + p * A.size(); G.b[p] = pb + p * B.size(); }
  G.lda = K; G.ldb = K; G.c = dC; G.ldc = N; G.M = M; G.N = N; G.K = K;
  const unsigned blocks = (unsigned)((M / 64) * (N / 64));
  hipLaunchKernelGGL(x3p::k_gemm_x3p, dim3(blocks), dim3(x3p::THREADS), 0, 0, G);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, rmax = 0;
  const int rows_checked = M < 256 ? M : 256;
  for (int r = 0; r < rows_checked; r++) {
    const int m = (int)(((long long)r * M) / rows_checked);
    for (int n = 0; n < N; n++) {
      double acc = 0;
      for (int k = 0; k < K; k++) acc += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
      const double e = fabs((double)C[(size_t)m * N + n] - acc);
      if (e > emax) emax = e;
      if (fabs(acc) > rmax) rmax = fabs(acc);
    }
  }
  printf("%d x %d x %d: max |err| vs float64 %.3e (max |ref| %.2f) over %d rows%s\n", M, N, K, emax, rmax, rows_checked, (X3P_EXP ? "  [EXPERIMENT BUILD: wrong results expected]" : ""));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(x3p::k_gemm_x3p, dim3(blocks), dim3(x3p::THREADS), 0, 0, G);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %.2f us per launch (%d launches back to back), %.1f TFLOP/s fp32-equivalent\n", ms * 1e3 / iters, iters, 2.0 * M * N * K / (ms * 1e-3 / iters) / 1e12);
  }
  return emax < 2e-4 * (rmax > 1 ? rmax : 1) || X3P_EXP ? 0 : 2;
}
