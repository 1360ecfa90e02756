// x3p_prod_test.hip — harness of scripts/micro/mlp_gemm_x3p.hpp (the planes-in bf16x3 product with its epilogues; not adopted): both layouts with their epilogues
// against float64 on the host (256 sampled rows), the planes of the output against the stored fp32 values, timing.
//   hipcc --offload-arch=gfx950 -O3 -I brl_amd/csrc -o scripts/micro/x3p_prod_test scripts/micro/x3p_prod_test.hip && scripts/micro/x3p_prod_test [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_gemm_x3p.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int M = 1024, N = 1024, K = 1024;
  std::vector<float> A((size_t)M * K), B((size_t)N * K), bias(N), gate((size_t)M * N), C((size_t)M * N), cs((size_t)(M / 64) * N);
  std::vector<uint16_t> CP((size_t)3 * M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd() * 0.05f;
  for (auto &v : bias) v = rnd();
  for (auto &v : gate) v = rnd();
  float *dA, *dB, *dC, *dbias, *dgate, *dcs;
  uint16_t *pa, *pb, *pc;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
  CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dgate, gate.size() * 4)); CK(hipMalloc(&dcs, cs.size() * 4));
  CK(hipMalloc(&pa, A.size() * 6)); CK(hipMalloc(&pb, B.size() * 6)); CK(hipMalloc(&pc, C.size() * 6));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dgate, gate.data(), gate.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(x3p::k_split_planes, dim3((unsigned)(A.size() / 4 + 255) / 256), dim3(256), 0, 0, dA, pa, (int64_t)A.size(), (int64_t)A.size() / 4);
  hipLaunchKernelGGL(x3p::k_split_planes, dim3((unsigned)(B.size() / 4 + 255) / 256), dim3(256), 0, 0, dB, pb, (int64_t)B.size(), (int64_t)B.size() / 4);
  int bad = 0;
  for (int mode = 0; mode < 2; mode++) {     // 0: NT + bias + ReLU; 1: NN + gate (tanh') + column sums
    x3p::Args G{};
    G.a = pa; G.lda = K; G.sa = (int64_t)A.size();
    G.b = pb; G.ldb = mode == 0 ? K : N; G.sb = (int64_t)B.size();      // mode 1 reads the same memory as [K][N]
    G.c = dC; G.ldc = N; G.cp = pc; G.ldcp = N; G.scp = (int64_t)C.size();
    G.M = M; G.N = N; G.K = K; G.act = mode; G.bias = dbias; G.gate = dgate; G.ldg = N; G.colsum = dcs;
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL((x3p::k_gemm_x3p<true, mg::EPI_BIAS_ACT>), dim3(256), dim3(x3p::THREADS), 0, 0, G);
      else hipLaunchKernelGGL((x3p::k_gemm_x3p<false, mg::EPI_GATE_COLSUM>), dim3(256), dim3(x3p::THREADS), 0, 0, G);
    };
    CK(hipMemset(dC, 0xff, C.size() * 4));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(CP.data(), pc, CP.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cs.data(), dcs, cs.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0, pmax = 0, csmax = 0;
    for (int r = 0; r < 256; r++) {
      const int m = r * 4 + (r & 3);
      for (int n = 0; n < N; n++) {
        double acc = 0;
        for (int k = 0; k < K; k++) acc += (double)A[(size_t)m * K + k] * (double)(mode == 0 ? B[(size_t)n * K + k] : B[(size_t)k * N + n]);
        if (mode == 0) { acc += bias[n]; acc = acc > 0 ? acc : 0; }
        else { const double g = gate[(size_t)m * N + n]; acc *= (1.0 - (double)(float)(g * g)); }
        const double e = fabs((double)C[(size_t)m * N + n] - acc);
        if (e > emax) emax = e;
        if (fabs(acc) > rmax) rmax = fabs(acc);
      }
    }
    for (size_t idx = 0; idx < C.size(); idx++) {       // planes: hi + mid + lo == the stored value, exactly
      const float sum = (bf(CP[idx]) + bf(CP[C.size() + idx])) + bf(CP[2 * C.size() + idx]);
      const double e = fabs((double)sum - (double)C[idx]);
      if (e > pmax) pmax = e;
    }
    if (mode == 1)
      for (int t = 0; t < M / 64; t++)
        for (int n = 0; n < N; n++) {
          double want = 0;
          for (int r = 0; r < 64; r++) want += C[(size_t)(64 * t + r) * N + n];
          const double e = fabs(want - cs[(size_t)t * N + n]);
          if (e > csmax) csmax = e;
        }
    printf("mode %d (%s): max |err| vs float64 %.3e (max |ref| %.2f); planes vs stored %.3e; column sums %.3e\n", mode,
           mode == 0 ? "NT, bias + ReLU" : "NN, gate (tanh') + column sums", emax, rmax, pmax, csmax);
    if (!(emax < 2e-4 * (rmax > 1 ? rmax : 1)) || pmax != 0.0 || csmax > 1e-3) bad = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; it++) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %.2f us per launch\n", ms * 1e3 / iters);
    }
  }
  printf(bad ? "FAIL\n" : "PASS\n");
  return bad;
}
