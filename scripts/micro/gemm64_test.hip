// Stand-alone check + timing of brl_amd/csrc/mlp_gemm.hpp (the fp32 MFMA GEMMs of the PPO minibatch step) beside rocBLAS on the
// same shapes in the same process (interleaved rounds: cdna_hip_programming.md §5.4 rule 24).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I brl_amd/csrc -o scripts/micro/gemm64_test scripts/micro/gemm64_test.hip -lrocblas
//   ./scripts/micro/gemm64_test [rounds]     # every layout / shape vs a float64 CPU reference, then timings
//   add -DMG_TIMING for in-kernel stamps (prologue / K loop / epilogue cycles, shader clock)
#include <hip/hip_runtime.h>
#include <math.h>
#include <rocblas/rocblas.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../../brl_amd/csrc/mlp_gemm.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define RB(x) do { rocblas_status e = (x); if (e != rocblas_status_success) { printf("%s: rocblas status %d\n", #x, (int)e); exit(1); } } while (0)

static uint32_t rs = 12345u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

struct Case { const char *name; bool akc, bkc; int M, N, K; int epi; int act; };

static int g_nb = 2;   // 16-column B blocks per wave: 2 = 64 x 64 tiles, 1 = 64 x 32 tiles (argv[2])
static void launch(const Case &c, const mg::Args &G, hipStream_t s) {
  const int tiles = ((G.M + 63) / 64) * ((G.N + 32 * g_nb - 1) / (32 * g_nb));
#define L(a, b, e) do { if (g_nb == 2) hipLaunchKernelGGL((mg::k_gemm64n<a, b, e, 2>), dim3(tiles), dim3(mg::THREADS), 0, s, G); \
                        else hipLaunchKernelGGL((mg::k_gemm64n<a, b, e, 1>), dim3(tiles), dim3(mg::THREADS), 0, s, G); } while (0)
  if (c.akc && c.bkc) { if (c.epi == mg::EPI_BIAS_ACT) L(true, true, mg::EPI_BIAS_ACT); else L(true, true, mg::EPI_NONE); }
  else if (c.akc && !c.bkc) { if (c.epi == mg::EPI_GATE_COLSUM) L(true, false, mg::EPI_GATE_COLSUM); else L(true, false, mg::EPI_NONE); }
  else if (!c.akc && !c.bkc) { if (c.epi == mg::EPI_SQSUM) L(false, false, mg::EPI_SQSUM); else L(false, false, mg::EPI_NONE); }
  else { printf("layout not instantiated\n"); exit(1); }
#undef L
}

// the same product by rocBLAS (row-major C[M][N] = column-major C^T [N x M])
static void launch_lib(rocblas_handle h, const Case &c, const float *A, const float *B, float *C) {
  const float one = 1.0f, zero = 0.0f;
  const int M = c.M, N = c.N, K = c.K;
  if (c.akc && c.bkc) RB(rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, B, K, A, K, &zero, C, N));
  else if (c.akc && !c.bkc) RB(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, N, M, K, &one, B, N, A, K, &zero, C, N));
  else RB(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, N, M, K, &one, B, N, A, M, &zero, C, N));
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  g_nb = argc > 2 ? atoi(argv[2]) : 2;
  printf("tiles 64 x %d\n", 32 * g_nb);
  const Case cases[] = {
      {"fwd  NT 1024x1024x1024 +bias relu", true, true, 1024, 1024, 1024, mg::EPI_BIAS_ACT, 0},
      {"fwd0 NT 1024x1024x480  +bias relu", true, true, 1024, 1024, 480, mg::EPI_BIAS_ACT, 0},
      {"dh   NN 1024x1024x1024 gate colsum", true, false, 1024, 1024, 1024, mg::EPI_GATE_COLSUM, 0},
      {"dW   TN 1024x1024x1024 sqsum", false, false, 1024, 1024, 1024, mg::EPI_SQSUM, 0},
      {"dW0  TN 1024x480x1024 sqsum (edge)", false, false, 1024, 480, 1024, mg::EPI_SQSUM, 0},
      {"plain NT 1024^3", true, true, 1024, 1024, 1024, mg::EPI_NONE, 0},
      {"plain NN 1024^3", true, false, 1024, 1024, 1024, mg::EPI_NONE, 0},
      {"plain NT 2048x1024x1024 (2 wg/CU)", true, true, 2048, 1024, 1024, mg::EPI_NONE, 0},
      {"plain NT 3072x1024x1024 (3 wg/CU)", true, true, 3072, 1024, 1024, mg::EPI_NONE, 0},
      {"odd  NT 200x72x64 bias tanh", true, true, 200, 72, 64, mg::EPI_BIAS_ACT, 1},
      {"odd  NN 100x36x96 gate tanh colsum", true, false, 100, 36, 96, mg::EPI_GATE_COLSUM, 1},
      {"odd  TN 68x132x1000 K tail sqsum", false, false, 68, 132, 1000, mg::EPI_SQSUM, 0},
      {"odd  NN 1000x256x256 gate colsum", true, false, 1000, 256, 256, mg::EPI_GATE_COLSUM, 0},
      {"odd  NT 48x256x480 bias relu", true, true, 48, 256, 480, mg::EPI_BIAS_ACT, 0},
  };
  hipStream_t s;
  CK(hipStreamCreate(&s));
  rocblas_handle rh;
  RB(rocblas_create_handle(&rh));
  RB(rocblas_set_stream(rh, s));
  int bad = 0;
  for (const Case &c : cases) {
    const int M = c.M, N = c.N, K = c.K;
    std::vector<float> A((size_t)M * K), B((size_t)N * K), bias(N), gate((size_t)M * N);
    for (auto &x : A) x = frand();
    for (auto &x : B) x = frand();
    for (auto &x : bias) x = frand();
    for (auto &x : gate) x = frand();
    const int64_t lda = c.akc ? K : M, ldb = c.bkc ? K : N;
    auto a_at = [&](int m, int k) { return c.akc ? A[(size_t)m * K + k] : A[(size_t)k * M + m]; };
    auto b_at = [&](int n, int k) { return c.bkc ? B[(size_t)n * K + k] : B[(size_t)k * N + n]; };
    float *dA, *dB, *dC, *dC2, *dbias, *dgate, *dcs, *dsq;
    const int csr = (M + 63) / 64, ntile = csr * ((N + 32 * g_nb - 1) / (32 * g_nb));
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dgate, (size_t)M * N * 4)); CK(hipMalloc(&dcs, (size_t)csr * N * 4)); CK(hipMalloc(&dsq, ntile * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgate, gate.data(), (size_t)M * N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4)); CK(hipMemset(dcs, 0xFF, (size_t)csr * N * 4)); CK(hipMemset(dsq, 0xFF, ntile * 4));
    mg::Args G{};
    G.A = dA; G.lda = lda; G.B = dB; G.ldb = ldb; G.C = dC; G.ldc = N; G.M = M; G.N = N; G.K = K; G.act = c.act;
    G.bias = dbias; G.gate = dgate; G.ldg = N; G.colsum = dcs; G.sqsum = dsq;
#ifdef MG_TIMING
    unsigned long long *ddbg;
    CK(hipMalloc(&ddbg, (size_t)ntile * 64));
    CK(hipMemset(ddbg, 0, (size_t)ntile * 64));
    G.dbg = ddbg;
#endif
    launch(c, G, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> C((size_t)M * N), cs((size_t)csr * N), sq(ntile);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cs.data(), dcs, cs.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sq.data(), dsq, sq.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    const int rstep = (M > 256) ? 37 : 1;
    for (int m = 0; m < M; m += rstep)
      for (int n = 0; n < N; n++) {
        double acc = 0;
        for (int k = 0; k < K; k++) acc += (double)a_at(m, k) * (double)b_at(n, k);
        if (c.epi == mg::EPI_BIAS_ACT) { acc += bias[n]; acc = c.act == 0 ? (acc > 0 ? acc : 0) : tanh(acc); }
        if (c.epi == mg::EPI_GATE_COLSUM) {
          const double hh = gate[(size_t)m * N + n];
          acc = c.act == 0 ? (hh > 0 ? acc : 0) : acc * (1.0 - hh * hh);
        }
        maxerr = fmax(maxerr, fabs(acc - (double)C[(size_t)m * N + n]));
        maxref = fmax(maxref, fabs(acc));
      }
    double cserr = 0, sqerr = 0;
    if (c.epi == mg::EPI_GATE_COLSUM)
      for (int b = 0; b < csr; b++)
        for (int n = 0; n < N; n++) {
          double t = 0;
          for (int m = 64 * b; m < 64 * b + 64 && m < M; m++) t += C[(size_t)m * N + n];
          cserr = fmax(cserr, fabs(t - cs[(size_t)b * N + n]));
        }
    if (c.epi == mg::EPI_SQSUM) {
      double t = 0, u = 0;
      for (size_t i = 0; i < C.size(); i++) t += (double)C[i] * C[i];
      for (int i = 0; i < ntile; i++) u += sq[i];
      sqerr = fabs(t - u) / fmax(1.0, t);
    }
    const bool ok = maxerr < 2e-4 * fmax(1.0, maxref) * sqrt((double)K / 1024.0 + 1.0) && cserr < 2e-3 && sqerr < 1e-5;
    bad += !ok;
    // timing: interleaved rounds of 200 launches between one event pair each, own kernel and rocBLAS
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> mine, lib;
    for (int i = 0; i < 20; i++) { launch(c, G, s); launch_lib(rh, c, dA, dB, dC2); }
    for (int r = 0; r < rounds; r++) {
      float ms;
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 200; i++) launch(c, G, s);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      mine.push_back(ms * 1e3 / 200);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 200; i++) launch_lib(rh, c, dA, dB, dC2);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      lib.push_back(ms * 1e3 / 200);
    }
    std::sort(mine.begin(), mine.end()); std::sort(lib.begin(), lib.end());
    const double us = mine[mine.size() / 2], ul = lib[lib.size() / 2];
    printf("%-38s %s max|err| %.2e (max|ref| %.1f) colsum %.1e sq %.1e | own %6.2f us (min %6.2f) %6.1f TF | rocBLAS %6.2f us (min %6.2f) %6.1f TF\n",
           c.name, ok ? "ok  " : "FAIL", maxerr, maxref, cserr, sqerr, us, mine[0], 2.0 * M * N * K / (us * 1e-6) / 1e12, ul, lib[0],
           2.0 * M * N * K / (ul * 1e-6) / 1e12);
#ifdef MG_TIMING
    {
      std::vector<unsigned long long> d((size_t)ntile * 8);
      CK(hipMemcpy(d.data(), ddbg, d.size() * 8, hipMemcpyDeviceToHost));
      double pro = 0, loop = 0, epi = 0, rt = 0; unsigned long long rmin = ~0ull, rmax = 0;
      for (int t = 0; t < ntile; t++) {
        pro += d[t * 8 + 1] - d[t * 8]; loop += d[t * 8 + 2] - d[t * 8 + 1]; epi += d[t * 8 + 3] - d[t * 8 + 2];
        rt += d[t * 8 + 6] - d[t * 8 + 5];
        if (d[t * 8 + 4] < rmin) rmin = d[t * 8 + 4];
        if (d[t * 8 + 7] > rmax) rmax = d[t * 8 + 7];
      }
      printf("    wg mean: prologue %.0f  loop %.0f cycles (%.0f per chunk)  epilogue %.0f;  loop %.2f us => clock %.2f GHz;  first stamp -> last stamp %.2f us\n",
             pro / ntile, loop / ntile, loop / ntile / ((K + 31) / 32), epi / ntile, rt / ntile / 100.0, (loop / ntile) / (rt / ntile * 10.0), (rmax - rmin) / 100.0);
      CK(hipFree(ddbg));
    }
#endif
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dC2)); CK(hipFree(dbias)); CK(hipFree(dgate)); CK(hipFree(dcs)); CK(hipFree(dsq));
  }
  printf(bad ? "FAILED: %d case(s)\n" : "all ok\n", bad);
  return bad ? 1 : 0;
}
