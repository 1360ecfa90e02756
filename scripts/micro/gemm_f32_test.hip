// Stand-alone check + timing of scripts/micro/mlp_gemm.hpp (the fp32 MFMA GEMMs of the PPO minibatch step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/gemm_f32_test scripts/micro/gemm_f32_test.hip
//   ./scripts/micro/gemm_f32_test            # every layout / shape of the step vs a float64 CPU reference, then timings
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "mlp_gemm.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static uint32_t rs = 12345u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

struct Case { const char *name; bool akc, bkc; int M, N, K; bool bias, relu, gate, colsum; };

static void launch(const Case &c, const mlpg::Args &G, hipStream_t s) {
  const int tiles = ((G.M + 63) / 64) * ((G.N + 63) / 64);
  if (c.akc && c.bkc) hipLaunchKernelGGL((mlpg::k_mlp_gemm<true, true>), dim3(tiles), dim3(mlpg::THREADS), 0, s, G);
  else if (!c.akc && !c.bkc) hipLaunchKernelGGL((mlpg::k_mlp_gemm<false, false>), dim3(tiles), dim3(mlpg::THREADS), 0, s, G);
  else if (c.akc && !c.bkc) hipLaunchKernelGGL((mlpg::k_mlp_gemm<true, false>), dim3(tiles), dim3(mlpg::THREADS), 0, s, G);
  else { printf("layout not instantiated\n"); exit(1); }
}

int main() {
  const Case cases[] = {
      {"fwd  NT 1024x1024x1024 +bias relu", true, true, 1024, 1024, 1024, true, true, false, false},
      {"fwd0 NT 1024x1024x480  +bias relu", true, true, 1024, 1024, 480, true, true, false, false},
      {"dW   TN 1024x1024x1024", false, false, 1024, 1024, 1024, false, false, false, false},
      {"dW0  TN 1024x480x1024 (edge tile)", false, false, 1024, 480, 1024, false, false, false, false},
      {"dh   NN 1024x1024x1024 gate colsum", true, false, 1024, 1024, 1024, false, false, true, true},
      {"odd  NT 200x72x64 edge tiles", true, true, 200, 72, 64, true, false, false, true},
      {"odd  NN 100x36x96 edge tiles gate", true, false, 100, 36, 96, false, false, true, true},
      {"odd  TN 68x132x32", false, false, 68, 132, 32, false, false, false, false},
  };
  hipStream_t s;
  CK(hipStreamCreate(&s));
  int bad = 0;
  for (const Case &c : cases) {
    const int M = c.M, N = c.N, K = c.K;
    // host operands in their memory layouts
    std::vector<float> A((size_t)M * K), B((size_t)N * K), bias(N), gate((size_t)M * N);
    for (auto &x : A) x = frand();
    for (auto &x : B) x = frand();
    for (auto &x : bias) x = frand();
    for (auto &x : gate) x = frand();
    const int64_t lda = c.akc ? K : M, ldb = c.bkc ? K : N;
    auto a_at = [&](int m, int k) { return c.akc ? A[(size_t)m * K + k] : A[(size_t)k * M + m]; };
    auto b_at = [&](int n, int k) { return c.bkc ? B[(size_t)n * K + k] : B[(size_t)k * N + n]; };
    float *dA, *dB, *dC, *dbias, *dgate, *dcs;
    const int csr = (M + 31) / 32;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dgate, (size_t)M * N * 4)); CK(hipMalloc(&dcs, (size_t)csr * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgate, gate.data(), (size_t)M * N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4)); CK(hipMemset(dcs, 0xFF, (size_t)csr * N * 4));
    mlpg::Args G{};
    G.A = dA; G.lda = lda; G.B = dB; G.ldb = ldb; G.C = dC; G.ldc = N; G.M = M; G.N = N; G.K = K; G.bias = c.bias ? dbias : nullptr;
    G.relu = c.relu ? 1 : 0; G.gate = c.gate ? dgate : nullptr; G.ldg = N; G.colsum = c.colsum ? dcs : nullptr;
#ifdef MLPG_TIMING
    unsigned long long *ddbg;
    const int ntile = ((M + 63) / 64) * ((N + 63) / 64);
    CK(hipMalloc(&ddbg, (size_t)ntile * 64));
    CK(hipMemset(ddbg, 0, (size_t)ntile * 64));
    G.dbg = ddbg;
#endif
    launch(c, G, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> C((size_t)M * N), cs((size_t)csr * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cs.data(), dcs, cs.size() * 4, hipMemcpyDeviceToHost));
    // reference on a sample of rows (every row for the small cases), all columns, float64
    double maxerr = 0, maxref = 0;
    const int rstep = (M > 256) ? 37 : 1;
    for (int m = 0; m < M; m += rstep)
      for (int n = 0; n < N; n++) {
        double acc = 0;
        for (int k = 0; k < K; k++) acc += (double)a_at(m, k) * (double)b_at(n, k);
        if (c.bias) acc += bias[n];
        if (c.relu) acc = acc > 0 ? acc : 0;
        if (c.gate) acc = gate[(size_t)m * N + n] > 0 ? acc : 0;
        maxerr = fmax(maxerr, fabs(acc - (double)C[(size_t)m * N + n]));
        maxref = fmax(maxref, fabs(acc));
      }
    double cserr = 0;
    if (c.colsum)
      for (int b = 0; b < csr; b++)
        for (int n = 0; n < N; n++) {
          double t = 0;
          for (int m = 32 * b; m < 32 * b + 32 && m < M; m++) t += C[(size_t)m * N + n];
          cserr = fmax(cserr, fabs(t - cs[(size_t)b * N + n]));
        }
    const bool ok = maxerr < 2e-4 * fmax(1.0, maxref) * sqrt((double)K / 1024.0 + 1.0) && cserr < 1e-3;
    bad += !ok;
    // timing: 200 launches between one event pair
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch(c, G, s);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; i++) launch(c, G, s);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 200, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-38s %s max|err| %.2e (max|ref| %.1f) colsum err %.1e  %7.2f us  %6.1f TFLOP/s\n", c.name, ok ? "ok  " : "FAIL", maxerr,
           maxref, cserr, us, tf);
#ifdef MLPG_TIMING
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
             pro / ntile, loop / ntile, loop / ntile / (K / 32), epi / ntile, rt / ntile / 100.0, (loop / ntile) / (rt / ntile * 10.0), (rmax - rmin) / 100.0);
      CK(hipFree(ddbg));
    }
#endif
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dbias)); CK(hipFree(dgate)); CK(hipFree(dcs));
  }
  // the backward pair of one hidden layer in one launch: dW = dz^T h_prev and dh = (dz W) * gate with column sums
  {
    const int Bn = 1024, H = 1024;
    std::vector<float> dz((size_t)Bn * H), hp((size_t)Bn * H), W((size_t)H * H), gate((size_t)Bn * H);
    for (auto &x : dz) x = frand();
    for (auto &x : hp) x = frand();
    for (auto &x : W) x = frand();
    for (auto &x : gate) x = frand();
    float *ddz, *dhp, *dW_, *dgate, *dGW, *ddh, *dcs;
    CK(hipMalloc(&ddz, dz.size() * 4)); CK(hipMalloc(&dhp, hp.size() * 4)); CK(hipMalloc(&dW_, W.size() * 4)); CK(hipMalloc(&dgate, gate.size() * 4));
    CK(hipMalloc(&dGW, (size_t)H * H * 4)); CK(hipMalloc(&ddh, (size_t)Bn * H * 4)); CK(hipMalloc(&dcs, (size_t)(Bn / 32) * H * 4));
    CK(hipMemcpy(ddz, dz.data(), dz.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dhp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW_, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgate, gate.data(), gate.size() * 4, hipMemcpyHostToDevice));
    // dW[o][i] = sum_b dz[b][o] hp[b][i]: A = dz as MC (k = b, m = o), B = hp as MC (k = b, n = i)
    mlpg::Args GW{};
    GW.A = ddz; GW.lda = H; GW.B = dhp; GW.ldb = H; GW.C = dGW; GW.ldc = H; GW.M = H; GW.N = H; GW.K = Bn;
    // dh[b][i] = sum_o dz[b][o] W[o][i]: A = dz as KC (m = b, k = o), B = W as MC (k = o, n = i)
    mlpg::Args GH{};
    GH.A = ddz; GH.lda = H; GH.B = dW_; GH.ldb = H; GH.C = ddh; GH.ldc = H; GH.M = Bn; GH.N = H; GH.K = H; GH.gate = dgate; GH.ldg = H; GH.colsum = dcs;
#ifdef MLPG_TIMING
    unsigned long long *dbgW, *dbgH;
    CK(hipMalloc(&dbgW, 256 * 64)); CK(hipMalloc(&dbgH, 256 * 64));
    GW.dbg = dbgW; GH.dbg = dbgH;
#endif
    auto pair = [&]() { hipLaunchKernelGGL(mlpg::k_mlp_gemm_bwd_pair, dim3(256, 2), dim3(mlpg::THREADS), 0, s, GW, GH); };
    pair();
    CK(hipStreamSynchronize(s));
    std::vector<float> oW((size_t)H * H), oh((size_t)Bn * H);
    CK(hipMemcpy(oW.data(), dGW, oW.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(oh.data(), ddh, oh.size() * 4, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0;
    for (int o = 0; o < H; o += 41)
      for (int i = 0; i < H; i++) {
        double a = 0, b = 0;
        for (int k = 0; k < Bn; k++) a += (double)dz[(size_t)k * H + o] * hp[(size_t)k * H + i];
        for (int k = 0; k < H; k++) b += (double)dz[(size_t)o * H + k] * W[(size_t)k * H + i];
        b = gate[(size_t)o * H + i] > 0 ? b : 0;
        e1 = fmax(e1, fabs(a - oW[(size_t)o * H + i]));
        e2 = fmax(e2, fabs(b - oh[(size_t)o * H + i]));
      }
    const bool ok = e1 < 3e-4 * 32 && e2 < 3e-4 * 32;
    bad += !ok;
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    for (int i = 0; i < 20; i++) pair();
    CK(hipEventRecord(ev0, s));
    for (int i = 0; i < 200; i++) pair();
    CK(hipEventRecord(ev1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, ev0, ev1));
    const double us = ms * 1e3 / 200;
#ifdef MLPG_TIMING
    for (unsigned long long *dp : {dbgW, dbgH}) {
      std::vector<unsigned long long> d(256 * 8);
      CK(hipMemcpy(d.data(), dp, d.size() * 8, hipMemcpyDeviceToHost));
      double pro = 0, loop = 0, epi = 0, rt = 0; unsigned long long rmin = ~0ull, rmax = 0;
      for (int t = 0; t < 256; t++) {
        pro += d[t * 8 + 1] - d[t * 8]; loop += d[t * 8 + 2] - d[t * 8 + 1]; epi += d[t * 8 + 3] - d[t * 8 + 2];
        rt += d[t * 8 + 6] - d[t * 8 + 5];
        if (d[t * 8 + 4] < rmin) rmin = d[t * 8 + 4];
        if (d[t * 8 + 7] > rmax) rmax = d[t * 8 + 7];
      }
      printf("    pair member: prologue %.0f  loop %.0f cycles (%.0f per chunk)  epilogue %.0f;  loop %.2f us => clock %.2f GHz;  first stamp -> last stamp %.2f us\n",
             pro / 256, loop / 256, loop / 256 / 32, epi / 256, rt / 256 / 100.0, (loop / 256) / (rt / 256 * 10.0), (rmax - rmin) / 100.0);
    }
#endif
    printf("%-38s %s max|err| dW %.2e dh %.2e  %7.2f us  %6.1f TFLOP/s (two products)\n", "bwd pair (dW TN + dh NN gate colsum)", ok ? "ok  " : "FAIL",
           e1, e2, us, 2.0 * 2.0 * 1024 * 1024 * 1024 / (us * 1e-6) / 1e12);
  }
  printf(bad ? "FAILED: %d case(s)\n" : "all ok\n", bad);
  return bad ? 1 : 0;
}
