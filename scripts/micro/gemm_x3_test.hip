// Stand-alone check + timing of brl_amd/csrc/mlp_gemm_x3.hpp ("bf16x3": fp32 products as six bf16 MFMA products of three-piece
// operands) beside the exact-fp32 kernel of mlp_gemm.hpp on the same inputs: error of BOTH against a float64 CPU reference
// (VERDICT r05 next-2 gate (i): max |err| of bf16x3 <= 1.5 x the exact path's), then interleaved timings (gate (ii)).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I brl_amd/csrc -o scripts/micro/gemm_x3_test scripts/micro/gemm_x3_test.hip
//   ./scripts/micro/gemm_x3_test [rounds]        build variants: -DMGX_NPROD=9|6|3  -DMGX_NACC=3|2|1
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "mlp_gemm_x3.hpp"
#include "mlp_gemm_x3w.hpp"
#include "../../brl_amd/csrc/mlp_gemm_x3.hpp"   // (namespace mgs: the library's kernel; scripts/micro/mlp_gemm_x3s.hpp is its first form)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static uint32_t rs = 12345u;
static float urand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; }   // 24-bit uniform [-1, 1)
static float nrand() {   // ~N(0,1): sum of 12 uniforms
  float s = 0;
  for (int i = 0; i < 12; i++) s += urand() * 0.5f + 0.5f;
  return s - 6.0f;
}

struct Case { const char *name; bool akc, bkc; int M, N, K; int epi; int act; int dist; };
// dist 0: both uniform [-1, 1) (full 24-bit mantissas); 1: A = relu(N(0,1)) (activations), B = N(0, 1/sqrt(K)) (weights);
// 2: A = N(0,1) * 1e-4 (gradients), B as 1

static void launch_x3(const Case &c, const mg::Args &G, hipStream_t s) {
  const int tiles = ((G.M + 63) / 64) * ((G.N + 63) / 64);
#define L(a, b, e) hipLaunchKernelGGL((mgx::k_gemm_x3<a, b, e>), dim3(tiles), dim3(mgx::THREADS), 0, s, G)
  if (c.akc && c.bkc) { if (c.epi == mg::EPI_BIAS_ACT) L(true, true, mg::EPI_BIAS_ACT); else L(true, true, mg::EPI_NONE); }
  else if (c.akc && !c.bkc) { if (c.epi == mg::EPI_GATE_COLSUM) L(true, false, mg::EPI_GATE_COLSUM); else L(true, false, mg::EPI_NONE); }
  else if (!c.akc && !c.bkc) { if (c.epi == mg::EPI_SQSUM) L(false, false, mg::EPI_SQSUM); else L(false, false, mg::EPI_NONE); }
  else { printf("layout not instantiated\n"); exit(1); }
#undef L
}
static float *g_slabs = nullptr;
static unsigned *g_tickets = nullptr;
static void launch_x3s(const Case &c, const mg::Args &G, hipStream_t s) {
  mgs::Args X{};
  X.g = G; X.slabs = g_slabs; X.tickets = g_tickets;
  X.splitk = getenv("X3S_SK") ? atoi(getenv("X3S_SK")) : mgs::pick_splitk(G.M, G.N, G.K);
  const int tiles = ((G.M + 127) / 128) * ((G.N + 127) / 128) * X.splitk;
#define L(a, b, e) hipLaunchKernelGGL((mgs::k_gemm_x3s<a, b, e>), dim3(tiles), dim3(mgs::THREADS), 0, s, X)
  if (c.akc && c.bkc) { if (c.epi == mg::EPI_BIAS_ACT) L(true, true, mg::EPI_BIAS_ACT); else L(true, true, mg::EPI_NONE); }
  else if (c.akc && !c.bkc) { if (c.epi == mg::EPI_GATE_COLSUM) L(true, false, mg::EPI_GATE_COLSUM); else L(true, false, mg::EPI_NONE); }
  else if (!c.akc && !c.bkc) L(false, false, mg::EPI_NONE);
  else { printf("layout not instantiated\n"); exit(1); }
#undef L
}
static bool launch_x3w(const Case &c, const mg::Args &G, hipStream_t s) {
  const int tiles = ((G.M + 63) / 64) * ((G.N + 63) / 64);
  if (!(c.akc && c.bkc)) return false;
  if (c.epi == mg::EPI_BIAS_ACT) hipLaunchKernelGGL((mgw::k_gemm_x3w_nt<mg::EPI_BIAS_ACT>), dim3(tiles), dim3(mgw::THREADS), 0, s, G);
  else hipLaunchKernelGGL((mgw::k_gemm_x3w_nt<mg::EPI_NONE>), dim3(tiles), dim3(mgw::THREADS), 0, s, G);
  return true;
}
static void launch_f32(const Case &c, const mg::Args &G, hipStream_t s, int nb) {
  const int tiles = ((G.M + 63) / 64) * ((G.N + 32 * nb - 1) / (32 * nb));
#define L(a, b, e) do { if (nb == 2) hipLaunchKernelGGL((mg::k_gemm64n<a, b, e, 2>), dim3(tiles), dim3(mg::THREADS), 0, s, G); \
                        else hipLaunchKernelGGL((mg::k_gemm64n<a, b, e, 1>), dim3(tiles), dim3(mg::THREADS), 0, s, G); } while (0)
  if (c.akc && c.bkc) { if (c.epi == mg::EPI_BIAS_ACT) L(true, true, mg::EPI_BIAS_ACT); else L(true, true, mg::EPI_NONE); }
  else if (c.akc && !c.bkc) { if (c.epi == mg::EPI_GATE_COLSUM) L(true, false, mg::EPI_GATE_COLSUM); else L(true, false, mg::EPI_NONE); }
  else if (!c.akc && !c.bkc) { if (c.epi == mg::EPI_SQSUM) L(false, false, mg::EPI_SQSUM); else L(false, false, mg::EPI_NONE); }
  else { printf("layout not instantiated\n"); exit(1); }
#undef L
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int pad = argc > 2 ? atoi(argv[2]) : 0;     // floats added to the operands' leading dimensions (L2 channel experiment)
  printf("bf16x3: NPROD %d  EXP %d / x3w EXP %d NACC %d / x3s EXP %d (%s)  ld pad %d\n", MGX_NPROD, MGX_EXP, MGW_EXP, MGW_NACC, MGS_EXP, getenv("X3S") ? "x3s: 128 x 128 tiles + split K" : getenv("X3W") ? "x3w" : "x3", pad);
  const Case cases[] = {
      {"NT 1024^3 uniform", true, true, 1024, 1024, 1024, mg::EPI_NONE, 0, 0},
      {"NT 1024^3 relu(N) x N/sqrt(K) +bias relu", true, true, 1024, 1024, 1024, mg::EPI_BIAS_ACT, 0, 1},
      {"NN 1024^3 1e-4 N x N/sqrt(K) gate colsum", true, false, 1024, 1024, 1024, mg::EPI_GATE_COLSUM, 0, 2},
      {"TN 1024^3 1e-4 N x relu(N) sqsum", false, false, 1024, 1024, 1024, mg::EPI_SQSUM, 0, 3},
      {"TN 1024x480x1024 sqsum (edge)", false, false, 1024, 480, 1024, mg::EPI_SQSUM, 0, 3},
      {"NT 1024x1024x480 uniform +bias relu", true, true, 1024, 1024, 480, mg::EPI_BIAS_ACT, 0, 0},
      {"NT 8192x1024x1024 relu(N) +bias relu", true, true, 8192, 1024, 1024, mg::EPI_BIAS_ACT, 0, 1},
      {"NT 200x72x64 uniform +bias tanh (edges)", true, true, 200, 72, 64, mg::EPI_BIAS_ACT, 1, 0},
      {"NT 48x256x992 uniform (31 chunks)", true, true, 48, 256, 992, mg::EPI_NONE, 0, 0},
      {"NN 100x36x96 uniform gate tanh colsum", true, false, 100, 36, 96, mg::EPI_GATE_COLSUM, 1, 0},
      {"NN 1000x256x256 gate colsum", true, false, 1000, 256, 256, mg::EPI_GATE_COLSUM, 0, 0},
      {"TN 68x132x992 uniform sqsum", false, false, 68, 132, 992, mg::EPI_SQSUM, 0, 0},
      {"TN 64x64x32 uniform one chunk", false, false, 64, 64, 32, mg::EPI_NONE, 0, 0},
      {"NT 64x64x64 two chunks", true, true, 64, 64, 64, mg::EPI_NONE, 0, 0},
  };
  hipStream_t s;
  CK(hipStreamCreate(&s));
  CK(hipMalloc(&g_slabs, (size_t)4096 * 128 * 128 * 4)); CK(hipMalloc(&g_tickets, 4096 * 4)); CK(hipMemset(g_tickets, 0, 4096 * 4));   // (x3s: up to 4096 slabs)
  int bad = 0;
  for (const Case &c : cases) {
    const int M = c.M, N = c.N, K = c.K;
    // leading dimensions: KC operand [rows][K + pad], MC operand [K][rows + pad]
    const int lda = (c.akc ? K : M) + pad, ldb = (c.bkc ? K : N) + pad;
    std::vector<float> A((size_t)(c.akc ? M : K) * lda), B((size_t)(c.bkc ? N : K) * ldb), bias(N), gate((size_t)M * N);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto &x : A) x = c.dist == 0 ? urand() : c.dist == 1 ? fmaxf(nrand(), 0.0f) : nrand() * 1e-4f;
    for (auto &x : B) x = c.dist == 0 ? urand() : c.dist == 3 ? fmaxf(nrand(), 0.0f) : nrand() * wsc;
    for (auto &x : bias) x = urand() * (c.dist == 0 ? 1.0f : 0.1f);
    for (auto &x : gate) x = urand();
    auto a_at = [&](int m, int k) { return c.akc ? A[(size_t)m * lda + k] : A[(size_t)k * lda + m]; };
    auto b_at = [&](int n, int k) { return c.bkc ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]; };
    const int csr = (M + 63) / 64, ntile = csr * ((N + 63) / 64);
    float *dA, *dB, *dC, *dC2, *dbias, *dgate, *dcs, *dsq;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dgate, (size_t)M * N * 4)); CK(hipMalloc(&dcs, (size_t)csr * N * 4)); CK(hipMalloc(&dsq, ntile * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgate, gate.data(), (size_t)M * N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4)); CK(hipMemset(dC2, 0xFF, (size_t)M * N * 4));
    CK(hipMemset(dcs, 0xFF, (size_t)csr * N * 4)); CK(hipMemset(dsq, 0xFF, ntile * 4));
    mg::Args G{};
    G.A = dA; G.lda = lda; G.B = dB; G.ldb = ldb; G.C = dC; G.ldc = N; G.M = M; G.N = N; G.K = K; G.act = c.act; G.bias = dbias;
    G.gate = dgate; G.ldg = N; G.colsum = dcs; G.sqsum = dsq;
    mg::Args G2 = G;
    G2.C = dC2; G2.colsum = nullptr; G2.sqsum = nullptr;
    if (c.epi == mg::EPI_SQSUM) { float *d2; CK(hipMalloc(&d2, 4 * ntile * 4)); G2.sqsum = d2; }
#ifdef MG_TIMING
    unsigned long long *ddbg;
    CK(hipMalloc(&ddbg, (size_t)ntile * 64));
    CK(hipMemset(ddbg, 0, (size_t)ntile * 64));
    G.dbg = ddbg;
    G2.dbg = nullptr;
#endif
    const bool usew = getenv("X3W") && c.akc && c.bkc;
    const bool uses = getenv("X3S") != nullptr;
    if (uses) launch_x3s(c, G, s); else if (usew) launch_x3w(c, G, s); else launch_x3(c, G, s);
    launch_f32(c, G2, s, 2);
    CK(hipStreamSynchronize(s));
    std::vector<float> C((size_t)M * N), C2((size_t)M * N);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> cs((size_t)csr * N), sqv(ntile);
    CK(hipMemcpy(cs.data(), dcs, cs.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sqv.data(), dsq, sqv.size() * 4, hipMemcpyDeviceToHost));
    // float64 reference on a sample of rows (every rstep-th), all columns
    double e3 = 0, e32 = 0, q3 = 0, q32 = 0, maxref = 0, absdot = 0;
    size_t cnt = 0, unstored = 0;
    const int rstep = (M > 256) ? 13 : 1;
    std::vector<double> arow(K);
    for (int m = 0; m < M; m += rstep) {
      for (int k = 0; k < K; k++) arow[k] = a_at(m, k);
      for (int n = 0; n < N; n++) {
        double acc = 0, ab = 0;
        if (c.bkc) { const float *b = &B[(size_t)n * ldb]; for (int k = 0; k < K; k++) { const double t = arow[k] * (double)b[k]; acc += t; ab += fabs(t); } }
        else for (int k = 0; k < K; k++) { const double t = arow[k] * (double)B[(size_t)k * ldb + n]; acc += t; ab += fabs(t); }
        if (c.epi == mg::EPI_BIAS_ACT) { acc += bias[n]; acc = c.act == 0 ? (acc > 0 ? acc : 0) : tanh(acc); }
        if (c.epi == mg::EPI_GATE_COLSUM) {
          const double hh = gate[(size_t)m * N + n];
          acc = c.act == 0 ? (hh > 0 ? acc : 0) : acc * (1.0 - hh * hh);
        }
        const double d3 = fabs(acc - (double)C[(size_t)m * N + n]), d32 = fabs(acc - (double)C2[(size_t)m * N + n]);
        if (!(d3 == d3) || !(d32 == d32)) unstored++;
        e3 = fmax(e3, d3); e32 = fmax(e32, d32); q3 += d3 * d3; q32 += d32 * d32;
        maxref = fmax(maxref, fabs(acc)); absdot = fmax(absdot, ab);
        cnt++;
      }
    }
    // the epilogue's sums against the STORED values (relative)
    double cserr = 0, sqerr = 0;
    if (c.epi == mg::EPI_GATE_COLSUM)
      for (int b = 0; b < csr; b++)
        for (int n = 0; n < N; n++) {
          double tt = 0, ta = 0;
          for (int m = 64 * b; m < 64 * b + 64 && m < M; m++) { tt += C[(size_t)m * N + n]; ta += fabs(C[(size_t)m * N + n]); }
          cserr = fmax(cserr, fabs(tt - cs[(size_t)b * N + n]) / fmax(ta, 1e-30));
        }
    if (c.epi == mg::EPI_SQSUM && !uses) {
      double tt = 0, u = 0;
      for (size_t i = 0; i < C.size(); i++) tt += (double)C[i] * C[i];
      for (int i = 0; i < ntile; i++) u += sqv[i];
      sqerr = fabs(tt - u) / fmax(1e-300, tt);
    }
    if (!(cserr < 1e-5) || !(sqerr < 1e-5)) unstored++;
    const bool ok = unstored == 0 && e3 <= 1.5 * e32 + 1e-30;
    bad += !ok;
    // timing: interleaved rounds, 200 launches between one event pair each
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> t3, t32, t32n;
    for (int i = 0; i < 20; i++) { if (uses) launch_x3s(c, G, s); else if (usew) launch_x3w(c, G, s); else launch_x3(c, G, s); launch_f32(c, G2, s, 2); launch_f32(c, G2, s, 1); }
    for (int r = 0; r < rounds; r++) {
      float ms;
      for (int which = 0; which < 3; which++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; i++) { if (which == 0) { if (uses) launch_x3s(c, G, s); else if (usew) launch_x3w(c, G, s); else launch_x3(c, G, s); } else launch_f32(c, G2, s, which == 1 ? 2 : 1); }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        (which == 0 ? t3 : which == 1 ? t32 : t32n).push_back(ms * 1e3 / 200);
      }
    }
    std::sort(t3.begin(), t3.end()); std::sort(t32.begin(), t32.end()); std::sort(t32n.begin(), t32n.end());
    const double u3 = t3[t3.size() / 2], u32 = t32[t32.size() / 2], u32n = t32n[t32n.size() / 2];
    printf("%-42s %s  max|err| x3 %.3e f32 %.3e (ratio %.2f)  rms x3 %.3e f32 %.3e  max|ref| %.2f max sum|ab| %.1f%s\n"
           "    %-38s x3 %6.2f us (%6.1f TF fp32-equivalent)  |  exact fp32 64x64 %6.2f us  64x32 %6.2f us (%6.1f TF)\n",
           uses ? c.name : usew ? "[x3w]" : c.name, ok ? "ok  " : "FAIL", e3, e32, e3 / fmax(e32, 1e-300), sqrt(q3 / cnt), sqrt(q32 / cnt), maxref, absdot,
           unstored ? "  NaN / unstored outputs / epilogue sums wrong!" : "", "", u3, 2.0 * M * N * K / (u3 * 1e-6) / 1e12, u32, u32n,
           2.0 * M * N * K / (fmin(u32, u32n) * 1e-6) / 1e12);
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
  printf(bad ? "gate (i) FAILED in %d case(s)\n" : "gate (i) holds in every case\n", bad);
  return 0;
}
