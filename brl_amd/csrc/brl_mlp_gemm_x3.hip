// brl_mlp_gemm_x3.hip — translation unit of libbrl_hip.so: fp32 products as six bf16 MFMA products of three-piece operands
// (csrc/mlp_gemm_x3.hpp: 128 x 128 tiles, the split in registers, optional split K: brl_mlp_gemm_x3 / _group; csrc/mlp_linear_x3p.hpp: the
// inference layer on operands already split into planes: brl_split_planes, brl_linear_x3p) — include/brl_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abi_common.hpp"
#include "mlp_gemm_x3.hpp"
#include "mlp_linear_x3p.hpp"

static int64_t x3_tiles(int64_t m, int64_t n) { return ((m + 127) / 128) * ((n + 127) / 128); }

static int64_t x3_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  const int sk = mgs::pick_splitk(m, n, k);
  if (sk == 1) return 0;
  const int64_t tiles = x3_tiles(m, n);
  return ((tiles * 4 + 255) / 256) * 256 + tiles * sk * (int64_t)mgs::SLAB_FLOATS * 4;
}

extern "C" int brl_mlp_gemm_x3_workspace(int64_t m, int64_t n, int64_t k, int64_t *bytes) {
  NEED(bytes && m > 0 && n > 0 && k > 0, "bytes / m / n / k");
  *bytes = x3_workspace_bytes(m, n, k);
  return BRL_OK;
}

extern "C" int brl_mlp_gemm_x3(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                               int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                               float *colsum, void *workspace, int64_t workspace_bytes, void *stream) {
  NEED(a && b && c && m > 0 && n > 0 && k > 0, "a / b / c / m / n / k");
  NEED(layout >= BRL_GEMM_NT && layout <= BRL_GEMM_TN, "layout (BRL_GEMM_NT / _NN / _TN)");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(n % 4 == 0 && ldc % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc >= n, "n and the leading dimensions multiples of 4 (16-byte pieces)");
  const bool akc = layout != BRL_GEMM_TN, bkc = layout == BRL_GEMM_NT;
  NEED(k % 32 == 0, "k a multiple of 32 (whole 32-deep chunks; other k: brl_mlp_gemm)");
  NEED(akc ? lda >= k : lda >= ((m + 3) & ~(int64_t)3), "lda (where m is contiguous in a: at least m rounded up to 4 — whole 16-byte pieces are read)");
  NEED(bkc ? ldb >= k : ldb >= n, "ldb");
  NEED((akc ? m * lda : k * lda) < (1ll << 29) && (bkc ? n * ldb : k * ldb) < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED((epilogue == BRL_GEMM_EPI_NONE) || (epilogue == BRL_GEMM_EPI_BIAS_ACT && layout == BRL_GEMM_NT && bias) ||
           (epilogue == BRL_GEMM_EPI_GATE_COLSUM && layout == BRL_GEMM_NN && gate && ldg >= n && ldg % 4 == 0),
       "epilogue (NONE; BIAS_ACT with NT + bias; GATE_COLSUM with NN + gate)");
  NEED((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)gate | (uintptr_t)colsum) & 15) == 0, "16-byte alignment");
  HIP_TRY(hipSetDevice(device));
  mgs::Args X{};
  mg::Args &G = X.g;
  G.A = a; G.lda = lda; G.B = b; G.ldb = ldb; G.C = c; G.ldc = ldc;
  G.M = (int)m; G.N = (int)n; G.K = (int)k; G.act = act;
  G.bias = bias; G.gate = gate; G.ldg = ldg; G.colsum = colsum;
  // K slices: as many as fill the chip — if the caller brought the memory the partial tiles need (else one slice per tile)
  const int64_t tiles = x3_tiles(m, n);
  int sk = mgs::pick_splitk(m, n, k);
  const int64_t tick = ((tiles * 4 + 255) / 256) * 256;
  if (sk > 1 && (workspace == nullptr || workspace_bytes < tick + tiles * sk * (int64_t)mgs::SLAB_FLOATS * 4)) sk = 1;
  NEED(sk == 1 || (((uintptr_t)workspace) & 255) == 0, "workspace: 256-byte alignment");
  NEED(sk == 1 || tiles * sk * (int64_t)mgs::SLAB_FLOATS * 4 < (1ll << 31), "partial tiles below 2 GB");
  X.splitk = sk;
  X.tickets = (unsigned *)workspace;                                   // zero before the first launch (the caller's memset): put back by the last arriver
  X.slabs = sk > 1 ? (float *)((char *)workspace + tick) : nullptr;
  const unsigned blocks = (unsigned)(tiles * sk);
  hipStream_t s = (hipStream_t)stream;
#define X3_LAUNCH(AK, BK_, E) hipLaunchKernelGGL((mgs::k_gemm_x3s<AK, BK_, E>), dim3(blocks), dim3(mgs::THREADS), 0, s, X)
  if (layout == BRL_GEMM_NT) {
    if (epilogue == BRL_GEMM_EPI_BIAS_ACT) X3_LAUNCH(true, true, mg::EPI_BIAS_ACT);
    else X3_LAUNCH(true, true, mg::EPI_NONE);
  } else if (layout == BRL_GEMM_NN) {
    if (epilogue == BRL_GEMM_EPI_GATE_COLSUM) X3_LAUNCH(true, false, mg::EPI_GATE_COLSUM);
    else X3_LAUNCH(true, false, mg::EPI_NONE);
  } else {
    X3_LAUNCH(false, false, mg::EPI_NONE);
  }
#undef X3_LAUNCH
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mlp_gemm_x3_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                                     const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                                     const int64_t *k, void *stream) {
  NEED(count >= 1 && count <= mgs::GROUP_MAX && a && lda && b && ldb && c && ldc && m && n && k, "count (1..8) / NULL array");
  NEED(layout >= BRL_GEMM_NT && layout <= BRL_GEMM_TN, "layout (BRL_GEMM_NT / _NN / _TN)");
  const bool akc = layout != BRL_GEMM_TN, bkc = layout == BRL_GEMM_NT;
  mgs::GroupArgs GA{};
  GA.n = count;
  for (int i = 0; i < count; i++) {
    NEED(a[i] && b[i] && c[i] && m[i] > 0 && n[i] > 0 && k[i] > 0, "a / b / c / m / n / k");
    NEED(m[i] < (1 << 24) && n[i] < (1 << 24) && k[i] < (1 << 24), "m / n / k below 2^24");
    NEED(n[i] % 4 == 0 && ldc[i] % 4 == 0 && lda[i] % 4 == 0 && ldb[i] % 4 == 0 && ldc[i] >= n[i], "n and the leading dimensions multiples of 4");
    NEED(k[i] % 32 == 0, "k a multiple of 32 (whole 32-deep chunks; other k: brl_mlp_gemm_group)");
    NEED(akc ? lda[i] >= k[i] : lda[i] >= ((m[i] + 3) & ~(int64_t)3), "lda (where m is contiguous in a: at least m rounded up to 4)");
    NEED(bkc ? ldb[i] >= k[i] : ldb[i] >= n[i], "ldb");
    NEED((akc ? m[i] * lda[i] : k[i] * lda[i]) < (1ll << 29) && (bkc ? n[i] * ldb[i] : k[i] * ldb[i]) < (1ll << 29), "operands below 2 GB");
    NEED((((uintptr_t)a[i] | (uintptr_t)b[i] | (uintptr_t)c[i]) & 15) == 0, "16-byte alignment");
    mgs::Args &X = GA.x[i];
    mg::Args &G = X.g;
    G.A = a[i]; G.lda = lda[i]; G.B = b[i]; G.ldb = ldb[i]; G.C = c[i]; G.ldc = ldc[i];
    G.M = (int)m[i]; G.N = (int)n[i]; G.K = (int)k[i];
    X.splitk = 1;
    GA.first[i + 1] = GA.first[i] + (int)x3_tiles(m[i], n[i]);
  }
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)GA.first[count];
  if (layout == BRL_GEMM_NT) hipLaunchKernelGGL((mgs::k_gemm_x3s_group<true, true>), dim3(blocks), dim3(mgs::THREADS), 0, s, GA);
  else if (layout == BRL_GEMM_NN) hipLaunchKernelGGL((mgs::k_gemm_x3s_group<true, false>), dim3(blocks), dim3(mgs::THREADS), 0, s, GA);
  else hipLaunchKernelGGL((mgs::k_gemm_x3s_group<false, false>), dim3(blocks), dim3(mgs::THREADS), 0, s, GA);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_split_planes(int device, const float *x, int64_t n, uint16_t *planes, int64_t plane_stride, void *stream) {
  NEED(x && planes && n > 0 && n % 4 == 0 && plane_stride >= n && plane_stride % 4 == 0, "x / planes / n (a multiple of 4) / plane_stride");
  NEED((((uintptr_t)x) & 15) == 0 && (((uintptr_t)planes) & 7) == 0, "alignment (x 16 bytes, planes 8)");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(lx3::k_split_planes, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, planes, plane_stride, n / 4);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_linear_x3p(int device, const uint16_t *x_planes, int npx, int64_t ldx, int64_t sx, const uint16_t *w_planes, int64_t ldw,
                              int64_t sw, const float *bias, int relu, float *y, int64_t ldy, uint16_t *y_planes, int64_t ldyp, int64_t syp,
                              int64_t m, int64_t n, int64_t k, void *stream) {
  NEED(x_planes && w_planes && bias && (y || y_planes) && m > 0 && n > 0 && k > 0, "x_planes / w_planes / bias / an output / m / n / k");
  NEED(npx == 1 || npx == 3, "npx (3 planes, or 1: x exact in bf16)");
  NEED(n % 128 == 0 && k % 32 == 0, "n a multiple of 128, k a multiple of 32");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(ldx >= k && ldx % 8 == 0 && ldw >= k && ldw % 8 == 0, "ldx / ldw (at least k, multiples of 8: 16-byte pieces)");
  NEED(m * ldx < (1ll << 31) && n * ldw < (1ll << 31), "planes below 4 GB each");
  NEED(!y || (ldy >= n && ldy % 4 == 0), "ldy");
  NEED(!y_planes || (ldyp >= n && ldyp % 8 == 0 && syp % 8 == 0), "ldyp / syp (multiples of 8)");
  NEED(sx % 8 == 0 && sw % 8 == 0, "plane strides multiples of 8");
  NEED((((uintptr_t)x_planes | (uintptr_t)w_planes | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)y_planes) & 15) == 0, "16-byte alignment");
  HIP_TRY(hipSetDevice(device));
  lx3::Args G{};
  G.x = x_planes; G.ldx = ldx; G.sx = sx; G.w = w_planes; G.ldw = ldw; G.sw = sw; G.bias = bias;
  G.y = y; G.ldy = ldy; G.yp = y_planes; G.ldyp = ldyp; G.syp = syp;
  G.M = (int)m; G.N = (int)n; G.K = (int)k; G.relu = relu ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)(((m + 127) / 128) * (n / 128));
  if (npx == 3) hipLaunchKernelGGL(lx3::k_linear_x3p<3>, dim3(blocks), dim3(lx3::THREADS), 0, s, G);
  else hipLaunchKernelGGL(lx3::k_linear_x3p<1>, dim3(blocks), dim3(lx3::THREADS), 0, s, G);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
