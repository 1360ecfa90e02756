// brl_mlp_gemm.hip — translation unit of libbrl_hip.so: the fp32 MFMA GEMMs of the PPO minibatch step with fused epilogues
// (csrc/mlp_gemm.hpp) behind one C-ABI entry point, brl_mlp_gemm (include/brl_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "abi_common.hpp"
#include "heads_dw_role.hpp"
#include "mlp_gemm.hpp"

// dh = (dz W) * act'(h) of the layer below the top (on the backward chain) with the head's weight-gradient role (NOT on the chain:
// csrc/heads_dw_role.hpp) as extra workgroups of the same launch: blocks [0, tiles) are GEMM tiles — one per CU at the step's
// shape — the rest sit beside them as a second workgroup per CU.
template <int NB>
__global__ __launch_bounds__(mg::THREADS) void k_gemm64_dh_heads_dw(mg::Args G, HeadsBwdArgs A, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[mg::lds_floats<NB>()];
  const int b = (int)blockIdx.x;
  if (b < tiles) mg::gemm_tile<true, false, mg::EPI_GATE_COLSUM, NB>(G, lds, b, tiles);
  else heads_bwd_dw_block(A, b - tiles);
}

// Tile width of a launch: 64 x 64 tiles (one workgroup per CU at 1024 x 1024) or 64 x 32 (two per CU, and twice the tiles for
// the step's narrow products: dW_0 is 1024 x 480 = 128 tiles of 64 x 64 on a 256-CU chip).  BRL_GEMM_TILE_N = 32 / 64 forces one
// (A/B runs); default: 64 x 32 up to one 64 x 64 tile per CU.
static int tile_nb(int64_t m, int64_t n) {
  static const int forced = [] {
    const char *e = getenv("BRL_GEMM_TILE_N");
    return e ? atoi(e) : 0;
  }();
  if (forced == 32) return 1;
  if (forced == 64) return 2;
  // up to one 64 x 64 tile per CU (the step's 1024 x 1024 products): two narrower workgroups per CU run the step 2-3 us faster
  // than one wide one (operands cold in L2: one's load stalls sit under the other's MFMAs; profiles/r04/r04_experiments.txt §2b)
  return (((m + 63) / 64) * ((n + 63) / 64) <= 256) ? 1 : 2;
}

extern "C" int brl_mlp_gemm(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                            int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                            float *colsum, float *sqsum, void *stream) {
  NEED(a && b && c && m > 0 && n > 0 && k > 0, "a / b / c / m / n / k");
  NEED(layout >= BRL_GEMM_NT && layout <= BRL_GEMM_TN, "layout (BRL_GEMM_NT / _NN / _TN)");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(n % 4 == 0 && ldc % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc >= n, "n and the leading dimensions multiples of 4 (16-byte pieces)");
  const bool akc = layout != BRL_GEMM_TN, bkc = layout == BRL_GEMM_NT;
  NEED(!akc || k % 4 == 0, "k a multiple of 4 where it is the contiguous index of an operand");
  NEED(akc ? lda >= k : lda >= ((m + 3) & ~(int64_t)3), "lda (where m is contiguous in a: at least m rounded up to 4 — whole 16-byte pieces are read)");
  NEED(bkc ? ldb >= k : ldb >= n, "ldb");
  // every byte offset inside an operand is a 32-bit buffer offset
  NEED((akc ? m * lda : k * lda) < (1ll << 29) && (bkc ? n * ldb : k * ldb) < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED((epilogue == BRL_GEMM_EPI_NONE) || (epilogue == BRL_GEMM_EPI_BIAS_ACT && layout == BRL_GEMM_NT && bias) ||
           (epilogue == BRL_GEMM_EPI_GATE_COLSUM && layout == BRL_GEMM_NN && gate && ldg >= n && ldg % 4 == 0) ||
           (epilogue == BRL_GEMM_EPI_SQSUM && layout == BRL_GEMM_TN && sqsum),
       "epilogue (BIAS_ACT with NT + bias, GATE_COLSUM with NN + gate, SQSUM with TN + sqsum)");
  HIP_TRY(hipSetDevice(device));
  mg::Args G{};
  G.A = a; G.lda = lda; G.B = b; G.ldb = ldb; G.C = c; G.ldc = ldc;
  G.M = (int)m; G.N = (int)n; G.K = (int)k; G.act = act;
  G.bias = bias; G.gate = gate; G.ldg = ldg; G.colsum = colsum; G.sqsum = sqsum;
  const int nb = tile_nb(m, n);
  const unsigned tiles = (unsigned)(((m + 63) / 64) * ((n + 32 * nb - 1) / (32 * nb)));
  hipStream_t s = (hipStream_t)stream;
#define MG_LAUNCH(AK, BK_, E)                                                                                   \
  do {                                                                                                          \
    if (nb == 2) hipLaunchKernelGGL((mg::k_gemm64n<AK, BK_, E, 2>), dim3(tiles), dim3(mg::THREADS), 0, s, G);    \
    else hipLaunchKernelGGL((mg::k_gemm64n<AK, BK_, E, 1>), dim3(tiles), dim3(mg::THREADS), 0, s, G);            \
  } while (0)
  if (layout == BRL_GEMM_NT) {
    if (epilogue == BRL_GEMM_EPI_BIAS_ACT) MG_LAUNCH(true, true, mg::EPI_BIAS_ACT);
    else MG_LAUNCH(true, true, mg::EPI_NONE);
  } else if (layout == BRL_GEMM_NN) {
    if (epilogue == BRL_GEMM_EPI_GATE_COLSUM) MG_LAUNCH(true, false, mg::EPI_GATE_COLSUM);
    else MG_LAUNCH(true, false, mg::EPI_NONE);
  } else {
    if (epilogue == BRL_GEMM_EPI_SQSUM) MG_LAUNCH(false, false, mg::EPI_SQSUM);
    else MG_LAUNCH(false, false, mg::EPI_NONE);
  }
#undef MG_LAUNCH
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mlp_gemm_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                                  const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                                  const int64_t *k, void *stream) {
  NEED(count >= 1 && count <= mg::GROUP_MAX && a && lda && b && ldb && c && ldc && m && n && k, "count (1..16) / NULL array");
  NEED(layout >= BRL_GEMM_NT && layout <= BRL_GEMM_TN, "layout (BRL_GEMM_NT / _NN / _TN)");
  const bool akc = layout != BRL_GEMM_TN, bkc = layout == BRL_GEMM_NT;
  mg::GroupArgs GA{};
  GA.n = count;
  int64_t tiles64 = 0;
  for (int i = 0; i < count; i++) tiles64 += ((m[i] + 63) / 64) * ((n[i] + 63) / 64);
  const int nb = (tiles64 <= 256) ? 1 : 2;     // (brl_mlp_gemm's rule: 64 x 32 tiles up to one 64 x 64 tile per CU)
  for (int i = 0; i < count; i++) {
    NEED(a[i] && b[i] && c[i] && m[i] > 0 && n[i] > 0 && k[i] > 0, "a / b / c / m / n / k");
    NEED(m[i] < (1 << 24) && n[i] < (1 << 24) && k[i] < (1 << 24), "m / n / k below 2^24");
    NEED(n[i] % 4 == 0 && ldc[i] % 4 == 0 && lda[i] % 4 == 0 && ldb[i] % 4 == 0 && ldc[i] >= n[i], "n and the leading dimensions multiples of 4");
    NEED(!akc || k[i] % 4 == 0, "k a multiple of 4 where it is the contiguous index of an operand");
    NEED(akc ? lda[i] >= k[i] : lda[i] >= ((m[i] + 3) & ~(int64_t)3), "lda (where m is contiguous in a: at least m rounded up to 4)");
    NEED(bkc ? ldb[i] >= k[i] : ldb[i] >= n[i], "ldb");
    NEED((akc ? m[i] * lda[i] : k[i] * lda[i]) < (1ll << 29) && (bkc ? n[i] * ldb[i] : k[i] * ldb[i]) < (1ll << 29), "operands below 2 GB");
    mg::Args &G = GA.g[i];
    G.A = a[i]; G.lda = lda[i]; G.B = b[i]; G.ldb = ldb[i]; G.C = c[i]; G.ldc = ldc[i];
    G.M = (int)m[i]; G.N = (int)n[i]; G.K = (int)k[i];
    GA.first[i + 1] = GA.first[i] + (int)(((m[i] + 63) / 64) * ((n[i] + 32 * nb - 1) / (32 * nb)));
  }
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const unsigned tiles = (unsigned)GA.first[count];
#define MG_GROUP(AK, BK_)                                                                                         \
  do {                                                                                                            \
    if (nb == 2) hipLaunchKernelGGL((mg::k_gemm_group<AK, BK_, 2>), dim3(tiles), dim3(mg::THREADS), 0, s, GA);     \
    else hipLaunchKernelGGL((mg::k_gemm_group<AK, BK_, 1>), dim3(tiles), dim3(mg::THREADS), 0, s, GA);             \
  } while (0)
  if (layout == BRL_GEMM_NT) MG_GROUP(true, true);
  else if (layout == BRL_GEMM_NN) MG_GROUP(true, false);
  else MG_GROUP(false, false);
#undef MG_GROUP
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mlp_gemm_dh_heads_dw(int device, const float *dz, int64_t lddz, const float *w, int64_t ldw, float *out, int64_t ldo,
                                        int64_t m, int64_t n, int64_t k, int act, const float *gate, int64_t ldg, float *colsum,
                                        const float *dheads, const float *h, int64_t ldh, int64_t batch, int64_t hidden, int nsplit,
                                        float *dw_partials, float *db_partials, const float *loss_partials,
                                        const float *gram_partials, int64_t ngroups, const int32_t *row_index, float *stat_sums,
                                        float *gram_sums, void *stream) {
  NEED(dz && w && out && gate && m > 0 && n > 0 && k > 0, "dz / w / out / gate / m / n / k");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(n % 4 == 0 && k % 4 == 0 && lddz % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0, "n, k and the leading dimensions multiples of 4");
  NEED(lddz >= k && ldw >= n && ldo >= n && ldg >= n, "leading dimensions");
  NEED(m * lddz < (1ll << 29) && k * ldw < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED(batch > 0 && hidden > 0 && hidden % 256 == 0 && ldh >= hidden && ldh % 4 == 0, "batch / hidden (a multiple of 256) / ldh");
  NEED(dheads && h && dw_partials && db_partials, "NULL array");
  NEED(nsplit >= 1 && (batch + nsplit - 1) / nsplit <= 64, "nsplit: at most 64 rows per split");
  HIP_TRY(hipSetDevice(device));
  mg::Args G{};
  G.A = dz; G.lda = lddz; G.B = w; G.ldb = ldw; G.C = out; G.ldc = ldo;
  G.M = (int)m; G.N = (int)n; G.K = (int)k; G.act = act; G.gate = gate; G.ldg = ldg; G.colsum = colsum;
  HeadsBwdArgs A{};
  A.dheads = dheads; A.h = h; A.ldh = ldh; A.B = batch; A.H = (int)hidden; A.act = act; A.nsplit = nsplit;
  A.rows_per_split = (int)((batch + nsplit - 1) / nsplit);
  A.dWh_partials = dw_partials; A.dbh_partials = db_partials;
  A.blocks_a = (int)(hidden / HB_JT) * nsplit;
  const bool sums = gram_sums != nullptr;
  NEED(!sums || (loss_partials && gram_partials && ngroups > 0 && row_index && stat_sums), "statistics sums: partials / ngroups / row_index / stat_sums");
  A.loss_partials = loss_partials; A.gram_partials = gram_partials; A.ngroups = (int)ngroups; A.row_index = row_index;
  A.stat_sums = stat_sums; A.gram_sums = gram_sums;
  const int nb = tile_nb(m, n);
  const int tiles = (int)(((m + 63) / 64) * ((n + 32 * nb - 1) / (32 * nb)));
  const unsigned blocks = (unsigned)(tiles + A.blocks_a + (sums ? HB_GRAM_BLOCKS : 0));
  if (nb == 2) hipLaunchKernelGGL(k_gemm64_dh_heads_dw<2>, dim3(blocks), dim3(mg::THREADS), 0, (hipStream_t)stream, G, A, tiles);
  else hipLaunchKernelGGL(k_gemm64_dh_heads_dw<1>, dim3(blocks), dim3(mg::THREADS), 0, (hipStream_t)stream, G, A, tiles);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
