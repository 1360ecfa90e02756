// brl_mlp_gemm.hip — translation unit of libbrl_hip.so: the fp32 MFMA GEMMs of the PPO minibatch step with fused epilogues
// (csrc/mlp_gemm.hpp) behind one C-ABI entry point, brl_mlp_gemm (include/brl_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "abi_common.hpp"
#include "adam_role.hpp"
#include "heads_dw_role.hpp"
#include "mlp_gemm.hpp"

// A forward layer of step i + 1 (bias + activation in the epilogue) with the part of step i's Adam sweep that updates the NEXT
// layer's weights as extra workgroups of the same launch (csrc/adam_role.hpp): blocks [0, tiles) are GEMM tiles — one per CU at
// the step's shape — the rest sit beside them as a second workgroup per CU, HBM-bound beside MFMA-bound.
template <int NB>
__global__ __launch_bounds__(mg::THREADS) void k_gemm64_fwd_adam(mg::Args G, AdamRange R, int tiles, int riders) {
  __shared__ __attribute__((aligned(16))) float lds[mg::lds_floats<NB>()];
  const int b = (int)blockIdx.x;
  if (b < tiles) mg::gemm_tile<true, true, mg::EPI_BIAS_ACT, NB>(G, lds, b, tiles);
  else adam_range_block(R, b - tiles, riders, lds);
}

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

// The two backward products of one hidden layer — both fed by dz_l and h_{l-1} — in ONE launch: blocks [0, th) are tiles of
// dz_{l-1} = (dz_l W_l) * act'(h_{l-1}) (+ the bias gradient's tile sums), blocks [th, th + tw) tiles of dW_l = dz_l^T h_{l-1}
// (+ its tile square sums), the rest (optional) the head's weight-gradient role.  Two or more workgroups per CU: one tile's
// prologue, barrier waits and epilogue sit under another's MFMAs — 2 x 1024^3 in ~33 us where the two launches took 21 + 16.
template <int NB>
__global__ __launch_bounds__(mg::THREADS) void k_gemm64_bwd_pair(mg::Args GH, mg::Args GW, HeadsBwdArgs A, int th, int tw) {
  __shared__ __attribute__((aligned(16))) float lds[mg::lds_floats<NB>()];
  const int b = (int)blockIdx.x;
  // (the two kinds alternate in groups of 8 blocks while both last — every CU gets some of each, and a tile index keeps its
  //  residue mod 8 = its XCD, which gemm_tile's tile mapping relies on for L2 locality; speed only)
  const int mn = (th < tw ? th : tw) & ~7, both = 2 * mn;
  int kind, idx;
  if (b < both) { kind = (b >> 3) & 1; idx = ((b >> 4) << 3) + (b & 7); }
  else if (b < th + tw) {
    const int r = b - both;                    // what is left of each kind, the gate product first
    if (r < th - mn) { kind = 0; idx = mn + r; } else { kind = 1; idx = mn + r - (th - mn); }
  } else { kind = 2; idx = b - th - tw; }
  if (kind == 0) mg::gemm_tile<true, false, mg::EPI_GATE_COLSUM, NB>(GH, lds, idx, th);
  else if (kind == 1) mg::gemm_tile<false, false, mg::EPI_SQSUM, NB>(GW, lds, idx, tw);
  else heads_bwd_dw_block(A, idx);
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
  NEED(akc ? lda >= k : (lda >= m && m % 4 == 0), "lda (and m a multiple of 4 where it is contiguous in a)");
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

extern "C" int brl_mlp_gemm_adam(int device, const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, int64_t m,
                                 int64_t n, int64_t k, int act, const float *bias, float *p, const float *g, float *mom, float *var,
                                 int64_t lo, int64_t hi, const float *scratch, int npartials, const float *step, float lr,
                                 const float *lr_dev, float beta1, float beta2, float eps, float max_norm, float grad_scale,
                                 const int32_t *pending, void *stream) {
  NEED(a && b && c && bias && m > 0 && n > 0 && k > 0, "a / b / c / bias / m / n / k");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(n % 4 == 0 && k % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && lda >= k && ldb >= k && ldc >= n,
       "n, k and the leading dimensions multiples of 4");
  NEED(m * lda < (1ll << 29) && n * ldb < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED(p && g && mom && var && scratch && step && pending && npartials > 0, "p / g / m / v / scratch / step / pending / npartials");
  NEED(lo % 4 == 0 && hi % 4 == 0 && lo >= 0 && lo < hi, "range (multiples of 4)");
  // the weights this launch READS must not be the ones it updates
  NEED(b + n * ldb <= p + lo || b >= p + hi, "the range must not overlap the layer's own weights");
  HIP_TRY(hipSetDevice(device));
  mg::Args G{};
  G.A = a; G.lda = lda; G.B = b; G.ldb = ldb; G.C = c; G.ldc = ldc; G.M = (int)m; G.N = (int)n; G.K = (int)k; G.act = act; G.bias = bias;
  AdamRange R{};
  R.p = p; R.g = g; R.m = mom; R.v = var; R.lo4 = lo >> 2; R.hi4 = hi >> 2; R.partials = scratch; R.npartials = npartials; R.step = step;
  R.lr_dev = lr_dev; R.lr = lr; R.b1 = beta1; R.b2 = beta2; R.eps = eps; R.max_norm = max_norm; R.gscale = grad_scale; R.pending = pending;
  const int nb = tile_nb(m, n);
  const int tiles = (int)(((m + 63) / 64) * ((n + 32 * nb - 1) / (32 * nb)));
  const int64_t n4 = (hi - lo) / 4;
  const int riders = (int)((n4 + 1023) / 1024);   // four float4s per thread: 256 riders for a 1024 x 1024 layer
  if (nb == 2) hipLaunchKernelGGL(k_gemm64_fwd_adam<2>, dim3((unsigned)(tiles + riders)), dim3(mg::THREADS), 0, (hipStream_t)stream, G, R, tiles, riders);
  else hipLaunchKernelGGL(k_gemm64_fwd_adam<1>, dim3((unsigned)(tiles + riders)), dim3(mg::THREADS), 0, (hipStream_t)stream, G, R, tiles, riders);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mlp_gemm_bwd_pair(int device, const float *dz, int64_t lddz, const float *w, int64_t ldw, const float *h_prev,
                                     int64_t ldh, float *dz_out, int64_t ldo, float *dw_out, int64_t lddw, int64_t batch, int64_t n_out,
                                     int64_t n_in, int act, float *colsum, float *sqsum, const float *dheads, const float *h_top,
                                     int64_t ldht, int64_t hidden, int nsplit, float *dwh_partials, float *dbh_partials,
                                     const float *loss_partials, const float *gram_partials, int64_t ngroups, const int32_t *row_index,
                                     float *stat_sums, float *gram_sums, void *stream) {
  NEED(dz && w && h_prev && dz_out && dw_out && batch > 0 && n_out > 0 && n_in > 0, "dz / w / h_prev / dz_out / dw_out / sizes");
  NEED(batch < (1 << 24) && n_out < (1 << 24) && n_in < (1 << 24), "sizes below 2^24");
  NEED(batch % 4 == 0 && n_out % 4 == 0 && n_in % 4 == 0 && lddz % 4 == 0 && ldw % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0 && lddw % 4 == 0,
       "sizes and leading dimensions multiples of 4");
  NEED(lddz >= n_out && ldw >= n_in && ldh >= n_in && ldo >= n_in && lddw >= n_in, "leading dimensions");
  NEED(batch * lddz < (1ll << 29) && n_out * ldw < (1ll << 29) && batch * ldh < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  const bool ride = dheads != nullptr;
  NEED(!ride || (h_top && dwh_partials && dbh_partials && hidden > 0 && hidden % 256 == 0 && ldht >= hidden && ldht % 4 == 0 &&
                 nsplit >= 1 && (batch + nsplit - 1) / nsplit <= 64), "the head's weight-gradient role: arrays / hidden / nsplit");
  HIP_TRY(hipSetDevice(device));
  // dz_{l-1} [batch, n_in] = dz [batch, n_out] w [n_out, n_in], gated by h_prev [batch, n_in]
  mg::Args GH{};
  GH.A = dz; GH.lda = lddz; GH.B = w; GH.ldb = ldw; GH.C = dz_out; GH.ldc = ldo; GH.M = (int)batch; GH.N = (int)n_in; GH.K = (int)n_out;
  GH.act = act; GH.gate = h_prev; GH.ldg = ldh; GH.colsum = colsum;
  // dW [n_out, n_in] = dz^T h_prev
  mg::Args GW{};
  GW.A = dz; GW.lda = lddz; GW.B = h_prev; GW.ldb = ldh; GW.C = dw_out; GW.ldc = lddw; GW.M = (int)n_out; GW.N = (int)n_in; GW.K = (int)batch;
  GW.act = act; GW.sqsum = sqsum;
  HeadsBwdArgs A{};
  int extra = 0;
  if (ride) {
    A.dheads = dheads; A.h = h_top; A.ldh = ldht; A.B = batch; A.H = (int)hidden; A.act = act; A.nsplit = nsplit;
    A.rows_per_split = (int)((batch + nsplit - 1) / nsplit);
    A.dWh_partials = dwh_partials; A.dbh_partials = dbh_partials;
    A.blocks_a = (int)(hidden / HB_JT) * nsplit;
    const bool sums = gram_sums != nullptr;
    NEED(!sums || (loss_partials && gram_partials && ngroups > 0 && row_index && stat_sums), "statistics sums: partials / ngroups / row_index / stat_sums");
    A.loss_partials = loss_partials; A.gram_partials = gram_partials; A.ngroups = (int)ngroups; A.row_index = row_index;
    A.stat_sums = stat_sums; A.gram_sums = gram_sums;
    extra = A.blocks_a + (sums ? HB_GRAM_BLOCKS : 0);
  }
  static const int forced = [] { const char *e = getenv("BRL_GEMM_PAIR_TILE_N"); return e ? atoi(e) : 0; }();
  const int nb = forced == 32 ? 1 : 2;   // (two 64 x 64 tiles per CU at the step's shape)
  const int th = (int)(((batch + 63) / 64) * ((n_in + 32 * nb - 1) / (32 * nb)));
  const int tw = (int)(((n_out + 63) / 64) * ((n_in + 32 * nb - 1) / (32 * nb)));
  const unsigned blocks = (unsigned)(th + tw + extra);
  if (nb == 2) hipLaunchKernelGGL(k_gemm64_bwd_pair<2>, dim3(blocks), dim3(mg::THREADS), 0, (hipStream_t)stream, GH, GW, A, th, tw);
  else hipLaunchKernelGGL(k_gemm64_bwd_pair<1>, dim3(blocks), dim3(mg::THREADS), 0, (hipStream_t)stream, GH, GW, A, th, tw);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

// The LAST hidden layer of the forward pass with its share of `actor(x), critic(x)` (src/models.py:30-33) in the epilogue: 64 x 64
// tiles, one partial head product [m, 39] per column tile — k_heads_loss adds bias + the parts in order, k_heads_product's launch
// (5 us, on the step's dependency chain) disappears.
extern "C" int brl_mlp_gemm_fwd_heads(int device, const float *a, int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                                      int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *head_w, int64_t ldhw,
                                      float *head_parts, int nparts, void *stream) {
  NEED(a && b && c && bias && head_w && head_parts && m > 0 && n > 0 && k > 0, "a / b / c / bias / head_w / head_parts / m / n / k");
  NEED(m < (1 << 24) && n < (1 << 24) && k < (1 << 24), "m / n / k below 2^24");
  NEED(n % 4 == 0 && k % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ldhw % 4 == 0 && lda >= k && ldb >= k && ldc >= n && ldhw >= n,
       "n, k and the leading dimensions multiples of 4");
  NEED(m * lda < (1ll << 29) && n * ldb < (1ll << 29), "operands below 2 GB");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED(nparts == (int)((n + 63) / 64) || nparts == (int)((n + 31) / 32), "nparts = ceil(n / 64) or ceil(n / 32): one part per column tile");
  const int nb = nparts == (int)((n + 63) / 64) ? 2 : 1;   // (the caller picks the tile width through the number of parts)
  HIP_TRY(hipSetDevice(device));
  mg::Args G{};
  G.A = a; G.lda = lda; G.B = b; G.ldb = ldb; G.C = c; G.ldc = ldc; G.M = (int)m; G.N = (int)n; G.K = (int)k; G.act = act; G.bias = bias;
  G.wh = head_w; G.ldwh = ldhw; G.hparts = head_parts;
  const unsigned tiles = (unsigned)(((m + 63) / 64) * nparts);
  if (nb == 2) hipLaunchKernelGGL((mg::k_gemm64n<true, true, mg::EPI_BIAS_ACT_HEADS, 2>), dim3(tiles), dim3(mg::THREADS), 0, (hipStream_t)stream, G);
  else hipLaunchKernelGGL((mg::k_gemm64n<true, true, mg::EPI_BIAS_ACT_HEADS, 1>), dim3(tiles), dim3(mg::THREADS), 0, (hipStream_t)stream, G);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
