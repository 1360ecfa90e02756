// brl_ppo.hip — translation unit of libbrl_hip.so: the PPO update's kernels (src/update.py:74-242: _loss_fn and its gradients, the
// 39-column head, minibatch gather, activation derivative + bias sums, clip_by_global_norm + Adam on flat buffers) and their
// C-ABI entry points (include/brl_hip.h).  No environment handle: every entry point takes the HIP device its arrays live on.
// The step's big products are the library's or csrc/brl_mlp_gemm.hip's.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "abi_common.hpp"

static inline unsigned thread_grid(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// ---- PPO-clip loss and its gradient w.r.t. the network outputs, one launch (src/update.py:90-167) ------------
// One wave per sample, lane a = action a.  Forward: masked log-softmax -> log-prob of the taken action, ratio,
// clipped surrogate; clipped value loss; entropy of the masked policy; approx-KL / clip-fraction.  Backward: the
// derivative of  loss_actor + vf_coef * value_loss - ent_coef * entropy  (means over the minibatch) w.r.t. logits
// and value — what autograd would hand to the last Linear layers, so torch only runs the GEMMs.
struct PpoArgs {
  const float *logits;
  int64_t ls;
  const float *value;
  const uint8_t *mask;
  const int32_t *action;
  const float *old_value, *old_logp, *gae, *tgt;
  int64_t B;
  float clip_eps, vf_coef, ent_coef;
  int masked, value_clipping;
  float *dlogits, *dvalue, *partials, *illp;
  int64_t vs, dls, dvs;  // strides of value, dlogits (row), dvalue: 1, 38, 1 for separate arrays; 39 each for the merged head
};

// Wave-wide sum / max, result on every lane.  DPP moves inside the rows of 16 lanes (quad swaps, row rotations: every lane ends
// with its row's value), row_bcast15 / row_bcast31 across the rows, v_readlane of lane 63: 6 register-to-register moves and one
// scalar read instead of 6 ds_bpermute round trips through the LDS crossbar (__shfl_xor) — the loss of one sample is a chain of
// ~12 such reductions.  (The order of the additions differs from a butterfly: same value up to fp32 rounding.)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_move_f(float old, float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_move_f<0xB1>(0.0f, v);            // quad_perm [1,0,3,2]
  v += dpp_move_f<0x4E>(0.0f, v);            // quad_perm [2,3,0,1]
  v += dpp_move_f<0x124>(0.0f, v);           // row_ror:4
  v += dpp_move_f<0x128>(0.0f, v);           // row_ror:8  -> every lane: its row's sum
  v += dpp_move_f<0x142, 0xA>(0.0f, v);      // row_bcast15 into rows 1, 3
  v += dpp_move_f<0x143, 0xC>(0.0f, v);      // row_bcast31 into rows 2, 3 -> lane 63: everything
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_move_f<0xB1>(ninf, v));
  v = fmaxf(v, dpp_move_f<0x4E>(ninf, v));
  v = fmaxf(v, dpp_move_f<0x124>(ninf, v));
  v = fmaxf(v, dpp_move_f<0x128>(ninf, v));
  v = fmaxf(v, dpp_move_f<0x142, 0xA>(ninf, v));
  v = fmaxf(v, dpp_move_f<0x143, 0xC>(ninf, v));
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// One sample (= one wave, lane a = action a): `lg` = the lane's logit (lanes >= 38: ignored), `v` = the critic's value, `g` =
// the advantage (already normalised when reward_scaling is on).  Writes dlogits / dvalue / illp of the sample and
// returns its five statistics terms in st[0..4] (valid on every lane).  `illp_lane` = this lane's illegal-action probability.
// (the sample's inputs are loaded by ppo_sample_load — callers issue it BEFORE whatever produces the logits, so that no global
//  load sits behind the barrier / GEMM in front of the loss)
struct PpoSampleIn {
  bool legal;           // this lane's action is legal
  int act;              // the action taken
  float old_logp, old_value, tgt, gae;
};
__device__ __forceinline__ PpoSampleIn ppo_sample_load(const PpoArgs &A, int64_t b, bool valid, int lane) {
  const int64_t bb = valid ? b : 0;
  PpoSampleIn S;
  S.legal = (lane < BRL_NUM_ACTIONS) && A.mask[bb * BRL_NUM_ACTIONS + ((lane < BRL_NUM_ACTIONS) ? lane : 0)] != 0;
  S.act = A.action[bb];
  S.old_logp = A.old_logp[bb];
  S.old_value = A.old_value[bb];
  S.tgt = A.tgt[bb];
  S.gae = A.gae[bb];
  return S;
}
__device__ __forceinline__ void ppo_loss_sample(const PpoArgs &A, const PpoSampleIn &S, int64_t b, bool valid, int lane, float lg, float v,
                                                float g, float (&st)[5], float &illp_lane) {
  const bool in = lane < BRL_NUM_ACTIONS;
  const bool legal = S.legal;
  const float invB = 1.0f / (float)A.B;
  // masked policy (src/update.py:12-16, 132-135): log-softmax over the legal actions
  const float mx = wave_max_f(legal ? lg : -INFINITY);
  const float lse = logf(wave_sum_f(legal ? expf(lg - mx) : 0.0f));
  const float lsm = legal ? (lg - mx) - lse : 0.0f;
  const float p = legal ? expf(lsm) : 0.0f;
  // the unmasked softmax: illegal-action probabilities (src/update.py:136-141) and, for the unmasked policy, log-prob
  const float mx2 = wave_max_f(in ? lg : -INFINITY);
  const float lse2 = logf(wave_sum_f(in ? expf(lg - mx2) : 0.0f));
  const float lsm2 = in ? (lg - mx2) - lse2 : 0.0f;
  const float p2 = in ? expf(lsm2) : 0.0f;
  const int act = S.act;
  const float lsel = A.masked ? lsm : lsm2, psel = A.masked ? p : p2;
  const float lp = __shfl(lsel, act & 63, 64);
  const float logratio = lp - S.old_logp;
  const float ratio = expf(logratio);
  const float eps = A.clip_eps;
  const float a1 = ratio * g, a2 = fminf(fmaxf(ratio, 1.0f - eps), 1.0f + eps) * g;
  const float la = -fminf(a1, a2);
  const bool inside = (ratio >= 1.0f - eps) && (ratio <= 1.0f + eps);
  const float dratio = ((a1 < a2) || inside) ? -g : 0.0f;  // d(-min(a1, a2)) / d ratio (ties: both branches agree)
  const float dlp = dratio * ratio * invB;
  // value loss (src/update.py:48-60)
  const float ov = S.old_value, t = S.tgt;
  float vl, dv;
  if (A.value_clipping) {
    const float dcl = fminf(fmaxf(v - ov, -eps), eps);
    const float vc = ov + dcl;
    const float l1 = (v - t) * (v - t), l2 = (vc - t) * (vc - t);
    vl = 0.5f * fmaxf(l1, l2);
    const bool unclipped = (v - ov >= -eps) && (v - ov <= eps);
    dv = (l1 >= l2) ? (v - t) : (unclipped ? (vc - t) : 0.0f);
  } else {
    vl = 0.5f * (v - t) * (v - t);
    dv = v - t;
  }
  // entropy of the masked policy, 0 log 0 = 0 (distrax)
  const float H = -wave_sum_f((legal && p > 0.0f) ? p * lsm : 0.0f);
  const float dH = legal ? -p * (lsm + H) : 0.0f;
  const float onehot = (lane == act) ? 1.0f : 0.0f;
  const bool live = A.masked ? legal : in;
  const float dz = (live ? dlp * (onehot - psel) : 0.0f) - A.ent_coef * invB * dH;
  illp_lane = legal ? 0.0f : p2;
  if (valid && in) {
    A.dlogits[b * A.dls + lane] = dz;
    if (A.illp) A.illp[b * BRL_NUM_ACTIONS + lane] = illp_lane;
  }
  if (lane == 0 && valid) A.dvalue[b * A.dvs] = A.vf_coef * dv * invB;
  st[0] = valid ? vl : 0.0f;
  st[1] = valid ? la : 0.0f;
  st[2] = valid ? H : 0.0f;
  st[3] = valid ? (ratio - 1.0f) - logratio : 0.0f;
  st[4] = (valid && fabsf(ratio - 1.0f) > eps) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void k_ppo_loss(PpoArgs A) {
  __shared__ float part[4][8];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  const bool valid = b < A.B;
  const int64_t bb = valid ? b : 0;
  const float lg = (lane < BRL_NUM_ACTIONS) ? A.logits[bb * A.ls + lane] : 0.0f;
  const PpoSampleIn S = ppo_sample_load(A, b, valid, lane);
  float st[5], ill;
  ppo_loss_sample(A, S, b, valid, lane, lg, A.value[bb * A.vs], S.gae, st, ill);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; k++) part[wave][k] = st[k];
  }
  __syncthreads();
  if (threadIdx.x < 8) {  // per-block partial sums in a fixed order (deterministic statistics)
    const int k = (int)threadIdx.x;
    A.partials[(int64_t)blockIdx.x * 8 + k] = (k < 5) ? ((part[0][k] + part[1][k]) + (part[2][k] + part[3][k])) : 0.0f;
  }
}

// The logged statistics of one minibatch step, one block: column sums of brl_ppo_loss's per-block partials / batch
// (fixed order: deterministic), total = loss_actor + vf_coef * value_loss - ent_coef * entropy, and the illegal-action
// norm: largest singular value / 2 of the non-negative [B, 38] matrix P = softmax(logits) * ~mask from its 38 x 38 Gram
// matrix G = P^T P (one small GEMM in torch): sqrt of G's top eigenvalue by 8 squarings (G^256 collapses onto the
// Perron vector) + a Rayleigh quotient — `jnp.linalg.norm(illegal_action_probabilities, ord=2) / 2`
// (src/update.py:138-141) without an SVD.
//   out[0] total  [1] value_loss  [2] loss_actor  [3] entropy  [4] approx_kl  [5] clipfrac  [6] illegal-action norm / 2
__global__ __launch_bounds__(512) void k_ppo_stats(const float *partials, int64_t nblk, int64_t batch, const float *G,
                                                   float vf_coef, float ent_coef, float *out, const int32_t *row_index) {
  if (row_index != nullptr) out += 8 * (int64_t)(*row_index);  // a log of [steps, 8] rows, indexed from device memory
  constexpr int D = BRL_NUM_ACTIONS, DD = D * D;
  __shared__ float g[DD], m[DD], t[DD], vec[D], red[2], st[8];
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (wv >= 3) {  // waves 3..7: one statistic each — lane l adds rows l, l + 64, ... in order, then a fixed butterfly
    const int k = wv - 3;
    float s = 0.0f;
    for (int64_t i = lane; i < nblk; i += 64) s += partials[i * 8 + k];
    s = wave_sum_f(s);
    if (lane == 0) st[k] = s / (float)batch;
  }
  for (int e = tid; e < DD; e += 512) g[e] = G ? G[e] : 0.0f;
  __syncthreads();
  auto trace_of = [&](const float *x) {  // wave 0, fixed butterfly
    if (wv == 0) {
      const float s = wave_sum_f((lane < D) ? x[lane * D + lane] : 0.0f);
      if (lane == 0) red[0] = fmaxf(s, 1.17549435e-38f);
    }
    __syncthreads();
    return red[0];
  };
  float tr = trace_of(g);
  for (int e = tid; e < DD; e += 512) m[e] = g[e] / tr;
  __syncthreads();
  for (int it = 0; it < 8; it++) {
    for (int e = tid; e < DD; e += 512) {
      const int i = e / D, j = e - i * D;
      float s = 0.0f;
#pragma unroll 2
      for (int k = 0; k < D; k++) s += m[i * D + k] * m[k * D + j];
      t[e] = s;
    }
    __syncthreads();
    tr = trace_of(t);
    for (int e = tid; e < DD; e += 512) m[e] = t[e] / tr;
    __syncthreads();
  }
  if (tid < D) {
    float s = 0.0f;
    for (int k = 0; k < D; k++) s += m[tid * D + k];
    vec[tid] = s;
  }
  __syncthreads();
  if (wv == 0) {  // Rayleigh quotient v^T G v / v^T v
    float gv = 0.0f;
    if (lane < D)
      for (int k = 0; k < D; k++) gv += g[lane * D + k] * vec[k];
    const float vi = (lane < D) ? vec[lane] : 0.0f;
    const float num = wave_sum_f(vi * gv), den = wave_sum_f(vi * vi);
    if (lane == 0) {
      out[6] = 0.5f * sqrtf(fmaxf(num / fmaxf(den, 1.17549435e-38f), 0.0f));
      out[0] = st[1] + vf_coef * st[0] - ent_coef * st[2];
      out[1] = st[0]; out[2] = st[1]; out[3] = st[2]; out[4] = st[3]; out[5] = st[4];
      out[7] = 0.0f;
    }
  }
}

#include "ppo_update.hpp"  // k_mb_gather_dev, k_relu_bwd_tiles4, k_bias_finalize, k_adam_norm_fin / k_adam_apply, k_shard_norm / k_shard_apply
#include "ppo_heads.hpp"   // k_heads_loss, k_heads_bwd, k_ppo_stats2: the 39-column head products and what hangs on them
#include "fair_chain.hpp"  // k_fair_chain: the FAIR network's forward + loss + backward chain, 16 samples per workgroup

// =====================================================================================
// C-ABI
// =====================================================================================
extern "C" int brl_ppo_loss(int device, const float *logits, int64_t logits_stride, const float *value, const uint8_t *mask,
                            const int32_t *action, const float *old_value, const float *old_log_prob, const float *gae,
                            const float *targets, int64_t batch, float clip_eps, float vf_coef, float ent_coef, int masked,
                            int value_clipping, float *dlogits, float *dvalue, float *partials, float *illegal_probs,
                            void *stream) {
  NEED(batch > 0, "batch");
  NEED(logits && value && mask && action && old_value && old_log_prob && gae && targets, "NULL input array");
  NEED(dlogits && dvalue && partials, "NULL output array");
  NEED(logits_stride >= BRL_NUM_ACTIONS, "logits_stride");
  HIP_TRY(hipSetDevice(device));
  PpoArgs A{logits, logits_stride, value, mask, action, old_value, old_log_prob, gae, targets, batch, clip_eps, vf_coef,
            ent_coef, masked, value_clipping, dlogits, dvalue, partials, illegal_probs, 1, BRL_NUM_ACTIONS, 1};
  hipLaunchKernelGGL(k_ppo_loss, dim3(thread_grid(batch, 4)), dim3(256), 0, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mb_gather_bind(int device, const brl_transition *flat, const float *adv, const float *targets, const int64_t *perm,
                                  const int32_t *mb_index, int64_t mbs, float *x0, uint8_t *mask, int32_t *action, float *old_value,
                                  float *old_log_prob, float *gae_out, float *targets_out, int64_t nsteps, void *args_dev,
                                  void *stream) {
  NEED(flat && flat->obs && flat->legal_action_mask && flat->action && flat->value && flat->log_prob, "trajectory");
  NEED(adv && targets && perm && mb_index && mbs > 0, "adv / targets / perm / mb_index / mbs");
  NEED(x0 && mask && action && old_value && old_log_prob && gae_out && targets_out && args_dev, "NULL output array / args_dev");
  HIP_TRY(hipSetDevice(device));
  static_assert(sizeof(GatherArgs) <= 256, "args_dev is 256 bytes");
  NEED(nsteps > 0, "nsteps (minibatches in perm)");
  GatherArgs A{flat->obs, flat->legal_action_mask, flat->action, flat->value, flat->log_prob, adv, targets, perm, mb_index, mbs,
               x0, mask, action, old_value, old_log_prob, gae_out, targets_out, nsteps};
  hipLaunchKernelGGL(k_mb_gather_bind, dim3(1), dim3(64), 0, (hipStream_t)stream, A, (GatherArgs *)args_dev);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_mb_gather_dev(int device, const void *args_dev, int64_t mbs, void *stream) {
  NEED(args_dev && mbs > 0, "args_dev / mbs");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_mb_gather_dev, dim3((unsigned)mbs), dim3(128), 0, (hipStream_t)stream, (const GatherArgs *)args_dev);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_act_bwd_colsum(int device, float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld, int act,
                                  float *scratch, void *stream) {
  NEED(dh && h && scratch && rows > 0 && cols > 0 && ld >= cols, "dh / h / scratch / rows / cols / ld");
  NEED(cols % 4 == 0 && ld % 4 == 0, "cols and ld multiples of 4");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  HIP_TRY(hipSetDevice(device));
  const int64_t tiles = (rows + 15) / 16;
  hipLaunchKernelGGL(k_relu_bwd_tiles4, dim3((unsigned)((cols + 255) / 256), (unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                     dh, h, rows, cols, ld, scratch, act);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_act_bwd_colsum_heads_dw(int device, float *dz, const float *hh, int64_t rows, int64_t cols, int64_t ld, int act,
                                           float *scratch, const float *dheads, const float *h, int64_t ldh, int64_t batch,
                                           int64_t hidden, int nsplit, float *dw_partials, float *db_partials,
                                           const float *loss_partials, const float *gram_partials, int64_t ngroups,
                                           const int32_t *row_index, float *stat_sums, float *gram_sums, void *stream) {
  NEED(dz && hh && scratch && rows > 0 && cols > 0 && ld >= cols, "dz / h / scratch / rows / cols / ld");
  NEED(cols % 4 == 0 && ld % 4 == 0, "cols and ld multiples of 4");
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED(batch > 0 && hidden > 0 && hidden % 256 == 0 && ldh >= hidden && ldh % 4 == 0, "batch / hidden (a multiple of 256) / ldh");
  NEED(dheads && h && dw_partials && db_partials, "NULL array");
  NEED(nsplit >= 1 && (batch + nsplit - 1) / nsplit <= 64, "nsplit: at most 64 rows per split");
  HIP_TRY(hipSetDevice(device));
  HeadsBwdArgs A{};
  A.dheads = dheads; A.h = h; A.ldh = ldh; A.B = batch; A.H = (int)hidden; A.act = act; A.nsplit = nsplit;
  A.rows_per_split = (int)((batch + nsplit - 1) / nsplit);
  A.dWh_partials = dw_partials; A.dbh_partials = db_partials;
  A.blocks_a = (int)(hidden / HB_JT) * nsplit;
  const bool sums = gram_sums != nullptr;
  NEED(!sums || (loss_partials && gram_partials && ngroups > 0 && row_index && stat_sums), "statistics sums: partials / ngroups / row_index / stat_sums");
  A.loss_partials = loss_partials; A.gram_partials = gram_partials; A.ngroups = (int)ngroups; A.row_index = row_index;
  A.stat_sums = stat_sums; A.gram_sums = gram_sums;
  const int gx = (int)((cols + 255) / 256), gy = (int)((rows + 15) / 16);
  const unsigned blocks = (unsigned)(gx * gy + A.blocks_a + (sums ? HB_GRAM_BLOCKS : 0));
  hipLaunchKernelGGL(k_relu_bwd_tiles4_heads_dw, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, hh, rows, cols, ld, scratch, act, gx, gy, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_bias_finalize_ex(int device, int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles,
                                    float *const *out, void *stream) {
  NEED(nseg >= 1 && nseg <= BIAS_MAX_SEGS && partials && cols && tiles && out, "nseg / partials / cols / tiles / out");
  HIP_TRY(hipSetDevice(device));
  BiasSegs S{};
  S.n = nseg;
  int64_t maxc = 0;
  for (int i = 0; i < nseg; i++) {
    NEED(partials[i] && out[i] && cols[i] > 0 && tiles[i] > 0, "segment");
    S.tiles[i] = tiles[i]; S.partials[i] = partials[i]; S.cols[i] = cols[i]; S.db[i] = out[i];
    maxc = cols[i] > maxc ? cols[i] : maxc;
  }
  hipLaunchKernelGGL(k_bias_finalize, dim3((unsigned)((maxc + 63) / 64), (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, S);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_bias_finalize_rows(int device, int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles,
                                      float *const *out, int first_row_seg, const int32_t *row_index, void *stream) {
  NEED(nseg >= 1 && nseg <= BIAS_MAX_SEGS && partials && cols && tiles && out, "nseg / partials / cols / tiles / out");
  NEED(first_row_seg >= 0 && first_row_seg <= nseg && (row_index || first_row_seg == nseg), "first_row_seg / row_index");
  HIP_TRY(hipSetDevice(device));
  BiasSegs S{};
  S.n = nseg;
  int64_t maxc = 0;
  for (int i = 0; i < nseg; i++) {
    NEED(partials[i] && out[i] && cols[i] > 0 && tiles[i] > 0, "segment");
    S.tiles[i] = tiles[i]; S.partials[i] = partials[i]; S.cols[i] = cols[i]; S.db[i] = out[i];
    maxc = cols[i] > maxc ? cols[i] : maxc;
  }
  S.row_index = (first_row_seg < nseg) ? row_index : nullptr;
  S.first_row_seg = first_row_seg;
  hipLaunchKernelGGL(k_bias_finalize, dim3((unsigned)((maxc + 63) / 64), (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, S);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

#ifdef FAIR_TIMING
extern "C" int brl_fair_set_dbg(void *p) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fair_dbg), &p, sizeof(p)));
  return BRL_OK;
}
#endif

extern "C" int brl_fair_chain(int device, const brl_fair_net *net, const float *x0, const uint8_t *mask, const int32_t *action,
                              const float *old_value, const float *old_log_prob, const float *gae, const float *targets,
                              int64_t batch, float clip_eps, float vf_coef, float ent_coef, int masked, int value_clipping,
                              int reward_scaling, int act, const brl_fair_work *work, void *stream) {
  NEED(net && work && x0 && mask && action && old_value && old_log_prob && gae && targets, "NULL input");
  NEED(batch > 0 && batch % fair::R == 0, "batch (a multiple of 16)");
  NEED(act == 0 || act == 1, "act");
  fair::Args A{};
  for (int l = 0; l < 11; l++) {
    NEED(net->w[l] && net->b[l], "net: NULL layer");
    NEED((((uintptr_t)net->w[l]) & 15) == 0 && (((uintptr_t)net->b[l]) & 15) == 0, "net: 16-byte alignment");
    A.net.w[l] = net->w[l]; A.net.b[l] = net->b[l];
  }
  NEED(net->head_w && net->head_b && (((uintptr_t)net->head_w) & 15) == 0, "net: heads");
  A.net.wh = net->head_w; A.net.bh = net->head_b;
  NEED(work->inp && work->dzs && work->gates && work->cat6 && work->x4 && work->dz0 && work->dz6 && work->dheads && work->tiles &&
       work->partials, "work: NULL array");
  const void *al[] = {x0, work->inp, work->dzs, work->gates, work->cat6, work->x4, work->dz0, work->dz6, work->tiles};
  for (const void *q : al) NEED((((uintptr_t)q) & 15) == 0, "16-byte alignment");
  A.o = fair::Bufs{work->inp, work->dzs, work->gates, work->cat6, work->x4, work->dz0, work->dz6, work->dheads, work->tiles,
                   work->partials, work->gram_partials};
  A.x0 = x0;
  A.P = PpoArgs{nullptr, 0, nullptr, mask, action, old_value, old_log_prob, gae, targets, batch, clip_eps, vf_coef,
                ent_coef, masked, value_clipping, nullptr, nullptr, nullptr, nullptr, 1, BRL_NUM_ACTIONS, 1};
  A.act = act;
  A.reward_scaling = reward_scaling;
  HIP_TRY(hipSetDevice(device));
  A.nrows = batch;
  hipLaunchKernelGGL(fair::k_fair_chain<true>, dim3((unsigned)(batch / fair::R)), dim3(fair::NW * 64), 0, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_fair_forward(int device, const brl_fair_net *net, const float *x, int64_t rows, int act, float *logits, float *value,
                                void *stream) {
  NEED(net && x && logits && value, "NULL array");
  NEED(rows >= 0 && rows < (1ll << 31), "rows");
  NEED(act == 0 || act == 1, "act");
  if (rows == 0) return BRL_OK;
  fair::Args A{};
  for (int l = 0; l < 11; l++) {
    NEED(net->w[l] && net->b[l], "net: NULL layer");
    NEED((((uintptr_t)net->w[l]) & 15) == 0 && (((uintptr_t)net->b[l]) & 15) == 0, "net: 16-byte alignment");
    A.net.w[l] = net->w[l]; A.net.b[l] = net->b[l];
  }
  NEED(net->head_w && net->head_b && (((uintptr_t)net->head_w) & 15) == 0 && (((uintptr_t)x) & 15) == 0, "net: heads / x: 16-byte alignment");
  A.net.wh = net->head_w; A.net.bh = net->head_b;
  A.x0 = x;
  A.act = act;
  A.nrows = rows;
  A.logits_out = logits;
  A.value_out = value;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(fair::k_fair_chain<false>, dim3((unsigned)((rows + fair::R - 1) / fair::R)), dim3(fair::NW * 64), 0,
                     (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_heads_loss_split(int device, const float *h, int64_t ldh, const float *head_w, const float *head_b,
                                        int64_t hidden, const uint8_t *mask, const int32_t *action, const float *old_value,
                                        const float *old_log_prob, const float *gae, const float *targets, int64_t batch,
                                        float clip_eps, float vf_coef, float ent_coef, int masked, int value_clipping,
                                        int reward_scaling, float *heads_out, float *dheads, float *partials, float *gram_partials,
                                        float *head_parts, int ksplit, void *stream) {
  NEED(batch > 0 && hidden > 0 && hidden % 16 == 0 && ldh >= hidden && ldh % 4 == 0, "batch / hidden (a multiple of 16) / ldh");
  NEED(h && head_w && head_b && mask && action && old_value && old_log_prob && gae && targets, "NULL input array");
  NEED(dheads && partials && head_parts, "NULL output array");
  NEED(ksplit >= 1 && ksplit <= 8, "ksplit (1..8)");
  HIP_TRY(hipSetDevice(device));
  constexpr int64_t HS = BRL_NUM_ACTIONS + 1;
  HeadsProductArgs G{};
  G.h = h; G.ldh = ldh; G.Wh = head_w; G.H = (int)hidden; G.B = batch; G.ksplit = ksplit; G.parts = head_parts; G.part_stride = batch * HS;
  hipLaunchKernelGGL(k_heads_product, dim3((unsigned)((batch + 15) / 16), (unsigned)ksplit), dim3(HD_WAVES * 64), 0, (hipStream_t)stream, G);
  HeadsLossArgs A{};
  A.h = h; A.ldh = ldh; A.Wh = head_w; A.bh = head_b; A.H = (int)hidden;
  A.P = PpoArgs{nullptr, HS, nullptr, mask, action, old_value, old_log_prob, gae, targets, batch, clip_eps, vf_coef,
                ent_coef, masked, value_clipping, dheads, dheads + BRL_NUM_ACTIONS, partials, nullptr, HS, HS, HS};
  A.heads_out = heads_out; A.gram_partials = gram_partials; A.reward_scaling = reward_scaling;
  A.parts = head_parts; A.nparts = ksplit; A.part_stride = batch * HS;
  hipLaunchKernelGGL(k_heads_loss, dim3((unsigned)((batch + HD_ROWS - 1) / HD_ROWS)), dim3(HD_WAVES * 64), 0, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_heads_bwd(int device, const float *dheads, const float *h, int64_t ldh, const float *head_w, int64_t batch,
                                 int64_t hidden, int act, int nsplit, float *dw_partials, float *db_partials, float *dh,
                                 float *tile_sums, const float *loss_partials, const float *gram_partials, int64_t ngroups,
                                 const int32_t *row_index, float *stat_sums, float *gram_sums, void *stream) {
  NEED(batch > 0 && hidden > 0 && hidden % 256 == 0 && ldh >= hidden && ldh % 4 == 0, "batch / hidden (a multiple of 256) / ldh");
  NEED(dheads && h && head_w && dh && tile_sums && (!dw_partials == !db_partials), "NULL array");
  const bool with_dw = dw_partials != nullptr;   // NULL: the weight-gradient role is launched elsewhere (brl_act_bwd_colsum_heads_dw)
  NEED(act == 0 || act == 1, "act (0 ReLU, 1 tanh)");
  NEED(nsplit >= 1 && (batch + nsplit - 1) / nsplit <= 64, "nsplit: at most 64 rows per split");
  HIP_TRY(hipSetDevice(device));
  HeadsBwdArgs A{};
  A.dheads = dheads; A.h = h; A.ldh = ldh; A.Wh = head_w; A.B = batch; A.H = (int)hidden; A.act = act; A.nsplit = nsplit;
  A.rows_per_split = (int)((batch + nsplit - 1) / nsplit);
  A.dWh_partials = dw_partials; A.dbh_partials = db_partials; A.dh = dh; A.tile_sums = tile_sums;
  A.blocks_a = (int)(hidden / HB_JT) * nsplit;
  const bool sums = with_dw && gram_sums != nullptr;
  NEED(!sums || (loss_partials && gram_partials && ngroups > 0 && row_index && stat_sums), "statistics sums: partials / ngroups / row_index / stat_sums");
  A.loss_partials = loss_partials; A.gram_partials = gram_partials; A.ngroups = (int)ngroups; A.row_index = row_index;
  A.stat_sums = stat_sums; A.gram_sums = gram_sums;
  const int64_t blocks_b = (hidden / 256) * ((batch + HB_ROWS - 1) / HB_ROWS);
  // two launches (independent: back to back on the stream): the weight-gradient partials and the activation gradient
  hipLaunchKernelGGL(k_heads_bwd_dh, dim3((unsigned)blocks_b), dim3(256), 0, (hipStream_t)stream, A);
  if (with_dw)
    hipLaunchKernelGGL(k_heads_bwd_dw, dim3((unsigned)(A.blocks_a + (sums ? HB_GRAM_BLOCKS : 0))), dim3(256), 0, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_stats_gram(int device, const float *partials, int64_t npartials, int64_t batch, const float *gram_partials,
                                  int64_t ngram, float vf_coef, float ent_coef, float illegal_coef, float *out_rows,
                                  const int32_t *row_index, float *vec_out, void *stream) {
  NEED(partials && out_rows && npartials > 0 && batch > 0, "partials / out_rows / npartials / batch");
  NEED(gram_partials && ngram > 0, "gram_partials / ngram");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_ppo_stats2, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, npartials, batch, gram_partials, ngram,
                     vf_coef, ent_coef, illegal_coef, out_rows, row_index, vec_out);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_illegal_grad(int device, const float *heads, const uint8_t *mask, const float *vec, float illegal_coef,
                                    int64_t batch, float *dheads, void *stream) {
  NEED(heads && mask && vec && dheads && batch > 0, "heads / mask / vec / dheads / batch");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_illegal_grad, dim3(thread_grid(batch, 4)), dim3(256), 0, (hipStream_t)stream, heads, mask, vec, illegal_coef,
                     batch, dheads);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_stats_rows(int device, const float *stat_sums, const float *gram_sums, int64_t rows, int64_t batch, float vf_coef,
                                  float ent_coef, float illegal_coef, float *out_rows, void *stream) {
  NEED(stat_sums && gram_sums && out_rows && rows > 0 && batch > 0, "stat_sums / gram_sums / out_rows / rows / batch");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_ppo_stats2, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, stat_sums, (int64_t)1, batch, gram_sums,
                     (int64_t)1, vf_coef, ent_coef, illegal_coef, out_rows, (const int32_t *)nullptr, (float *)nullptr);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

// clip_by_global_norm + Adam of the whole flat buffers, single rank: the norm launch (which also finishes the sums of partials,
// see k_adam_norm_fin) and the sweep (+ the NEXT minibatch's gather as extra workgroups)
extern "C" int brl_adam_clip_fin_gather(int device, float *p, float *g, float *m, float *v, int64_t n, float *step, float lr,
                                        const float *lr_dev, float beta1, float beta2, float eps, float max_norm, float *scratch,
                                        int64_t scratch_len, int32_t *mb_index, float *norm_out, const void *gather_args, int64_t mbs,
                                        int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles,
                                        float *const *out, void *stream) {
  NEED(nseg >= 1 && nseg <= BIAS_MAX_SEGS && partials && cols && tiles && out, "nseg / partials / cols / tiles / out");
  NEED(p && g && m && v && step && scratch && n > 0 && n % 4 == 0, "p / g / m / v / step / scratch / n (a multiple of 4)");
  NEED(!gather_args || (mb_index && mbs > 0), "gather_args needs mb_index and the minibatch size");
  BiasSegs S{};
  S.n = nseg;
  int64_t covered = 0;
  const float *lo = g + n;
  for (int i = 0; i < nseg; i++) {
    NEED(partials[i] && out[i] && cols[i] > 0 && tiles[i] > 0, "segment");
    NEED(out[i] >= g && out[i] + cols[i] <= g + n, "segment outputs must lie inside the gradient buffer");
    S.tiles[i] = tiles[i]; S.partials[i] = partials[i]; S.cols[i] = cols[i]; S.db[i] = out[i];
    covered += cols[i];
    lo = (out[i] < lo) ? out[i] : lo;
  }
  const int64_t tail_lo = lo - g;
  // the segments must be exactly the tail of the buffer (up to its zero padding): everything in front is square-summed as it is
  NEED(tail_lo % 4 == 0 && covered <= n - tail_lo && n - tail_lo - covered < 4, "the finalised segments must tile the end of the gradient buffer");
  FinBlocks FB{};
  for (int i = 0; i < nseg; i++) FB.off[i + 1] = FB.off[i] + (int)((cols[i] + 63) / 64);
  NEED(scratch_len >= ADAM_BLOCKS + (int64_t)FB.off[nseg], "scratch too small for the finalize blocks' partials");
  HIP_TRY(hipSetDevice(device));
  const int npartials = ADAM_BLOCKS + FB.off[nseg];
  hipLaunchKernelGGL(k_adam_norm_fin, dim3((unsigned)npartials), dim3(ADAM_THREADS), 0, (hipStream_t)stream, g, n, 1.0f, scratch, step,
                     mb_index, S, tail_lo, FB);
  const unsigned extra = gather_args ? (unsigned)((mbs + 1) / 2) : 0u;
  AdamRange R{};
  R.p = p; R.g = g; R.m = m; R.v = v; R.lo4 = 0; R.hi4 = n >> 2; R.partials = scratch; R.npartials = npartials; R.step = step;
  R.lr_dev = lr_dev; R.lr = lr; R.b1 = beta1; R.b2 = beta2; R.eps = eps; R.max_norm = max_norm; R.gscale = 1.0f; R.norm_out = norm_out;
  hipLaunchKernelGGL(k_adam_apply, dim3((unsigned)ADAM_BLOCKS + extra), dim3(ADAM_THREADS), 0, (hipStream_t)stream, R, ADAM_BLOCKS,
                     (const GatherArgs *)gather_args);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

// ---- the multi-rank step's clip + Adam: on the rank slices of a bucketed flat buffer (ppo_update.hpp: ShardGeom) ------------
static int shard_geom(const brl_shard_geom *geom, int rank_lo, int rank_hi, ShardGeom *out) {
  NEED(geom && geom->nbuckets >= 1 && geom->nbuckets <= SHARD_MAX_BUCKETS && geom->world >= 1 && geom->nsub >= 1, "geometry");
  NEED(rank_lo >= 0 && rank_lo < rank_hi && rank_hi <= geom->world, "rank range");
  ShardGeom G{};
  G.nb = geom->nbuckets; G.world = geom->world; G.J = geom->nsub;
  for (int b = 0; b < G.nb; b++) {
    NEED(geom->off[b] >= 0 && geom->off[b] % 4 == 0 && geom->len[b] > 0 && geom->len[b] % 4 == 0, "bucket offsets / slice lengths multiples of 4 floats");
    G.off4[b] = geom->off[b] >> 2; G.len4[b] = geom->len[b] >> 2;
    G.blk[b + 1] = G.blk[b] + (int)((G.len4[b] + SHARD_PER4 - 1) / SHARD_PER4);
  }
  *out = G;
  return BRL_OK;
}

extern "C" int brl_adam_shard_norm(int device, const float *g, const brl_shard_geom *geom, int rank_lo, int rank_hi, float grad_scale,
                                   float *partials, float *step, int32_t *mb_index, void *stream) {
  NEED(g && partials && step, "g / partials / step");
  ShardGeom G;
  if (int rc = shard_geom(geom, rank_lo, rank_hi, &G)) return rc;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_shard_norm, dim3((unsigned)((rank_hi - rank_lo) * G.nb * G.J)), dim3(ADAM_THREADS), 0, (hipStream_t)stream, g, G,
                     rank_lo, grad_scale, partials, step, mb_index);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_adam_shard_apply(int device, float *p, const float *g, float *m, float *v, const brl_shard_geom *geom, int rank_lo,
                                    int rank_hi, const float *partials, const float *step, float lr, const float *lr_dev, float beta1,
                                    float beta2, float eps, float max_norm, float grad_scale, float *norm_out, const void *gather_args,
                                    int64_t mbs, void *stream) {
  NEED(p && g && m && v && partials && step, "p / g / m / v / partials / step");
  NEED(!gather_args || mbs > 0, "gather_args needs the minibatch size");
  ShardGeom G;
  if (int rc = shard_geom(geom, rank_lo, rank_hi, &G)) return rc;
  HIP_TRY(hipSetDevice(device));
  AdamRange R{};
  R.p = p; R.g = g; R.m = m; R.v = v; R.partials = partials; R.npartials = G.world * G.nb * G.J; R.step = step;
  R.lr_dev = lr_dev; R.lr = lr; R.b1 = beta1; R.b2 = beta2; R.eps = eps; R.max_norm = max_norm; R.gscale = grad_scale; R.norm_out = norm_out;
  const int sweep = (rank_hi - rank_lo) * G.blk[G.nb];
  const unsigned extra = gather_args ? (unsigned)((mbs + 1) / 2) : 0u;
  hipLaunchKernelGGL(k_shard_apply, dim3((unsigned)sweep + extra), dim3(ADAM_THREADS), 0, (hipStream_t)stream, R, G, rank_lo, sweep,
                     (const GatherArgs *)gather_args);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_ppo_stats(int device, const float *partials, int64_t batch, const float *gram, float vf_coef,
                             float ent_coef, float *out, void *stream) {
  NEED(partials && out && batch > 0, "partials / out / batch");
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_ppo_stats, dim3(1), dim3(512), 0, (hipStream_t)stream, partials, (int64_t)thread_grid(batch, 4), batch,
                     gram, vf_coef, ent_coef, out, (const int32_t *)nullptr);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

