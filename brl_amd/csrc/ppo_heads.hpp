// ppo_heads.hpp — the 39-column head of the actor-critic (38 logits + 1 value: src/models.py:30-33) inside one PPO minibatch
// step (src/update.py:90-167).  Its three products are ~80 MFLOP each — nothing for the matrix cores of a 256-CU chip — yet
// as library GEMMs (N = 39 or K = 39) they cost 9.5-13 us apiece plus the launches around them (loss 5 us, Gram matrix 5.6 us,
// head column sums 5.3 us, top layer's ReLU backward 6.2 us: profiles/r02/r02w_policy_path_kernel_stats.txt).  Here:
//   k_heads_product  h W_h^T (16 samples per workgroup, the K = 1024 sum split over workgroups AND over their 8 waves of
//                 v_mfma_f32_16x16x4_f32; partial sums per K range);
//   k_heads_loss  heads = b_h + the parts -> `_loss_fn` of 4 samples per workgroup (ppo_loss_sample: one wave per sample) ->
//                 d(loss)/d(heads), the statistics partials and the workgroup's 38 x 38 Gram-matrix partial of the illegal-action
//                 probabilities;
//   k_heads_bwd   role A: dW_h = d(heads)^T h and db_h = column sums of d(heads), per batch split (deterministic partials,
//                 summed by k_bias_finalize);  role B: dh = d(heads) W_h, times the top layer's activation derivative, with
//                 that layer's bias-gradient column sums per 16-row tile;
//   k_ppo_stats2  the logged statistics + the illegal-action spectral norm from the Gram partials (4 squarings by the whole block
//                 + 16 power-iteration steps by one wave = G^256 v, like k_ppo_stats' 8 squarings, in ~3 us instead of 62).
// Included by brl_ppo.hip after ppo_update.hpp.
#pragma once

constexpr int HD_ROWS = 4;     // samples per workgroup of k_heads_loss (of the 16 rows of an MFMA tile; see the kernel)
constexpr int HD_WAVES = 8;    // waves per workgroup = K splits; waves 0..3 finish one sample each
constexpr int HD_MAX_PARTS = 8;    // partial products k_heads_loss adds (k_heads_product: <= 8 K ranges)
#include "heads_dw_role.hpp"   // HD_NOUT, HD_GRAM, HB_JT, hd_f32x4, HeadsBwdArgs, heads_bwd_dw_block


struct HeadsLossArgs {
  const float *h;       // [B, H] the last hidden layer's output
  int64_t ldh;
  const float *Wh;      // [39, H]: actor rows, then the critic row
  const float *bh;      // [39]
  int H;                // % 16 == 0
  PpoArgs P;            // .logits / .value unused (computed here); outputs in the merged [B, 39] layout
  float *heads_out;     // [B, 39] or NULL
  float *gram_partials; // [ceil(B / 4)][1444] or NULL
  int reward_scaling;   // src/update.py:31-44: advantages normalised over the minibatch (jnp std: ddof = 0)
  // the heads product, formed by k_heads_product as `nparts` partial sums over K ranges:
  // heads[b][n] = bh[n] + parts[0][b][n] + parts[1][b][n] + ..  (in that order); h / Wh are not read here
  const float *parts;   // [nparts][part_stride], row b at b * 39
  int nparts;           // <= HD_MAX_PARTS
  int64_t part_stride;
};

__global__ __launch_bounds__(HD_WAVES * 64) void k_heads_loss(HeadsLossArgs A) {
  __shared__ float illp_s[HD_ROWS][BRL_NUM_ACTIONS + 2];
  __shared__ float part_s[HD_ROWS][8];
  __shared__ float rs_red[HD_WAVES], rs_stat[2];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t B = A.P.B, row0 = (int64_t)blockIdx.x * HD_ROWS;

  // ---- optional: mean / std of the minibatch's advantages (every workgroup computes them itself, in the same fixed
  // order: identical on all of them, no cross-workgroup hand-off)
  float adv_mean = 0.0f, adv_inv = 1.0f;
  if (A.reward_scaling) {
    float s = 0.0f;
    for (int64_t i = tid; i < B; i += HD_WAVES * 64) s += A.P.gae[i];
    s = wave_sum_f(s);
    if (lane == 0) rs_red[w] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int k = 0; k < HD_WAVES; k++) t += rs_red[k];
      rs_stat[0] = t / (float)B;
    }
    __syncthreads();
    adv_mean = rs_stat[0];
    float q = 0.0f;
    for (int64_t i = tid; i < B; i += HD_WAVES * 64) {
      const float d = A.P.gae[i] - adv_mean;
      q += d * d;
    }
    q = wave_sum_f(q);
    __syncthreads();
    if (lane == 0) rs_red[w] = q;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int k = 0; k < HD_WAVES; k++) t += rs_red[k];
      rs_stat[1] = 1.0f / (sqrtf(t / (float)B) + 1e-8f);
    }
    __syncthreads();
    adv_inv = rs_stat[1];
  }

  // everything the loss of this wave's sample reads from global memory, issued at once
  const int64_t my_b = row0 + ((w < HD_ROWS) ? w : 0);
  const bool my_valid = w < HD_ROWS && my_b < B;
  const PpoSampleIn my_in = ppo_sample_load(A.P, my_b, my_valid, lane);
  const float my_bias = (lane < HD_NOUT) ? A.bh[lane] : 0.0f;
  float my_part[HD_MAX_PARTS];   // the heads product comes in as partial sums over K ranges (k_heads_product)
#pragma unroll
  for (int p = 0; p < HD_MAX_PARTS; p++)
    my_part[p] = (my_valid && lane < HD_NOUT && p < A.nparts) ? A.parts[(int64_t)p * A.part_stride + my_b * HD_NOUT + lane] : 0.0f;

  // ---- `_loss_fn` (src/update.py:90-167): wave w < 4 takes sample w; lane a = head a
  if (w < HD_ROWS) {
    const int sl = w;
    const int64_t b = my_b;
    const bool valid = my_valid;
    float hv = 0.0f;
    if (lane < HD_NOUT) {
      hv = my_bias;
#pragma unroll
      for (int p = 0; p < HD_MAX_PARTS; p++) hv += my_part[p];   // fixed order (absent parts are exact zeros)
      if (valid && A.heads_out) A.heads_out[b * HD_NOUT + lane] = hv;
    }
    const float v = __shfl(hv, BRL_NUM_ACTIONS, 64);
    const float g = A.reward_scaling ? (my_in.gae - adv_mean) * adv_inv : my_in.gae;
    float st[5], ill;
    ppo_loss_sample(A.P, my_in, b, valid, lane, hv, v, g, st, ill);
    if (lane < BRL_NUM_ACTIONS) illp_s[sl][lane] = valid ? ill : 0.0f;
    if (lane == 0) {   // (st is wave-uniform)
#pragma unroll
      for (int k = 0; k < 8; k++) part_s[sl][k] = (k < 5) ? st[k] : 0.0f;
    }
  }
  __syncthreads();
  if (tid < 8) {   // per-workgroup partial sums over its 16 samples, in order (deterministic statistics)
    float s = 0.0f;
    for (int k = 0; k < HD_ROWS; k++) s += part_s[k][tid];
    A.P.partials[(int64_t)blockIdx.x * 8 + tid] = s;
  }
  if (A.gram_partials != nullptr) {   // G_wg[i][j] = sum over the workgroup's samples of illp[i] illp[j]
    for (int e = tid; e < HD_GRAM; e += HD_WAVES * 64) {
      const int i = e / BRL_NUM_ACTIONS, j = e - i * BRL_NUM_ACTIONS;
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < HD_ROWS; k++) s += illp_s[k][i] * illp_s[k][j];
      A.gram_partials[(int64_t)blockIdx.x * HD_GRAM + e] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward of the head.  act: 0 = ReLU (derivative 1 where the layer's output > 0), 1 = tanh (1 - output^2)
constexpr int HB_NG = 10;       // role A: heads per thread (4 groups of 10 >= 39)
constexpr int HB_ROWS = 16;     // role B: rows per workgroup (= the tile of the bias-gradient column sums)

// The heads product of k_heads_loss as a launch of its own, split over K ACROSS workgroups: workgroup (rt, ks) multiplies the 16
// samples rt * 16 .. with K groups [ks * gps, (ks + 1) * gps) of W_h (its 8 waves take every 8th group of that range) and writes
// parts[ks][b][n] — 64 KB of operands per workgroup at H = 1024 / 4 splits instead of the 224-256 KB a whole-K workgroup pulls
// through its CU (which was 9.3 k of k_heads_loss's 24.4 k cycles).  The loss launch adds bias + parts in order.
struct HeadsProductArgs {
  const float *h;
  int64_t ldh;
  const float *Wh;
  int H;
  int64_t B;
  int ksplit;
  float *parts;          // [ksplit][part_stride]
  int64_t part_stride;   // >= B * 39
};

__global__ __launch_bounds__(HD_WAVES * 64) void k_heads_product(HeadsProductArgs A) {
  __shared__ float red[HD_WAVES][3][4][64];   // 24 KB
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int ks = (int)blockIdx.y;
  const int ngroups = A.H / 16, gps = (ngroups + A.ksplit - 1) / A.ksplit;
  const int g_lo = ks * gps, g_hi = (g_lo + gps < ngroups) ? g_lo + gps : ngroups;
  const int r = lane & 15, kq = lane >> 4;
  const int64_t arow = (row0 + r < A.B) ? row0 + r : A.B - 1;
  const float *ap = A.h + arow * A.ldh + 4 * kq;
  const float *bp[3];
  bool bok[3];
#pragma unroll
  for (int nb = 0; nb < 3; nb++) {
    const int n = 16 * nb + r;
    bok[nb] = n < HD_NOUT;
    bp[nb] = A.Wh + (int64_t)(bok[nb] ? n : 0) * A.H + 4 * kq;
  }
  hd_f32x4 acc[3];
#pragma unroll
  for (int nb = 0; nb < 3; nb++) acc[nb] = hd_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int GP = 4;   // groups of a wave in flight at once
  for (int g0 = g_lo + w; g0 < g_hi; g0 += GP * HD_WAVES) {
    hd_f32x4 av[GP], bv[GP][3];
#pragma unroll
    for (int u = 0; u < GP; u++) {
      const int g = g0 + u * HD_WAVES;
      const int gc = (g < g_hi) ? g : g0;
      av[u] = *reinterpret_cast<const hd_f32x4 *>(ap + 16 * gc);
#pragma unroll
      for (int nb = 0; nb < 3; nb++) bv[u][nb] = *reinterpret_cast<const hd_f32x4 *>(bp[nb] + 16 * gc);
    }
#pragma unroll
    for (int u = 0; u < GP; u++) {
      if (g0 + u * HD_WAVES >= g_hi) break;
#pragma unroll
      for (int nb = 0; nb < 3; nb++) {
#pragma unroll
        for (int s_ = 0; s_ < 4; s_++)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s_], bok[nb] ? bv[u][nb][s_] : 0.0f, acc[nb], 0, 0, 0);
      }
    }
  }
  // accumulator register q of lane (c, rq): D[sample 4 rq + q][head 16 nb + c]
#pragma unroll
  for (int nb = 0; nb < 3; nb++)
#pragma unroll
    for (int q = 0; q < 4; q++) red[w][nb][q][lane] = acc[nb][q];
  __syncthreads();
  float *out = A.parts + (int64_t)ks * A.part_stride;
  for (int e = tid; e < 16 * HD_NOUT; e += HD_WAVES * 64) {
    const int row = e / HD_NOUT, col = e - row * HD_NOUT;
    const int nb = col >> 4, c = col & 15, q = row & 3, rq = row >> 2;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < HD_WAVES; k++) v += red[k][nb][q][16 * rq + c];   // fixed order
    if (row0 + row < A.B) out[(row0 + row) * HD_NOUT + col] = v;
  }
}

// (HeadsBwdArgs and heads_bwd_dw_block live in heads_dw_role.hpp: the GEMM translation unit lets that role ride in its launches)

__global__ __launch_bounds__(256) void k_heads_bwd_dw(HeadsBwdArgs A) { heads_bwd_dw_block(A, (int)blockIdx.x); }

// The activation-derivative pass of a hidden layer (k_relu_bwd_tiles4: on the backward chain) with the head's weight-gradient
// partials (k_heads_bwd_dw: NOT on the chain — nothing needs them before the sums at the end of the step) as extra workgroups of
// the same launch: blocks [0, gx * gy) are the tiles of the former, the rest the blocks of the latter.
__global__ __launch_bounds__(256) void k_relu_bwd_tiles4_heads_dw(float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld,
                                                                   float *partials, int act, int gx, int gy, HeadsBwdArgs A) {
  const int b = (int)blockIdx.x;
  if (b < gx * gy) relu_bwd_tiles4_block(dh, h, rows, cols, ld, partials, act, b % gx, b / gx);
  else heads_bwd_dw_block(A, b - gx * gy);
}

__global__ __launch_bounds__(256) void k_heads_bwd_dh(HeadsBwdArgs A) {
  __shared__ __attribute__((aligned(16))) float dh_s[HB_ROWS][HD_NOUT + 1];
  __shared__ float4 cs_s[4][64];
  const int tid = (int)threadIdx.x;
  // ---- role B: dh[b][j] = act'(h[b][j]) * sum_n d(heads)[b][n] W_h[n][j]; workgroup = 16 rows x 256 columns,
  // thread = (4 columns c4, rows rg, rg + 4, rg + 8, rg + 12)
  const int bb = (int)blockIdx.x;
  const int ct = bb % (A.H / 256), rt = bb / (A.H / 256);
  const int64_t r0 = (int64_t)rt * HB_ROWS;
  for (int e = tid; e < HB_ROWS * HD_NOUT; e += 256) {
    const int rr = e / HD_NOUT, c = e - rr * HD_NOUT;
    dh_s[rr][c] = (r0 + rr < A.B) ? A.dheads[(r0 + rr) * HD_NOUT + c] : 0.0f;
  }
  __syncthreads();
  const int c4 = tid & 63, rg = tid >> 6;
  const int col = ct * 256 + 4 * c4;
  float4 acc[4];
#pragma unroll
  for (int u = 0; u < 4; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 hv[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {   // the gate's operand, issued before the product
    const int64_t rw = r0 + rg + 4 * u;
    hv[u] = *reinterpret_cast<const float4 *>(A.h + ((rw < A.B) ? rw : A.B - 1) * A.ldh + col);
  }
  const float *wp = A.Wh + col;
  static_assert(HD_NOUT == 39, "three batches of 13 head rows");
#pragma unroll 1
  for (int n0 = 0; n0 < HD_NOUT; n0 += 13) {   // 13 rows of W_h in flight (L2-resident, shared by every workgroup)
    float4 wv[13];
#pragma unroll
    for (int k = 0; k < 13; k++) wv[k] = *reinterpret_cast<const float4 *>(wp + (int64_t)(n0 + k) * A.H);
#pragma unroll
    for (int k = 0; k < 13; k++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float d = dh_s[rg + 4 * u][n0 + k];
        acc[u].x += d * wv[k].x; acc[u].y += d * wv[k].y; acc[u].z += d * wv[k].z; acc[u].w += d * wv[k].w;
      }
    }
  }
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t rw = r0 + rg + 4 * u;
    float4 z;
    if (A.act == 0) {
      z = make_float4(hv[u].x > 0.f ? acc[u].x : 0.f, hv[u].y > 0.f ? acc[u].y : 0.f, hv[u].z > 0.f ? acc[u].z : 0.f,
                      hv[u].w > 0.f ? acc[u].w : 0.f);
    } else {
      z = make_float4(acc[u].x * (1.0f - hv[u].x * hv[u].x), acc[u].y * (1.0f - hv[u].y * hv[u].y),
                      acc[u].z * (1.0f - hv[u].z * hv[u].z), acc[u].w * (1.0f - hv[u].w * hv[u].w));
    }
    if (rw < A.B) {
      *reinterpret_cast<float4 *>(A.dh + rw * (int64_t)A.H + col) = z;
      cs.x += z.x; cs.y += z.y; cs.z += z.z; cs.w += z.w;
    }
  }
  cs_s[rg][c4] = cs;
  __syncthreads();
  if (rg == 0) {
    const float4 a = cs_s[0][c4], b = cs_s[1][c4], c = cs_s[2][c4], e = cs_s[3][c4];
    *reinterpret_cast<float4 *>(A.tile_sums + (int64_t)rt * A.H + col) =
        make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y), (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The logged statistics of one minibatch step from k_heads_loss's per-workgroup partials, one block of 1024 threads:
//   out[0] total  [1] value_loss  [2] loss_actor  [3] entropy  [4] approx_kl  [5] clipfrac  [6] illegal-action norm / 2  [7] 0
// The norm = largest singular value / 2 of the non-negative [B, 38] matrix P of illegal-action probabilities
// (`jnp.linalg.norm(..., ord=2) / 2`, src/update.py:138-141) = sqrt(top eigenvalue of G = P^T P) / 2 with G the sum of the
// workgroups' Gram partials: M = G / trace, squared 4 times by the whole block (M symmetric: (M^2)[i][j] = sum_k M[k][i] M[k][j]
// — lane-contiguous, conflict-free LDS reads — rescaled by ONE reciprocal of the trace per squaring), then 16 steps of
// v <- M^16 v / |.| by one wave (= G^256 v, the Perron vector to (l2 / l1)^256), Rayleigh quotient with the original G.
// v1 and the norm are also written to `vec_out` [40] (v1[0..37], sigma_1, 0) when given: the gradient of the norm needs them.
__global__ __launch_bounds__(1024) void k_ppo_stats2(const float *partials, int64_t nblk, int64_t batch, const float *gram_partials,
                                                    int64_t ngram, float vf_coef, float ent_coef, float ill_coef, float *out,
                                                    const int32_t *row_index, float *vec_out) {
  if (row_index != nullptr) out += 8 * (int64_t)(*row_index);
  // (brl_ppo_stats_rows: one block per row of per-update buffers — already reduced to one partial row each)
  out += 8 * (int64_t)blockIdx.x;
  partials += (int64_t)blockIdx.x * nblk * 8;
  gram_partials += (int64_t)blockIdx.x * ngram * (BRL_NUM_ACTIONS * BRL_NUM_ACTIONS);
  constexpr int D = BRL_NUM_ACTIONS, DD = D * D;
  __shared__ float g[DD], m[DD], t[DD], vec[64], st[8], tr_s;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NT = (int)blockDim.x;   // 1024 (one step, latency-bound) or 256 (brl_ppo_stats_rows: one block per row)
  const int w0 = (NT >= 1024) ? 8 : 3;   // waves w0..: one statistic each (wave 3 takes all five in a 256-thread block)
  for (int k = (wv >= w0) ? wv - w0 : 5; k < 5; k += (NT >= 1024) ? 5 : 1) {   // lane l adds rows l, l + 64, ... in order, then a fixed butterfly
    float s = 0.0f;
    for (int64_t i = lane; i < nblk; i += 64) s += partials[i * 8 + k];
    s = wave_sum_f(s);
    if (lane == 0) st[k] = s / (float)batch;
  }
  // G = sum of the Gram partials, in workgroup order.  One block sums 1444 x ngram floats: latency, not bandwidth — so
  // 16 loads are in flight per thread (2 threads' worth of entries per thread at 1024 threads)
  for (int e = tid; e < DD; e += NT) {
    float s = 0.0f;
    for (int64_t i = 0; i < ngram; i += 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; u++) v[u] = gram_partials[((i + u < ngram) ? i + u : i) * DD + e];
#pragma unroll
      for (int u = 0; u < 32; u++) s += (i + u < ngram) ? v[u] : 0.0f;
    }
    g[e] = s;
  }
  __syncthreads();
  auto trace_of = [&](const float *x) {   // wave 0, fixed butterfly -> 1 / max(trace, tiny) for everybody
    if (wv == 0) {
      const float s = wave_sum_f((lane < D) ? x[lane * D + lane] : 0.0f);
      if (lane == 0) tr_s = 1.0f / fmaxf(s, 1.17549435e-38f);
    }
    __syncthreads();
    return tr_s;
  };
  float inv = trace_of(g);
  for (int e = tid; e < DD; e += NT) m[e] = g[e] * inv;
  __syncthreads();
  for (int it = 0; it < 4; it++) {
    for (int e = tid; e < DD; e += NT) {
      const int i = e / D, j = e - i * D;
      float s = 0.0f;
#pragma unroll 2
      for (int k = 0; k < D; k++) s += m[k * D + i] * m[k * D + j];
      t[e] = s;
    }
    __syncthreads();
    inv = trace_of(t);
    for (int e = tid; e < DD; e += NT) m[e] = t[e] * inv;
    __syncthreads();
  }
  if (wv == 0) {   // v <- M v / |M v| sixteen times, M = (G / trace)^16; lane i owns v[i] and column i of M (= row i: M symmetric)
    float mc[D];
#pragma unroll
    for (int k = 0; k < D; k++) mc[k] = (lane < D) ? m[k * D + lane] : 0.0f;
    float v = (lane < D) ? 1.0f : 0.0f;
    for (int it = 0; it < 16; it++) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < D; k++)   // v[k] of lane k as a scalar: v_readlane, no LDS round trip
        s += mc[k] * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k));
      const float nrm = wave_sum_f(s * s);
      v = (nrm > 0.0f) ? s * (1.0f / sqrtf(nrm)) : v;
    }
    float gv = 0.0f;
#pragma unroll
    for (int k = 0; k < D; k++)
      gv += ((lane < D) ? g[k * D + lane] : 0.0f) * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k));
    const float num = wave_sum_f(v * gv), den = wave_sum_f(v * v);
    const float sigma = sqrtf(fmaxf(num / fmaxf(den, 1.17549435e-38f), 0.0f));
    if (vec_out != nullptr && lane < D + 2) vec_out[lane] = (lane < D) ? v * (1.0f / sqrtf(fmaxf(den, 1.17549435e-38f))) : ((lane == D) ? sigma : 0.0f);
    if (lane == 0) {
      out[6] = 0.5f * sigma;
      out[0] = st[1] + vf_coef * st[0] - ent_coef * st[2] + ill_coef * (0.5f * sigma);   // src/update.py:146-152
      out[1] = st[0]; out[2] = st[1]; out[3] = st[2]; out[4] = st[3]; out[5] = st[4];
      out[7] = 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// illegal_action_l2norm_coef != 0 (src/update.py:136-152): the loss gains coef * sigma_1(P) / 2 with P[b][a] = softmax(logits)[b][a]
// on the illegal actions.  d sigma_1 / dP = u1 v1^T (top singular pair), u1 = P v1 / sigma_1, so with q_b = (P v1)[b]:
//     d loss / d logit[b][j] += (coef / 2) (q_b / sigma_1) p[b][j] (illegal[b][j] v1[j] - q_b)
// One wave per sample, lane = action; v1 / sigma_1 from k_ppo_stats2's vec_out of THIS step; added to dheads in place.
__global__ __launch_bounds__(256) void k_illegal_grad(const float *heads, const uint8_t *mask, const float *vec, float coef, int64_t B,
                                                      float *dheads) {
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;
  const bool in = lane < BRL_NUM_ACTIONS;
  const float lg = in ? heads[b * HD_NOUT + lane] : 0.0f;
  const bool illegal = in && mask[b * BRL_NUM_ACTIONS + lane] == 0;
  const float v1 = in ? vec[lane] : 0.0f, sigma = vec[BRL_NUM_ACTIONS];
  const float mx = wave_max_f(in ? lg : -INFINITY);
  const float ex = in ? expf(lg - mx) : 0.0f;
  const float p = ex / wave_sum_f(ex);
  const float q = wave_sum_f(illegal ? p * v1 : 0.0f);
  if (in && sigma > 0.0f) dheads[b * HD_NOUT + lane] += (0.5f * coef) * (q / sigma) * p * ((illegal ? v1 : 0.0f) - q);
}
