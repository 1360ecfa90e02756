// ppo_update.hpp — the non-GEMM half of one PPO minibatch step (src/update.py:74-242) as a handful of launches:
// minibatch gather, activation backward + bias gradient, global-norm clip + Adam on flat buffers (whole, or the rank's slices
// under a process group).  brl_amd/update.py::FusedMinibatch strings them together with the GEMMs and captures the step in a
// hipGraph.  Included by brl_ppo.hip (C-ABI there).
#pragma once

// ---- minibatch gather: row perm[mb * B + b] of the flattened [T*N] trajectory -> static minibatch buffers, observation
// bytes -> float (`take(batch, permutation)` + one minibatch slice, src/update.py:193-206; G5: obs.astype(float32)).
// One 128-thread block per sample.  `mb_index` lives in device memory (advanced by k_adam_apply), so a captured graph walks
// through the epoch's minibatches by itself.
struct GatherArgs {
  const uint8_t *obs, *mask;
  const int32_t *action;
  const float *value, *log_prob, *adv, *tgt;
  const int64_t *perm;
  const int32_t *mb_index;
  int64_t B;
  float *x0;
  uint8_t *o_mask;
  int32_t *o_action;
  float *o_value, *o_log_prob, *o_adv, *o_tgt;
  int64_t nsteps;   // minibatches the bound permutation holds (a gather beyond the last one is skipped: k_adam_apply's blocks)
};

__device__ __forceinline__ void mb_gather_row(const GatherArgs &A, int64_t b, int t);

// the launch reads its arguments from DEVICE memory (written by k_mb_gather_bind once per update): the captured minibatch
// step starts with its own gather and stays valid when the next update brings another trajectory / permutation
__global__ __launch_bounds__(128) void k_mb_gather_dev(const GatherArgs *Ad) {
  const GatherArgs A = *Ad;
  mb_gather_row(A, blockIdx.x, (int)threadIdx.x);
}
__global__ void k_mb_gather_bind(GatherArgs A, GatherArgs *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = A;
}

__device__ __forceinline__ void mb_gather_row(const GatherArgs &A, int64_t b, int t) {
  const int64_t row = A.perm[(int64_t)(*A.mb_index) * A.B + b];
  if (t < 120) {  // 4 observation bytes -> 4 floats
    const uint32_t w = reinterpret_cast<const uint32_t *>(A.obs + row * BRL_OBS_SIZE)[t];
    reinterpret_cast<float4 *>(A.x0 + b * BRL_OBS_SIZE)[t] =
        make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24));
  }
  if (t < BRL_NUM_ACTIONS) A.o_mask[b * BRL_NUM_ACTIONS + t] = A.mask[row * BRL_NUM_ACTIONS + t];
  if (t >= 64 && t < 69) {  // the five 4-byte scalars of the sample: ONE load instruction (lane = column), not five with a wait each
    const int k = t - 64;
    const uint32_t *src = (k == 0) ? reinterpret_cast<const uint32_t *>(A.action)
                        : (k == 1) ? reinterpret_cast<const uint32_t *>(A.value)
                        : (k == 2) ? reinterpret_cast<const uint32_t *>(A.log_prob)
                        : (k == 3) ? reinterpret_cast<const uint32_t *>(A.adv) : reinterpret_cast<const uint32_t *>(A.tgt);
    uint32_t *dst = (k == 0) ? reinterpret_cast<uint32_t *>(A.o_action)
                  : (k == 1) ? reinterpret_cast<uint32_t *>(A.o_value)
                  : (k == 2) ? reinterpret_cast<uint32_t *>(A.o_log_prob)
                  : (k == 3) ? reinterpret_cast<uint32_t *>(A.o_adv) : reinterpret_cast<uint32_t *>(A.o_tgt);
    dst[b] = src[row];
  }
}

// ---- activation backward + bias gradient of one hidden layer (where the own GEMM's epilogue does not do it): dz = dh * act'(h)
// in place and the column sums of every 16-row tile; k_bias_finalize: db[c] = sum over the row tiles in index order (deterministic;
// no atomics, no cross-block hand-off inside a launch).  cols % 4 == 0: float4 accesses, a block covers 256 columns.
// act: 0 = ReLU (dz = dh where h > 0), 1 = tanh (dz = dh * (1 - h^2): src/models.py:16 `activation == "tanh"`)
__device__ __forceinline__ void relu_bwd_tiles4_block(float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld,
                                                      float *partials, int act, const int bx, const int by) {
  __shared__ float4 part[4][64];
  const int cg = (int)(threadIdx.x & 63u), rg = (int)(threadIdx.x >> 6);
  const int64_t col = ((int64_t)bx * 64 + cg) * 4, r0 = (int64_t)by * 16 + rg;
  const bool cv = col < cols;
  float4 d[4], hv[4];
  const int64_t cc = cv ? col : cols - 4;  // (cols % 4 == 0)
  // (unconditional loads from clamped addresses, the h == NULL case decided once: a guard per element makes the compiler
  //  branch around every load and wait for each — 8 memory round trips instead of one)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t r = r0 + 4 * k, rc = (r < rows) ? r : rows - 1;
    d[k] = *reinterpret_cast<const float4 *>(dh + rc * ld + cc);
  }
  if (h != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t r = r0 + 4 * k, rc = (r < rows) ? r : rows - 1;
      hv[k] = *reinterpret_cast<const float4 *>(h + rc * ld + cc);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) hv[k] = make_float4(1.f, 1.f, 1.f, 1.f);
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t r = r0 + 4 * k;
    const bool v = cv && r < rows;
    const float4 z = (act == 0 || h == nullptr)
        ? make_float4((v && hv[k].x > 0.f) ? d[k].x : 0.f, (v && hv[k].y > 0.f) ? d[k].y : 0.f,
                      (v && hv[k].z > 0.f) ? d[k].z : 0.f, (v && hv[k].w > 0.f) ? d[k].w : 0.f)
        : make_float4(v ? d[k].x * (1.f - hv[k].x * hv[k].x) : 0.f, v ? d[k].y * (1.f - hv[k].y * hv[k].y) : 0.f,
                      v ? d[k].z * (1.f - hv[k].z * hv[k].z) : 0.f, v ? d[k].w * (1.f - hv[k].w * hv[k].w) : 0.f);
    if (h != nullptr && cv && r < rows) *reinterpret_cast<float4 *>(dh + r * ld + col) = z;
    s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w;
  }
  part[rg][cg] = s;
  __syncthreads();
  if (rg == 0 && cv) {
    const float4 a = part[0][cg], b = part[1][cg], c = part[2][cg], e = part[3][cg];
    *reinterpret_cast<float4 *>(partials + (int64_t)by * cols + col) =
        make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y), (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
  }
}

__global__ __launch_bounds__(256) void k_relu_bwd_tiles4(float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld,
                                                          float *partials, int act = 0) {
  relu_bwd_tiles4_block(dh, h, rows, cols, ld, partials, act, (int)blockIdx.x, (int)blockIdx.y);
}

constexpr int BIAS_MAX_SEGS = 16;  // DeepMind_8: 8 hidden layers + the head; FAIR: 11 layers + the head + 2 statistics rows
struct BiasSegs {  // up to 16 sums finalised by one launch (blockIdx.y = segment)
  int n;
  int64_t tiles[BIAS_MAX_SEGS];
  const float *partials[BIAS_MAX_SEGS];
  int64_t cols[BIAS_MAX_SEGS];
  float *db[BIAS_MAX_SEGS];
  // segments >= first_row_seg are rows of a log: written at db[seg] + *row_index * cols[seg] (row_index: device memory; NULL: none)
  const int32_t *row_index;
  int first_row_seg;
};

// (64 columns per block, 4 threads per column: thread group g adds tiles g, g + 4, ... in order, then (g0 + g1) + (g2 + g3):
//  a fixed order, and 16 loads per thread instead of a 64-long chain — 17 -> ~5 us for five 1024-column layers)
// -> (threads of group 0) the finished sum of this thread's column, already stored; 0 for the others / beyond `cols`
__device__ __forceinline__ float bias_finalize_block(const BiasSegs &S, const int seg, const int bx, float (*part)[64]) {
  const int c = (int)(threadIdx.x & 63u), g = (int)(threadIdx.x >> 6);
  const int64_t col = (int64_t)bx * 64 + c, cols = S.cols[seg];
  const float *p = S.partials[seg];
  float s = 0.0f;
  if (col < cols) {   // (latency-bound: the thread's tiles are requested 16 at a time, then added in index order)
    const int64_t nt = S.tiles[seg];
    for (int64_t t0 = g; t0 < nt; t0 += 64) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = p[((t0 + 4 * u < nt) ? t0 + 4 * u : t0) * cols + col];
#pragma unroll
      for (int u = 0; u < 16; u++) s += (t0 + 4 * u < nt) ? v[u] : 0.0f;
    }
  }
  part[g][c] = s;
  __syncthreads();
  float v = 0.0f;
  if (g == 0 && col < cols) {
    v = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    float *db = S.db[seg];
    if (S.row_index != nullptr && seg >= S.first_row_seg) db += (int64_t)(*S.row_index) * cols;
    db[col] = v;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_bias_finalize(BiasSegs S) {
  __shared__ float part[4][64];
  (void)bias_finalize_block(S, (int)blockIdx.y, (int)blockIdx.x, part);
}

// ---- optax.chain(clip_by_global_norm(max_norm), adam(lr, eps)) (ppo.py:195-211) on flat fp32 buffers, two launches:
// k_adam_norm: per-block partial sums of g^2 (fixed order) and the step counter; k_adam_apply: every block re-adds the
// partials in the same order (deterministic, no cross-block hand-off), scales the gradient, updates m, v, p the way
// torch.optim.Adam does (bias corrections 1 - beta^t, denominator sqrt(v) / sqrt(bc2) + eps).
constexpr int ADAM_BLOCKS = 1024, ADAM_THREADS = 256;  // n is a multiple of 4 (the caller pads its flat buffers)

// The norm launch + k_bias_finalize in ONE launch (single rank: nothing sits between them).  The flat gradient buffer ends with the
// ranges k_bias_finalize produces (the head's weight gradient and every bias gradient, floats [tail_lo, n)): blocks
// [0, ADAM_BLOCKS) square-sum their chunk of g[0, tail_lo) (block 0 also advances the step / minibatch counters); block ADAM_BLOCKS + FB.off[seg] + bx finishes 64 columns of
// segment `seg` (the sums k_bias_finalize forms, in its order), stores them AND adds their squares to its own partial — one
// small launch (~5 us of latency chain) less per step.
struct FinBlocks { int off[BIAS_MAX_SEGS + 1]; };   // segment `seg` owns the finalize blocks [off[seg], off[seg + 1]): 64 columns each

__global__ __launch_bounds__(ADAM_THREADS) void k_adam_norm_fin(const float *g, int64_t n, float gscale, float *partials, float *step,
                                                                int32_t *mb_index, BiasSegs S, int64_t tail_lo, FinBlocks FB) {
  static_assert(ADAM_THREADS == 256, "the finalize blocks are 4 x 64 threads");
  __shared__ float red[ADAM_THREADS / 64];
  __shared__ float part[4][64];
  if ((int)blockIdx.x >= ADAM_BLOCKS) {
    const int e = (int)blockIdx.x - ADAM_BLOCKS;
    int seg = 0;
    while (seg + 1 < S.n && e >= FB.off[seg + 1]) seg++;
    const int bx = e - FB.off[seg];
    float v = bias_finalize_block(S, seg, bx, part) * gscale;
    v = wave_sum_f(v * v);   // (the stored columns live in wave 0; the other waves add zeros)
    if (threadIdx.x == 0) partials[blockIdx.x] = v;
    return;
  }
  const int64_t n4 = n >> 2, t4 = tail_lo >> 2;
  const int64_t chunk = (n4 + ADAM_BLOCKS - 1) / ADAM_BLOCKS;
  int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n4) ? lo + chunk : n4;
  hi = (hi < t4) ? hi : t4;
  float s = 0.0f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += ADAM_THREADS) {
    float4 x = reinterpret_cast<const float4 *>(g)[i];
    x.x *= gscale; x.y *= gscale; x.z *= gscale; x.w *= gscale;
    s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  s = wave_sum_f(s);
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0) {
      *step += 1.0f;
      if (mb_index) *mb_index += 1;
    }
  }
}

#include "adam_role.hpp"   // AdamRange, adam_range_block: the sweep itself (shared with the GEMM translation unit, where parts of it ride)

// blocks [0, nb): the sweep of range R; the rest (gather != NULL): the NEXT minibatch's rows -> the static minibatch buffers,
// beside the parameter update.  Nothing of this step reads those buffers any more (stream order), the norm launch has already
// advanced mb_index, and the 1024 random 535-byte rows are pure latency — 5.6 us as a launch of its own in front of the forward pass.
__global__ __launch_bounds__(ADAM_THREADS) void k_adam_apply(AdamRange R, int nb, const GatherArgs *gather) {
  __shared__ float red[8];
  const int b = (int)blockIdx.x;
  if (b >= nb) {
    const GatherArgs A = *gather;
    if ((int64_t)(*A.mb_index) >= A.nsteps) return;   // (the update's last step: nothing follows)
    const int64_t row = ((int64_t)b - nb) * 2 + (threadIdx.x >> 7);
    if (row < A.B) mb_gather_row(A, row, (int)(threadIdx.x & 127u));
    return;
  }
  adam_range_block(R, b, nb, red);
}

// ---- the multi-rank step: the flat buffers are cut into BUCKETS (one per all-reduce / reduce-scatter), every
// bucket into `world` equal SLICES (slice r = what rank r owns after a reduce-scatter): bucket b = float4s [off4[b], off4[b] +
// world * len4[b]).  The norm's partial sums are laid out per (rank, bucket, sub-block) — index (r * nb + b) * J + j — so that
//   * a rank that holds only ITS reduced slices (reduce-scatter) computes its nb * J partials and an all-gather of 4 KB completes
//     the array, and
//   * a rank that holds the whole reduced gradient (all-reduce) computes all of them,
// and both end with the SAME array, summed by every sweep block in the same fixed order: the two forms of the step give
// bit-identical parameters.
constexpr int SHARD_MAX_BUCKETS = 12;
constexpr int64_t SHARD_PER4 = 1024;   // float4s per sweep block (four per thread)
struct ShardGeom {
  int nb, world, J;
  int64_t off4[SHARD_MAX_BUCKETS], len4[SHARD_MAX_BUCKETS];
  int blk[SHARD_MAX_BUCKETS + 1];      // sweep blocks of one rank's slices, prefix sums over the buckets
};

// block ((r - rank_lo) * nb + b) * J + j: the squares of sub-block j of rank r's slice of bucket b; block 0 advances the counters
__global__ __launch_bounds__(ADAM_THREADS) void k_shard_norm(const float *g, ShardGeom G, int rank_lo, float gscale, float *partials,
                                                             float *step, int32_t *mb_index) {
  __shared__ float red[ADAM_THREADS / 64];
  const int e = (int)blockIdx.x, j = e % G.J, rb = e / G.J, b = rb % G.nb, r = rank_lo + rb / G.nb;
  const int64_t len = G.len4[b], chunk = (len + G.J - 1) / G.J;
  const int64_t base = G.off4[b] + (int64_t)r * len;
  const int64_t lo = base + (int64_t)j * chunk, hi = (lo + chunk < base + len) ? lo + chunk : base + len;
  float s = 0.0f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += ADAM_THREADS) {
    float4 x = reinterpret_cast<const float4 *>(g)[i];
    x.x *= gscale; x.y *= gscale; x.z *= gscale; x.w *= gscale;  // (the SUM of the ranks' gradients -> their mean)
    s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  s = wave_sum_f(s);
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[((int64_t)r * G.nb + b) * G.J + j] = (red[0] + red[1]) + (red[2] + red[3]);
    if (e == 0) {
      *step += 1.0f;                  // read by k_shard_apply (a later launch)
      if (mb_index) *mb_index += 1;   // ... and the minibatch counter: the sweep's extra blocks gather the NEXT minibatch with it
    }
  }
}

// blocks [0, sweep): clip + Adam on the slices of ranks [rank_lo, ..) — G.blk[nb] blocks per rank, bucket by bucket; the rest: the
// next minibatch's gather (as k_adam_apply)
__global__ __launch_bounds__(ADAM_THREADS) void k_shard_apply(AdamRange R, ShardGeom G, int rank_lo, int sweep, const GatherArgs *gather) {
  __shared__ float red[8];
  const int e = (int)blockIdx.x;
  if (e >= sweep) {
    const GatherArgs A = *gather;
    if ((int64_t)(*A.mb_index) >= A.nsteps) return;
    const int64_t row = ((int64_t)e - sweep) * 2 + (threadIdx.x >> 7);
    if (row < A.B) mb_gather_row(A, row, (int)(threadIdx.x & 127u));
    return;
  }
  const int per_rank = G.blk[G.nb], r = rank_lo + e / per_rank, q = e % per_rank;
  int b = 0;
  while (b + 1 < G.nb && q >= G.blk[b + 1]) b++;
  R.lo4 = G.off4[b] + (int64_t)r * G.len4[b];
  R.hi4 = R.lo4 + G.len4[b];
  if (e != 0) R.norm_out = nullptr;   // (adam_range_block stores the norm from ITS block 0: only the launch's first block may)
  adam_range_block(R, q - G.blk[b], G.blk[b + 1] - G.blk[b], red);
}
