// brl_kernels.hip — translation unit of libbrl_hip.so: the environment (init / step / observe), the fused rollouts, the evaluators' step
// and the 16-bit inference layer, with their C-ABI entry points (include/brl_hip.h) and the handle.  The PPO update lives in
// brl_ppo.hip, the step's own fp32 GEMM in brl_mlp_gemm.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see brl_amd/build.py)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/brl_hip.h"
#include "bridge_device.hpp"

using namespace brl;

// =====================================================================================
// wave-level helpers (K tables per 64-lane wave, 4 waves per 256-thread workgroup)
// =====================================================================================
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;

// Consecutive table groups go to the same XCD (blocks b and b+8 share one): neighbouring
// rows of the [n,480] / [n,38] outputs share 128-B lines, keep those in ONE L2.  Speed only.
__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nb) {
  return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
}

template <int K>
struct Wave {
  LaneConst c;
  uint8_t *wimg;   // this wave's K x 128 B LDS images
  int tl;          // local table of this lane's logic (lane % K)
  int64_t table0;  // first table of the wave
  int64_t table;   // table of this lane's logic
  bool valid;      // table < n
};

template <int K>
__device__ __forceinline__ Wave<K> wave_begin(uint8_t *lds, const uint64_t *state_in, int64_t n, Tbl &t) {
  Wave<K> w;
  w.c = make_lane_const();
  int wave = (int)(threadIdx.x >> 6);
  int64_t blk = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x);
  w.table0 = (blk * WAVES_PER_BLOCK + wave) * K;
  w.wimg = lds + wave * K * TABLE_BYTES;
  uint64_t *wimg64 = reinterpret_cast<uint64_t *>(w.wimg);
#pragma unroll
  for (int i = w.c.lane; i < K * 16; i += 64) {
    int64_t tb = w.table0 + i / 16;
    wimg64[i] = (state_in != nullptr && tb < n) ? state_in[w.table0 * 16 + i] : 0ull;
  }
  wave_lds_fence();
  w.tl = w.c.lane % K;
  w.table = w.table0 + w.tl;
  w.valid = w.table < n;
  load_scalars(t, w.wimg + w.tl * TABLE_BYTES);
  return w;
}

template <int K>
__device__ __forceinline__ void wave_end(const Wave<K> &w, const Tbl &t, uint64_t *state_out, int64_t n) {
  if (w.c.lane < K) store_scalars(t, w.wimg + w.tl * TABLE_BYTES);
  wave_lds_fence();
  const uint64_t *wimg64 = reinterpret_cast<const uint64_t *>(w.wimg);
#pragma unroll
  for (int i = w.c.lane; i < K * 16; i += 64) {
    int64_t tb = w.table0 + i / 16;
    if (tb < n) state_out[w.table0 * 16 + i] = wimg64[i];
  }
}

template <int K>
__device__ __forceinline__ void wave_or_hist(const Wave<K> &w, int hist_bit) {
  if (w.c.lane < K && hist_bit >= 0) {
    uint32_t *p = reinterpret_cast<uint32_t *>(w.wimg + w.tl * TABLE_BYTES) + (hist_bit >> 5);
    atomicOr(p, 1u << (hist_bit & 31));  // ds_or_b32
  }
}

typedef uint32_t brl_u32x4 __attribute__((ext_vector_type(4)));

// Write-through 16-byte store (sc0 sc1: the bytes go to memory now and the line is not kept in L2).  For everything a
// fused rollout launch writes besides the observations: a plain store leaves a dirty line in the XCD's L2 and all of them — 15 MB of
// mask rows and scalar columns — are written back when the kernel ENDS, after the last wave: 2.6 us of 27.4.
__device__ __forceinline__ void store_wt16(void *p, brl_u32x4 v) {
  // (s_nop: the compiler does not know that the instruction still reads its data registers for two more cycles)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

struct LutRef {
  const int4 *keys;
  const int4 *values;
  uint32_t len;
  const uint4 *hands;  // per row: the four packed hand words (hand_obs[seat] << 4, image words 7..10), derived from keys
};

// the packed hand words of every LUT row, once per upload: what a re-deal copies into a table image
// (same card -> observation-bit mapping as deal_image: obs bit i = rank * 4 + suit, wb5/vis_pgx.py:13-24)
__global__ void k_lut_hands(const int4 *keys, uint4 *hands, int64_t len) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= len) return;
  const int4 k = keys[r];
  uint64_t h[4] = {0ull, 0ull, 0ull, 0ull};
  for (int i = 0; i < 52; i++) {
    const int os_rank = i >> 2, os_suit = i & 3, dsuit = 3 - os_suit, rank = (os_rank + 1) % 13;
    const uint32_t w = (uint32_t)((dsuit == 0) ? k.x : ((dsuit == 1) ? k.y : ((dsuit == 2) ? k.z : k.w)));
    const uint32_t owner = (w >> (2 * (12 - rank))) & 3u;
#pragma unroll
    for (int s = 0; s < 4; s++) h[s] |= (owner == (uint32_t)s) ? (1ull << (4 + i)) : 0ull;
  }
  hands[2 * r] = make_uint4((uint32_t)h[0], (uint32_t)(h[0] >> 32), (uint32_t)h[1], (uint32_t)(h[1] >> 32));
  hands[2 * r + 1] = make_uint4((uint32_t)h[2], (uint32_t)(h[2] >> 32), (uint32_t)h[3], (uint32_t)(h[3] >> 32));
}

// A5 post-step half of auto_reset (src/utils.py:45-55) for every table of the wave that
// just terminated: deal board bctr+1 of that slot, keep (terminated, rewards).
template <int K>
__device__ __forceinline__ void wave_reset(const Wave<K> &w, Tbl &t, bool need, const Rng &g, uint64_t env_offset,
                                           const LutRef &lut, uint32_t next_ctr) {
  uint64_t needm = __ballot(need) & ((1ull << K) - 1ull);
  if (needm == 0ull) return;
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (need) {
    uint32_t keep = t.sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL));
    fresh_scalars(t, g, env_offset + (uint64_t)w.table, next_ctr, lut.len, keep);
    int4 kv = lut.keys[t.lut];
    int4 vv = lut.values[t.lut];
    q0 = (uint32_t)kv.x; q1 = (uint32_t)kv.y; q2 = (uint32_t)kv.z; q3 = (uint32_t)kv.w;
    pack_tricks(t, (uint32_t)vv.x, (uint32_t)vv.y, (uint32_t)vv.z, (uint32_t)vv.w);
  }
  while (needm) {
    int j = __ffsll((unsigned long long)needm) - 1;
    needm &= needm - 1ull;
    deal_image(w.wimg + j * TABLE_BYTES, __builtin_amdgcn_readlane(q0, j), __builtin_amdgcn_readlane(q1, j),
               __builtin_amdgcn_readlane(q2, j), __builtin_amdgcn_readlane(q3, j), w.c);
  }
  wave_lds_fence();
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int j) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, j);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), j);
  return ((uint64_t)hi << 32) | lo;
}

// emit obs/mask rows of the wave's K tables; row of table0 is row0 (rows are consecutive)
template <int K>
__device__ __forceinline__ void wave_emit(const Wave<K> &w, int64_t n, int oseat, uint32_t vulnib, uint64_t legal,
                                          uint8_t *obs, uint8_t *mask, int64_t row0) {
  uint32_t pack = (uint32_t)oseat | (vulnib << 2);
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (w.table0 + j < n) {
      if (obs) {
        uint32_t p = __builtin_amdgcn_readlane(pack, j);
        emit_obs_row(w.wimg + j * TABLE_BYTES, (int)(p & 3u), p >> 2, obs + (row0 + j) * BRL_OBS_SIZE, w.c);
      }
      if (mask) emit_mask_row(readlane64(legal, j), mask + (row0 + j) * BRL_NUM_ACTIONS, w.c);
    }
  }
}

__device__ __forceinline__ float4 rewards_f32(const Tbl &t) {
  return make_float4((float)reward_of(t, 0), (float)reward_of(t, 1), (float)reward_of(t, 2), (float)reward_of(t, 3));
}

__device__ __forceinline__ int sanitize_action(int a, uint32_t &bad) {
  bad = ((uint32_t)a >= (uint32_t)BRL_NUM_ACTIONS) ? 1u : 0u;
  return bad ? 0 : a;
}

// =====================================================================================
// kernels
// =====================================================================================
struct StepOut {
  uint8_t *obs;
  uint8_t *mask;
  float *rewards;
  uint8_t *terminated;
  int32_t *current_player;
};

template <int K>
__device__ __forceinline__ void wave_step_outputs(const Wave<K> &w, const Tbl &t, int64_t n, const StepOut &o) {
  int oseat = cur_seat(t);
  wave_emit<K>(w, n, oseat, vul_nibble(t, oseat), legal_mask(t), o.obs, o.mask, w.table0);
  if (w.c.lane < K && w.valid) {
    if (o.rewards) reinterpret_cast<float4 *>(o.rewards)[w.table] = rewards_f32(t);
    if (o.terminated) o.terminated[w.table] = (uint8_t)bits(t.sc, SC_TERM, 1);
    if (o.current_player) o.current_player[w.table] = cur_player(t);
  }
}

// ---- A1 init(random) ----------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_init_random(uint64_t *state, int64_t n, Rng g, uint64_t env_offset,
                                                               LutRef lut, uint32_t board_ctr0) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, nullptr, n, t);
  t.sc = 0;
  wave_reset<K>(w, t, w.valid, g, env_offset, lut, board_ctr0);
  wave_end<K>(w, t, state, n);
}

// ---- A1 init(explicit deals) — one thread per table (not a hot path) ------------------
__global__ void k_init_explicit(uint64_t *state, int64_t n, const int32_t *hand, const int32_t *dealer,
                                const uint8_t *vul_ns, const uint8_t *vul_ew, const int32_t *shuffled,
                                const uint8_t *tricks) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  uint64_t *s = state + e * BRL_STATE_WORDS;
  for (int i = 0; i < 7; i++) s[i] = 0;
  for (int seat = 0; seat < 4; seat++) {
    uint64_t m = 0;
    for (int i = 0; i < 13; i++) {
      int card = hand[e * 52 + seat * 13 + i];
      int suit = card / 13, rank = card % 13;
      int idx = ((rank + 12) % 13) * 4 + (3 - suit);  // wb5/utils.py:18-19 via pgx card order
      m |= 1ull << idx;
    }
    s[W_HAND + seat] = m << 4;
  }
  uint32_t shuf = 0;
  for (int seat = 0; seat < 4; seat++) shuf |= ((uint32_t)shuffled[e * 4 + seat] & 3u) << (2 * seat);
  uint32_t sc = ((uint32_t)dealer[e] & 3u) | ((uint32_t)(vul_ns[e] != 0) << SC_VULNS) |
                ((uint32_t)(vul_ew[e] != 0) << SC_VULEW) | (shuf << SC_SHUF);
  uint32_t v[4];
  for (int seat = 0; seat < 4; seat++) {
    uint32_t x = 0;
    for (int d = 0; d < 5; d++) x = x * 16u + (tricks[e * 20 + seat * 5 + d] & 15u);
    v[seat] = x;
  }
  Tbl t;
  pack_tricks(t, v[0], v[1], v[2], v[3]);
  s[W_SC] = (uint64_t)sc;
  s[W_FD] = (uint64_t)t.t2 << 32;
  s[W_TR] = (uint64_t)t.t0 | ((uint64_t)t.t1 << 32);
  s[W_CTR] = 0xFFFFFFFFull;
  s[W_REW] = 0;
}

// ---- A2/A5 step ------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_step(const uint64_t *state_in, uint64_t *state_out, int64_t n,
                                                        const int32_t *action, int autoreset, Rng g,
                                                        uint64_t env_offset, LutRef lut, StepOut o) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, state_in, n, t);
  uint32_t bad;
  int a = sanitize_action(w.valid ? action[w.table] : 0, bad);
  if (autoreset) auto_reset_clear(t);
  bool live = !bits(t.sc, SC_TERM, 1);
  int hb = table_step(t, a);
  if (bad && live) t.sc |= (1u << SC_TERM) | (1u << SC_ILLEGAL) | (1u << SC_MASKALL);
  wave_or_hist<K>(w, hb);
  wave_lds_fence();
  if (autoreset) wave_reset<K>(w, t, w.valid && bits(t.sc, SC_TERM, 1), g, env_offset, lut, t.bctr + 1u);
  wave_step_outputs<K>(w, t, n, o);
  wave_end<K>(w, t, state_out, n);
}

// ---- A3 observe --------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_observe(const uint64_t *state, int64_t n, const int32_t *player_id,
                                                           uint8_t *obs, uint8_t *mask) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, state, n, t);
  int oseat = cur_seat(t);
  if (player_id != nullptr && w.valid) oseat = seat_of_player(t, player_id[w.table] & 3);
  wave_emit<K>(w, n, oseat, vul_nibble(t, oseat), legal_mask(t), obs, mask, w.table0);
}

// ---- A7 fused random-policy rollout ------------------------------------------------------
struct RolloutArgs {
  uint64_t *state;
  int64_t n;
  int T;
  int substeps;
  uint32_t draw_base;
  float reward_scale;
  Rng g;
  uint64_t env_offset;
  LutRef lut;
  const float *neg_log_n;  // [39] -log(n) as float, host-computed
  brl_transition out;
  uint8_t *last_obs;   // [n,480] observation of the post-rollout state (runner_state's last_obs), may be NULL
  uint8_t *last_mask;  // [n,38]
  unsigned long long *terminated_count;
  int debug;  // timing experiments only (BRL_DEBUG): 1 = emit waves idle, 2 = loader idle
  // optional (k_rollout_fs only, brl_rollout_random_gae): calc_gae of THIS trajectory by the same launch — with the random
  // policy value == 0, so everything the scan of src/gae.py:20-39 needs besides last_val is produced here
  const float *gae_last_val;
  float gae_gamma, gae_gamma_lambda;
  float *gae_adv, *gae_tgt;  // [T,n]; gae_adv == NULL: off
};

template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_rollout_random(RolloutArgs A) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, A.state, A.n, t);
  const uint64_t env_id = A.env_offset + (uint64_t)w.table;
  uint32_t rb[4] = {0, 0, 0, 0};
  uint32_t rb_idx = 0xFFFFFFFFu;
  uint32_t tcount = 0;
  for (int step = 0; step < A.T; step++) {
    const int64_t row0 = (int64_t)step * A.n + w.table0;
    // G4: the stored obs / mask are the PRE-step view of the acting player
    uint64_t legal = legal_mask(t);
    int oseat = cur_seat(t);
    wave_emit<K>(w, A.n, oseat, vul_nibble(t, oseat), legal, A.out.obs, A.out.legal_action_mask, row0);
    const int actor = player_at(t, oseat);  // src/roll_out.py:72
    int racc0 = 0, racc1 = 0, racc2 = 0, racc3 = 0;
    uint32_t term_any = 0;
    int first_action = 0, first_n = 1;
    for (int k = 0; k < A.substeps; k++) {
      uint32_t draw = A.draw_base + (uint32_t)(step * A.substeps + k);
      if ((draw >> 2) != rb_idx) {
        rb_idx = draw >> 2;
        philox4x32_10((uint32_t)env_id, rb_idx, STREAM_ACTION, (uint32_t)(env_id >> 32), A.g.k0, A.g.k1, rb);
      }
      uint32_t sel = draw & 3u;
      uint32_t u = (sel == 0) ? rb[0] : ((sel == 1) ? rb[1] : ((sel == 2) ? rb[2] : rb[3]));
      if (k > 0) legal = legal_mask(t);
      int nl;
      int a = random_legal_action(t, legal, u, nl);
      if (k == 0) {
        first_action = a;
        first_n = nl;
      }
      auto_reset_clear(t);
      int hb = table_step(t, a);
      wave_or_hist<K>(w, hb);
      wave_lds_fence();
      uint32_t term = bits(t.sc, SC_TERM, 1);
      racc0 += reward_of(t, 0); racc1 += reward_of(t, 1); racc2 += reward_of(t, 2); racc3 += reward_of(t, 3);
      term_any |= term;
      wave_reset<K>(w, t, w.valid && term, A.g, A.env_offset, A.lut, t.bctr + 1u);
    }
    if (A.substeps > 1) {  // src/utils.py:126-128
      set_rewards(t, racc0, racc1, racc2, racc3);
      t.sc = (t.sc & ~(1u << SC_TERM)) | (term_any << SC_TERM);
    }
    if (w.c.lane < K && w.valid) {
      const int64_t row = row0 + w.tl;
      int ra = (actor == 0) ? racc0 : ((actor == 1) ? racc1 : ((actor == 2) ? racc2 : racc3));
      if (A.out.done) A.out.done[row] = (uint8_t)term_any;
      if (A.out.action) A.out.action[row] = first_action;
      if (A.out.value) A.out.value[row] = 0.0f;
      if (A.out.reward) A.out.reward[row] = (float)ra / A.reward_scale;  // G1, src/roll_out.py:90
      if (A.out.log_prob) A.out.log_prob[row] = A.neg_log_n[first_n];
      tcount += term_any;
    }
  }
  if (A.last_obs || A.last_mask) {
    int oseat = cur_seat(t);
    wave_emit<K>(w, A.n, oseat, vul_nibble(t, oseat), legal_mask(t), A.last_obs, A.last_mask, w.table0);
  }
  if (A.terminated_count != nullptr) {  // G2, src/roll_out.py:85
    uint32_t v = (w.c.lane < K && w.valid) ? tcount : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (w.c.lane == 0 && v) atomicAdd(A.terminated_count, (unsigned long long)v);
  }
  wave_end<K>(w, t, A.state, A.n);
}

// ---- A7 fused random-policy rollout, wave-specialised ("ws"): constants and command format; the kernel and
// the description of its roles follow below ---------------------------------------------------------------
constexpr int CMD_WORDS = 4;
constexpr int RING_WORDS = 16;  // hands[8] (the four packed hand words, k_lut_hands) values[4] idx sc_bits pad pad
constexpr int WS_BATCH = 8;     // sub-steps per workgroup barrier
constexpr int WS_RING = 12;     // boards kept ahead per table (see the loader wave)
constexpr uint32_t NO_SLOT = 0xFFu;  // scorer: the table still plays the board it came in with (ring slots are 0..15)
// cmd[s] word 0: about sub-step s-1: [8:0] history bit + 1 (0 none) | [9] deal | [19:16] ring slot dealt
//                | [22:21] acting seat | [28:23] n_legal
//                about state s: [11:10] observer seat | [15:12] vul nibble
// word 1: scalar word `sc` right after sub-step s-1 (before any re-deal)
// word 2: legal mask of state s, low 32 | word 3: [5:0] legal high ; [13:8] action of sub-step s-1

// Command batches: ONE slot first, so that the follower waves — and with them the HBM stores, which the launch is
// bound by once its output is larger than the Infinity Cache — start after one sub-step instead of eight; then 3, 4 and
// 8s: every batch costs the loader (Philox calls of a refill) and the scorer (passes 2 / 3) a fixed ~3 k cycles, more
// than the logic wave needs for one or two sub-steps, so 1- and 2-slot batches after the first are paced by them
// (1,1,2,4,8,.. -> 1,3,4,8,..: 34.0 -> 32.9 us, profiles/r02).  Slot s is entry s - ws_bstart(b) of batch b.
__device__ __forceinline__ int ws_bstart(int b) { return (b < 3) ? ((b == 0) ? 0 : ((b == 1) ? 1 : 4)) : 8 * (b - 2); }  // 0,1,4,8,16,24,..
__device__ __forceinline__ int ws_blen(int b) { return (b < 3) ? ((b == 0) ? 1 : ((b == 1) ? 3 : 4)) : WS_BATCH; }           // 1,3,4,8,8,..
__device__ __forceinline__ int ws_nbatch(int total) {  // batches needed for slots 0..total
  if (total < 8) return (total < 1) ? 1 : ((total < 4) ? 2 : 3);
  return 3 + (total - 8) / WS_BATCH + 1;
}

__device__ __forceinline__ void lds_barrier() {
  // LDS-visible workgroup barrier that leaves global loads/stores in flight (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef BRL_TIMING
#define LDS_BARRIER()                                         \
  do {                                                        \
    unsigned long long _t0 = __builtin_amdgcn_s_memtime();    \
    lds_barrier();                                            \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime();    \
    t_wait += _t1 - _t0;                                      \
    if (t_nb < 16) { t_arr[t_nb] = _t0 - t_begin; t_rel[t_nb] = _t1 - t_begin; t_nb++; } \
  } while (0)
#else
#define LDS_BARRIER() lds_barrier()
#endif

// The T-step scan is latency-bound on ONE dependency chain per table (state(t+1) needs
// state(t)), so the kernel keeps that chain as short as possible and moves everything that does
// not feed it onto other waves of the same workgroup.  One workgroup owns TPB consecutive tables:
//   wave 0        LOGIC  : lane l advances table l in registers — legal mask, action draw,
//                          auction transition, re-deal bookkeeping.  Nothing else: no reward, no
//                          first-denomination table, no HBM access in the loop.
//   wave 1        LOADER : keeps, per table, an LDS ring of the next WS_RING boards of that slot
//                          (Philox -> LUT row -> its packed hand words + DDS values, 48 B) and computes
//                          the Philox action draws one batch ahead.  The only wave that waits on
//                          loads, so nobody else's vmcnt ever includes them.
//   wave 2        SCORER : lane l follows table l behind the logic wave: first denominations; finished
//                          boards are queued, compacted over the tables, and scored one lane per board
//                          (contract + DDS tricks -> reward, A4); sums over sub-steps (G1); writes the
//                          scalar Transition columns, coalesced over tables.  (Runs on hardware wave 3.)
//   waves 3..NW-1 EMIT   : each owns a fixed subset of the tables' LDS images, applies the logic
//                          wave's per-sub-step command (set one history bit, or deal a new board
//                          from the ring) and streams 4 x 480-B observation rows + 4 x 38-B mask
//                          rows per store instruction.  Stores only: they never wait on memory.
// The logic wave posts one 16-byte command per table per sub-step into a double-buffered LDS
// batch of WS_BATCH sub-steps; ONE s_barrier per batch (preceded by lgkmcnt(0) only — global
// loads/stores stay in flight across it).  The other waves work one batch behind, each at its own
// pace, so a slow sub-step on one wave (a deal, a contract to score) is averaged over the batch
// instead of stalling everybody.
// Ring safety: a table deals at most once every 4 sub-steps (the shortest auction is four
// passes), i.e. <= 3 boards per batch.  Boards dealt in batch b are still read by the scorer and
// emit waves during batch b+1, so the loader refills their slots during batch b+2 (finished before
// that batch's barrier); the logic wave, then at most in batch b+3, has consumed <= 9 boards since
// the start of batch b+1 < WS_RING - 1.  (- 1: a board's ring entry is kept one deal longer than that, because
// the scorer reads its DDS values when the board ENDS, i.e. in the batch of the next deal.)
template <int TPB, int NW, int MW = 0>
__global__ __launch_bounds__(NW * 64) void k_rollout_ws(RolloutArgs A) {
  static_assert(TPB <= 64 && NW >= 4 + MW, "logic + loader + scorer + >=1 emit wave (+ mask wave)");
  static_assert(TPB % 4 == 0, "emit waves write 4 consecutive tables per instruction");
  static_assert(MW == 0 || TPB == 32, "the mask wave writes the 32 x 38 mask bytes of a sub-step as 76 16-byte chunks");
  constexpr int NE = NW - 3 - MW;  // MW = 1: the last wave writes every table's legal-mask row instead of the emit waves
  constexpr int B = WS_BATCH;
#ifdef BRL_TIMING
  unsigned long long t_probe0 = 0, t_probe1 = 0, t_probe_n = 0;
  unsigned long long t_wait = 0, t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_arr[16], t_rel[16];
  int t_nb = 0;
#endif
  __shared__ __attribute__((aligned(16))) uint8_t img[TPB * TABLE_BYTES];
  __shared__ __attribute__((aligned(16))) uint32_t cmd[2][B][TPB][CMD_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t ring[TPB][WS_RING][RING_WORDS];
  __shared__ int ring_ready;  // set by the loader wave once the first three boards of every table are in the ring
  __shared__ uint32_t udraw[2][WS_BATCH][TPB];  // action draws of a batch, precomputed by the loader wave
  __shared__ float s_neglog[BRL_NUM_ACTIONS + 2];
  const int tid = (int)threadIdx.x;
  // role index; hardware wave w runs on SIMD w % 4, and with NW = 11 SIMD 3 hosts only two waves: the scorer (the
  // longest chain after the logic wave) takes hardware wave 3 there, the first emit role hardware wave 2
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = (NW >= 11 && !(A.debug & 2048)) ? ((hw_wave == 2) ? 3 : ((hw_wave == 3) ? 2 : hw_wave)) : hw_wave;
  const LaneConst c = make_lane_const();
  const int64_t table0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * TPB;
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    img64[i] = (tb < A.n) ? A.state[table0 * 16 + i] : 0ull;
  }
  if (tid <= BRL_NUM_ACTIONS) s_neglog[tid] = A.neg_log_n[tid];
  const int total = A.T * A.substeps;   // sub-steps; command slots are s = 0..total
  // (MW) the mask wave takes over the legal-mask rows of a workgroup whose 32 tables all exist, in the fast-path batches
  const bool mask_by_wave = (MW != 0) && (A.substeps == 1) && A.out.obs && A.out.legal_action_mask &&
                            !(A.debug & ~(256 | 1024 | 2048)) && (table0 + TPB <= A.n);
  const int nbatch = ws_nbatch(total);  // batches of command slots (ws_bstart / ws_blen)
  const int tl = c.lane;                // logic / loader / scorer: lane = table
  const int tls = (tl < TPB) ? tl : 0;
  const bool valid = (tl < TPB) && (table0 + tl < A.n);
  const uint64_t env_id = A.env_offset + (uint64_t)(table0 + tl);
  // loader state (wave 1): next board to fetch, boards in flight
  uint32_t nb = 0, nb0 = 0, pbase = 0, pidx[3] = {0, 0, 0}, pscb[3] = {0, 0, 0};
  brl_u32x4 pha[3], phb[3], pv[3];  // (native vectors: HIP's uint4 struct arrays are not promoted to registers here)
  uint64_t ctr_word = 0;
  // (loader) lanes 32..63 shadow tables 0..31 in the prologue: they fetch each table's THIRD board in the same Philox
  // pass in which lanes 0..31 fetch the first — the wave is half empty otherwise (TPB = 32 tables)
  const int lt = c.lane & (TPB - 1);
  const bool lup = (TPB == 32) && (c.lane >= TPB);
  const bool lvalid = (table0 + lt < A.n);
  if (wave == 1 && lvalid) ctr_word = A.state[(table0 + lt) * 16 + W_CTR];  // issued now, needed after the barrier
  // action draws (Philox is state-independent, so it does not belong on the logic wave's dependency
  // chain): the loader computes udraw[b & 1][j][table] for command batch b one batch ahead of the logic wave
  uint32_t rbk[4] = {0, 0, 0, 0};
  uint32_t rbk_idx = 0xFFFFFFFFu;
  auto draws = [&](int b) {
    for (int j = 0; j < ws_blen(b); j++) {
      const uint32_t draw = A.draw_base + (uint32_t)(ws_bstart(b) + j);
      if ((draw >> 2) != rbk_idx) {
        rbk_idx = draw >> 2;
        philox4x32_10((uint32_t)env_id, rbk_idx, STREAM_ACTION, (uint32_t)(env_id >> 32), A.g.k0, A.g.k1, rbk);
      }
      const uint32_t sel = draw & 3u;
      if (tl < TPB) udraw[b & 1][j][tl] = (sel == 0) ? rbk[0] : ((sel == 1) ? rbk[1] : ((sel == 2) ? rbk[2] : rbk[3]));
    }
  };
  if (wave == 1) draws(0);
  if (tid == 0) ring_ready = 0;
  __syncthreads();  // images and the draws of batch 0 are in LDS; the ring follows (ring_ready)
  uint32_t pcount = 0;  // (loader) boards whose loads are in flight
  if (wave == 1 && lvalid) {
    // the first three boards of every table: loads ISSUED here, committed to the ring after the first batch
    // barrier (loader loop) — everybody else has already started; only a DEAL needs the ring, and the logic
    // wave checks ring_ready before its first one.  (Three: a table that deals at sub-steps 0 and 4 reads the third
    // entry during the third batch, just after the refill issued behind the first barrier is committed.)
    // Lanes 0..31: boards nb0, nb0 + 1; lanes 32..63: board nb0 + 2 of table lane - 32.
    const uint64_t eid = A.env_offset + (uint64_t)(table0 + lt);
    nb0 = (uint32_t)(ctr_word >> 32) + 1u;
    pbase = nb0 + (lup ? 2u : 0u);
    pcount = lup ? 1u : 2u;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if ((uint32_t)k < pcount) {
        board_params(A.g, eid, pbase + (uint32_t)k, A.lut.len, pidx[k], pscb[k]);
        pha[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k]];
        phb[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k] + 1];
        pv[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.values)[pidx[k]];
      }
    }
    nb = nb0 + ((TPB == 32) ? 3u : 2u);
  }

  if (wave == 1) {
    // ------------------------------------------------------------------ loader wave
    // (its first three boards were fetched in the prologue, before the workgroup's first barrier)
    uint32_t dealt_total = 0, dealt_prev_total = 0;
    for (int bi = 0; bi < nbatch; bi++) {
      if (bi + 1 < nbatch) draws(bi + 1);  // the logic wave starts batch bi+1 right after this barrier
      LDS_BARRIER();
      // commit what was issued one batch ago (its loads landed long before)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if ((uint32_t)k < pcount) {
          uint4 *dst = reinterpret_cast<uint4 *>(&ring[lt][(pbase + (uint32_t)k) % WS_RING][0]);
          brl_u32x4 *dv = reinterpret_cast<brl_u32x4 *>(dst);
          dv[0] = pha[k];
          dv[1] = phb[k];
          dv[2] = pv[k];
          dst[3] = make_uint4(pidx[k], pscb[k], 0u, 0u);
        }
      }
      pcount = 0;
      if (bi == 0) {  // the first three boards are in the ring now
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c.lane == 0) __hip_atomic_store(&ring_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      uint32_t dealt = 0;  // boards this table consumed in batch bi
      for (int j = 0; j < ws_blen(bi); j++) {
        const int s = ws_bstart(bi) + j;
        if (s <= total) dealt += (cmd[bi & 1][j][tls][0] >> 9) & 1u;
      }
      // keep WS_RING boards ahead of what had been consumed by the end of batch bi-1 (slots of boards
      // dealt in batch bi are still being read by the scorer / emit waves): at most 3 fetches per batch,
      // issued now, committed after the next barrier — the loader never holds a barrier up.
      const uint32_t want = nb0 + (uint32_t)WS_RING - 1u + dealt_prev_total;  // (-1: a board's entry lives until the NEXT deal: the scorer reads its DDS values when it ends)
      if (valid && nb < want) {
        pbase = nb;
        pcount = min(3u, want - nb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          if ((uint32_t)k < pcount) {
            board_params(A.g, env_id, nb + (uint32_t)k, A.lut.len, pidx[k], pscb[k]);
            pha[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k]];
            phb[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k] + 1];
            pv[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.values)[pidx[k]];
          }
        }
        nb += pcount;
      }
      dealt_prev_total = dealt_total;
      dealt_total += dealt;
    }
  } else if (wave == 0) {
    // ------------------------------------------------------------------ logic wave
    uint32_t sc, sch, lut, bctr;
    {
      const uint2 *p = reinterpret_cast<const uint2 *>(img + tls * TABLE_BYTES);
      uint2 a = p[W_SC], d = p[W_CTR];
      sc = a.x; sch = a.y; lut = d.x; bctr = d.y;
    }
    __builtin_amdgcn_s_setprio(3);  // the critical chain wins issue arbitration on its SIMD
    // (LUT row, fresh scalars) of the NEXT board of this slot, read ahead of the deal that uses them; the very
    // first read waits for the loader's ring_ready (the ring is filled while the first sub-steps run)
    uint2 nxt = make_uint2(0u, 0u);
    bool have_nxt = false;
    uint32_t pend = 0, pend_act = 0, pend_sc = 0, term_any = 0;
    int sub = 0;
    for (int bi = 0; bi < nbatch; bi++) {
      const int blen = ws_blen(bi);
      uint32_t un = udraw[bi & 1][0][tls];
      for (int j = 0; j < blen; j++) {
        const int s = ws_bstart(bi) + j;
        if (s > total) break;
        const uint32_t u = un;
        un = udraw[bi & 1][(j + 1 < blen) ? j + 1 : j][tls];  // next sub-step's draw, off the chain
        uint32_t nsc = sc, nsch = sch;
        const LeanStep st = lean_random_step(nsc, nsch, u);
        if (tl < TPB) {  // command slot s: what sub-step s-1 did + how state s looks
          uint32_t w0 = pend | ((uint32_t)st.seat << 10) | (vul_nibble_sc(sc, st.seat) << 12);
          uint32_t w3 = ((uint32_t)(st.legal >> 32) & 63u) | (pend_act << 8);
          *reinterpret_cast<uint4 *>(&cmd[bi & 1][j][tl][0]) = make_uint4(w0, pend_sc, (uint32_t)st.legal, w3);
        }
        // the barrier that publishes a batch sits right after its LAST post — before that sub-step is
        // committed — so followers start one sub-step earlier and a deal at sub-step 0 can wait for ring_ready
        // (the loader raises it after the first barrier)
        if (j == blen - 1 || s == total) LDS_BARRIER();
        if (s == total) break;
        const bool first = sub == 0;
        const bool last = sub + 1 == A.substeps;
        sub = last ? 0 : sub + 1;
        term_any = first ? st.term : (term_any | st.term);
        sc = nsc;
        sch = nsch;
        pend_sc = sc;
        pend_act = (uint32_t)st.action;
        const bool deal = valid && st.term;
        const uint32_t slot = (bctr + 1u) % WS_RING;
        pend = st.hb1 | ((uint32_t)deal << 9) | (slot << 16) | ((uint32_t)st.seat << 21) | ((uint32_t)st.n_legal << 23);
        if (!have_nxt && __any(deal)) {  // uniform: first deal of the wave
          while (__hip_atomic_load(&ring_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
          nxt = *reinterpret_cast<const uint2 *>(&ring[tls][(bctr + 1u) % WS_RING][12]);
          have_nxt = true;
        }
        if (deal) {  // A5 post-step half of auto_reset (src/utils.py:45-55): next board from the ring
          sc = nxt.y | (sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL)));
          sch = 0;
          lut = nxt.x;
          bctr += 1u;
          nxt = *reinterpret_cast<const uint2 *>(&ring[tl][(bctr + 1u) % WS_RING][12]);
        }
        if (last && A.substeps > 1) sc = (sc & ~(1u << SC_TERM)) | (term_any << SC_TERM);  // src/utils.py:127
      }
    }
    if (tl < TPB) {
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      p[W_SC] = make_uint2(sc, sch);
      p[W_CTR] = make_uint2(lut, bctr);
    }
  } else if (wave == 2) {
    // ------------------------------------------------------------------ scorer wave
    // Three passes per batch, so that the expensive contract scoring runs once per finished
    // board (<= 2 per table and batch) instead of once per sub-step in which ANY table finishes:
    //   1. per sub-step, cheap: first denominations, new tricks on a re-deal, queue finished boards
    //   2. per queued board: contract -> DDS tricks -> score -> reward vector (A4), summed per macro-step
    //   3. per macro-step: the scalar Transition columns, coalesced over tables
    __shared__ __attribute__((aligned(16))) uint32_t ev[3 * 64][8];  // finished boards of this batch, compacted (<= 3 per table)
    __shared__ __attribute__((aligned(16))) int acc[WS_BATCH][64][4];  // reward sums by player id per macro-step
    __shared__ uint32_t minfo[WS_BATCH][64];                          // per macro-step: actor, action, n_legal, done
    Tbl ts;
    load_scalars(ts, img + tls * TABLE_BYTES);  // fd / tricks / rewards are live here
    int sub = 0;
    uint32_t cur_info = 0, tcount = 0;
    uint32_t vslot = NO_SLOT;  // ring slot of the table's current board; NO_SLOT: the board it came in with
    int64_t row = table0 + tl;  // this table's Transition row of the next macro-step to be written
    int4 last_acc = make_int4(reward_of(ts, 0), reward_of(ts, 1), reward_of(ts, 2), reward_of(ts, 3));
    *reinterpret_cast<int4 *>(&acc[0][tl][0]) = make_int4(0, 0, 0, 0);
    for (int bi = 0; bi < nbatch; bi++) {
      LDS_BARRIER();
#ifdef BRL_TIMING
      const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
      // ---- pass 1
      int nev = 0, m = 0;  // nev: boards queued (uniform); m: macro-steps completed so far in this batch
      // acc[0] carries the partial sums of a macro-step that straddles the batch boundary
#pragma unroll
      for (int q = 1; q < B; q++) *reinterpret_cast<int4 *>(&acc[q][tl][0]) = make_int4(0, 0, 0, 0);
      const int blen = ws_blen(bi);
      uint4 wn = *reinterpret_cast<const uint4 *>(&cmd[bi & 1][0][tls][0]);
      for (int j = 0; j < blen; j++) {
        const int s = ws_bstart(bi) + j;
        if (s > total) break;
        const uint4 w = wn;  // the next command is fetched while this one is processed
        wn = *reinterpret_cast<const uint4 *>(&cmd[bi & 1][(j + 1 < blen) ? j + 1 : j][tls][0]);
        if (s == 0) continue;  // cmd slot 0 describes no sub-step
        const int a = (int)((w.w >> 8) & 63u);
        const int seat = (int)((w.x >> 21) & 3u);
        ts.sc = w.y;
        if (sub == 0)  // first sub-step of a macro-step: the acting player (src/roll_out.py:72), its action
          cur_info = (uint32_t)player_at(ts, seat) | ((uint32_t)a << 2) | (((w.x >> 23) & 63u) << 8);
        note_first_denomination(ts.fd, seat, a);
        const bool fin = (tl < TPB) && bits(ts.sc, SC_TERM, 1);
        const uint64_t fm = __ballot(fin);
        if (fm) {  // queue the finished boards for pass 2, compacted over the tables: one lane per board there
          if (fin) {
            const int pos = nev + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            *reinterpret_cast<uint4 *>(&ev[pos][0]) = make_uint4(ts.sc, ts.fd, (uint32_t)m | ((uint32_t)tl << 8), vslot);
            cur_info |= 1u << 14;  // done (G2)
          }
          nev += __popcll(fm);
        }
        {  // the slot was re-dealt: no strain named yet; the new board's DDS values stay in its ring entry until
           // the board is scored (the loader frees an entry one board late for that)
          const bool dealt = (w.x & 0x200u) != 0u;
          vslot = dealt ? ((w.x >> 16) & 15u) : vslot;
          ts.fd = dealt ? 0u : ts.fd;
        }
        if (++sub == A.substeps) {
          sub = 0;
          minfo[m][tl] = cur_info;
          m++;
        }
      }
      // ---- pass 2
#ifdef BRL_TIMING
      const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
      wave_lds_order();
      for (int e0 = 0; e0 < nev; e0 += 64) {  // one lane per finished board
        const int e = e0 + c.lane;
        if (e < nev) {
          const uint4 q = *reinterpret_cast<const uint4 *>(&ev[e][0]);
          const uint32_t tt = (q.z >> 8) & 63u;
          Tbl tb;
          tb.sc = q.x; tb.fd = q.y;
          if (q.w != NO_SLOT) {  // a board dealt in this launch: DDS values from its ring entry
            const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tt][q.w][8]);
            pack_tricks(tb, vv.x, vv.y, vv.z, vv.w);
          } else {          // the board the table came in with: its tricks are in the packed image
            const uint2 *ip = reinterpret_cast<const uint2 *>(img + tt * TABLE_BYTES);
            const uint2 tr = ip[W_TR], fdw = ip[W_FD];
            tb.t0 = tr.x; tb.t1 = tr.y; tb.t2 = fdw.y;
          }
          terminal_reward(tb);  // A4
          int *ac = &acc[q.z & (WS_BATCH - 1)][tt][0];
          atomicAdd(&ac[0], reward_of(tb, 0)); atomicAdd(&ac[1], reward_of(tb, 1));  // (substeps >= 8: two boards of a
          atomicAdd(&ac[2], reward_of(tb, 2)); atomicAdd(&ac[3], reward_of(tb, 3));  //  table can end in one macro-step)
        }
      }
      // ---- pass 3: the m macro-steps completed in this batch; with TPB <= 32 the upper half of the wave
      //      writes the odd ones, so one store instruction covers two rows of a column
#ifdef BRL_TIMING
      const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
      t_probe0 += ts1 - ts0;
      t_probe1 += ts2 - ts1;
#endif
      wave_lds_order();
      if (m > 0) last_acc = *reinterpret_cast<const int4 *>(&acc[m - 1][tl][0]);
      const bool wide = (TPB == 32) && (table0 + TPB <= A.n) && A.out.done && A.out.action && A.out.value &&
                        A.out.reward && A.out.log_prob;
      if (wide) {
        // every table of the workgroup exists and every column is requested: lane l writes 4 consecutive tables of
        // macro-step l / 8 — ONE 16-byte store per lane and float column covers all 8 macro-steps of a batch
        // (5 store instructions per batch instead of 20)
        const int q = c.lane >> 3, t4 = 4 * (c.lane & 7);
        if (q < m) {
          const uint4 info4 = *reinterpret_cast<const uint4 *>(&minfo[q][t4]);
          const uint32_t inf[4] = {info4.x, info4.y, info4.z, info4.w};
          float rew[4], lgp[4];
          uint32_t act[4], dn = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int4 r = *reinterpret_cast<const int4 *>(&acc[q][t4 + k][0]);
            const int actor = (int)(inf[k] & 3u);
            const int ra = (actor == 0) ? r.x : ((actor == 1) ? r.y : ((actor == 2) ? r.z : r.w));
            rew[k] = (float)ra / A.reward_scale;  // G1, src/roll_out.py:90
            lgp[k] = s_neglog[(inf[k] >> 8) & 63u];
            act[k] = (inf[k] >> 2) & 63u;
            const uint32_t done = (inf[k] >> 14) & 1u;
            dn |= done << (8 * k);
            tcount += done;
          }
          const int64_t rw = (row - tl) + (int64_t)q * A.n + t4;  // row - tl: the workgroup's first table at this macro-step
          *reinterpret_cast<brl_u32x4 *>(A.out.action + rw) = brl_u32x4{act[0], act[1], act[2], act[3]};
          *reinterpret_cast<float4 *>(A.out.value + rw) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          *reinterpret_cast<float4 *>(A.out.reward + rw) = make_float4(rew[0], rew[1], rew[2], rew[3]);
          *reinterpret_cast<float4 *>(A.out.log_prob + rw) = make_float4(lgp[0], lgp[1], lgp[2], lgp[3]);
          *reinterpret_cast<uint32_t *>(A.out.done + rw) = dn;  // G2
        }
        row += (int64_t)m * A.n;
      } else
      {
        constexpr bool TWO = (TPB <= 32);
        const int half = TWO ? (c.lane >> 5) : 0;
        const int tq = TWO ? (c.lane & 31) : c.lane;  // table handled by this lane in pass 3
        const bool vq = (tq < TPB) && (table0 + tq < A.n);
        for (int q0 = 0; q0 < m; q0 += (TWO ? 2 : 1)) {
          const int q = q0 + half;
          if (q < m && vq) {
            const uint32_t info = minfo[q][tq];
            const int4 r = *reinterpret_cast<const int4 *>(&acc[q][tq][0]);
            const int actor = (int)(info & 3u);
            const int ra = (actor == 0) ? r.x : ((actor == 1) ? r.y : ((actor == 2) ? r.z : r.w));
            const uint32_t done = (info >> 14) & 1u;
            const int64_t rw = row + (int64_t)q * A.n + (tq - tl);
            if (A.out.done) A.out.done[rw] = (uint8_t)done;  // G2
            if (A.out.action) A.out.action[rw] = (int32_t)((info >> 2) & 63u);
            if (A.out.value) A.out.value[rw] = 0.0f;
            if (A.out.reward) A.out.reward[rw] = (float)ra / A.reward_scale;  // G1, src/roll_out.py:90
            if (A.out.log_prob) A.out.log_prob[rw] = s_neglog[(info >> 8) & 63u];
            tcount += done;
          }
        }
        row += (int64_t)m * A.n;
      }
      {  // a macro-step still in progress (sub != 0) keeps its partial sums in acc[0]; otherwise zero
        int4 carry = (sub != 0) ? *reinterpret_cast<const int4 *>(&acc[m & (WS_BATCH - 1)][tl][0]) : make_int4(0, 0, 0, 0);
        if (m >= WS_BATCH) carry = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int4 *>(&acc[0][tl][0]) = carry;
      }
    }
    set_rewards(ts, last_acc.x, last_acc.y, last_acc.z, last_acc.w);  // rewards of the last macro-step (src/utils.py:126)
    if (A.terminated_count != nullptr) {  // src/roll_out.py:85
      uint32_t v = tcount;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
#ifndef BRL_TIMING
      if (c.lane == 0 && v) atomicAdd(A.terminated_count, (unsigned long long)v);
#endif
    }
    if (tl < TPB) {
      if (vslot != NO_SLOT) {
        const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tl][vslot][8]);
        pack_tricks(ts, vv.x, vv.y, vv.z, vv.w);
      }
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      p[W_FD] = make_uint2(ts.fd, ts.t2);
      p[W_TR] = make_uint2(ts.t0, ts.t1);
      p[W_REW] = make_uint2(ts.r01, ts.r23);
    }
  } else if (MW != 0 && wave == NW - 1) {
    // ------------------------------------------------------------------ mask wave
    // The 32 legal-mask rows of a sub-step are 1216 contiguous bytes = 76 chunks of 16 B: lane l writes chunk l, lanes
    // < 12 also chunk 64 + l.  A chunk holds the bytes of table ta (from action `off` on) and possibly of ta + 1.
    uint32_t ta[2], tb[2], off[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t cidx = (uint32_t)c.lane + 64u * (uint32_t)q;
      const uint32_t byte0 = 16u * ((cidx < 76u) ? cidx : 75u);
      ta[q] = byte0 / BRL_NUM_ACTIONS;
      off[q] = byte0 - ta[q] * BRL_NUM_ACTIONS;
      tb[q] = (ta[q] + 1u < (uint32_t)TPB) ? ta[q] + 1u : ta[q];
    }
    uint8_t *mrow = A.out.legal_action_mask + table0 * BRL_NUM_ACTIONS;
    const int64_t mstep = A.n * BRL_NUM_ACTIONS;
    for (int bi = 0; bi < nbatch; bi++) {
      LDS_BARRIER();
      const int bstart = ws_bstart(bi), blen = ws_blen(bi);
      if (!mask_by_wave || bstart + blen > total) continue;  // that batch (and every ragged block) is the emit waves'
      for (int j = 0; j < blen; j++) {
        const uint32_t(*cs)[CMD_WORDS] = cmd[bi & 1][j];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (q == 1 && c.lane >= 12) break;
          const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[ta[q]][2]) & ALL_ACTIONS;
          const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[tb[q]][2]) & ALL_ACTIONS;
          const uint32_t bits16 = (uint32_t)((la >> off[q]) | (lb << (BRL_NUM_ACTIONS - off[q])));
          uint32_t d[4];
#pragma unroll
          for (int i = 0; i < 4; i++) d[i] = __umul24((bits16 >> (4 * i)) & 0xFu, 0x204081u) & 0x01010101u;
          *reinterpret_cast<uint4 *>(mrow + 16 * (c.lane + 64 * q)) = make_uint4(d[0], d[1], d[2], d[3]);
        }
        mrow += mstep;
      }
    }
  } else {
    // ------------------------------------------------------------------ emit waves
    const GroupLane gl = make_group_lane();
    const MaskLane ml = make_mask_lane();
    constexpr int NG = TPB / 4;              // groups of 4 consecutive tables
    constexpr int GPW = (NG + NE - 1) / NE;  // groups per emit wave, interleaved to overlap LDS latency
    const bool head = (gl.r < 4) && (gl.ch == 0);  // one lane per row does the row's bookkeeping
    const int rr = (gl.r < 4) ? gl.r : 3;
    int sub = 0;            // s % substeps
    int64_t row0 = table0;  // first Transition row of this workgroup at macro-step s / substeps
    int left[GPW];          // rows of each group that exist (0..4)
#pragma unroll
    for (int k = 0; k < GPW; k++) {
      const int g = (wave - 3) + k * NE;
      int64_t rem = (g < NG && !(A.debug & 1)) ? A.n - (table0 + 4 * g) : 0;
      left[k] = (int)max((int64_t)0, min((int64_t)4, rem));
    }
    // FAST PATH (substeps == 1, every group of this wave complete, obs + mask requested — the BASELINE
    // configuration): the same work as the general loop below with everything loop-invariant hoisted: per-lane
    // output pointers advanced by a constant, no per-step emit / tail / pointer selection.  The slot of the
    // post-rollout state (s == total) is left to the general code.
    bool fast = (A.substeps == 1) && A.out.obs && A.out.legal_action_mask && !(A.debug & ~(256 | 1024 | 2048));
#pragma unroll
    for (int k = 0; k < GPW; k++) fast = fast && (left[k] == 4 || left[k] == 0);
    int bi0 = 0;  // first batch the general loop still has to process
    if (fast) {
      uint8_t *optr[GPW];
      uint32_t *mptr[GPW];
#pragma unroll
      for (int k = 0; k < GPW; k++) {
        const int g = (wave - 3) + k * NE;
        optr[k] = A.out.obs + (table0 + 4 * g) * BRL_OBS_SIZE + gl.out_off;
        mptr[k] = reinterpret_cast<uint32_t *>(A.out.legal_action_mask + (table0 + 4 * g) * BRL_NUM_ACTIONS) + c.lane;
      }
      const int64_t ostep = A.n * BRL_OBS_SIZE, mstep = A.n * BRL_NUM_ACTIONS;
      const bool olane = gl.r < 4;
      for (; bi0 < nbatch; bi0++) {
        const int bstart = ws_bstart(bi0), blen = ws_blen(bi0);
        if (bstart + blen > total) break;  // the batch holding slot `total` goes through the general loop
        LDS_BARRIER();
        for (int j = 0; j < blen; j++) {
          const uint32_t(*cs)[CMD_WORDS] = cmd[bi0 & 1][j];
#pragma unroll
          for (int k = 0; k < GPW; k++) {
            if (left[k] == 0) continue;
            const int g = (wave - 3) + k * NE;
            uint8_t *img_g = img + 4 * g * TABLE_BYTES;
            const uint32_t w0 = cs[4 * g + rr][0];
            if (head && !(w0 & 0x200u) && (w0 & 0x1FFu)) {
              int hb = (int)(w0 & 0x1FFu) - 1;
              atomicOr(reinterpret_cast<uint32_t *>(img_g + gl.r * TABLE_BYTES) + (hb >> 5), 1u << (hb & 31));
            }
            uint64_t dealm = __ballot(head && (w0 & 0x200u));
            while (dealm) {  // rare: ~1 table in 25 per sub-step
              const int l = __ffsll((unsigned long long)dealm) - 1;  // lane 15*q holds row q's command
              dealm &= dealm - 1ull;
              const int q = l / 15;
              const uint32_t wq = __builtin_amdgcn_readlane(w0, l);
              deal_hands(img_g + q * TABLE_BYTES, &ring[4 * g + q][(wq >> 16) & 15u][0], c);
            }
            wave_lds_order();
            uint32_t a;
            uint64_t H;
            obs_chunk_load(img_g, (int)((w0 >> 10) & 3u), gl, a, H);
            {
              GroupLane gz = gl;
              gz.out_off = 0;
              if (olane) obs_chunk_store(a, H, (int)((w0 >> 10) & 3u), (w0 >> 12) & 15u, optr[k], gz);
            }
            optr[k] += ostep;
            if (!mask_by_wave) {  // (uniform)
              const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
              const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
              if (ml.active) *mptr[k] = mask_dword(la, lb, ml);
              mptr[k] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(mptr[k]) + mstep);
            }
          }
        }
      }
      row0 = table0 + (int64_t)ws_bstart(bi0) * A.n;  // substeps == 1: macro-step index == slot index
    }
    for (int bi = bi0; bi < nbatch; bi++) {
      LDS_BARRIER();
      for (int j = 0; j < ws_blen(bi); j++) {
        const int s = ws_bstart(bi) + j;
        if (s > total) break;
        const bool fin = (s == total);  // the post-rollout state: emitted as last_obs / last_mask
        const bool emit = ((s < total) && (sub == 0) && !(A.debug & 2)) || (fin && (A.last_obs || A.last_mask));
        uint8_t *obs_base = fin ? A.last_obs : A.out.obs;
        uint8_t *mask_base = fin ? A.last_mask : A.out.legal_action_mask;
        const int64_t rowb = fin ? table0 : row0;
        const uint32_t(*cs)[CMD_WORDS] = cmd[bi & 1][j];
        // ---- round trip 1: each row's command
        uint32_t w0[GPW];
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          const int g = (wave - 3) + k * NE;
          w0[k] = (left[k] > 0) ? cs[4 * g + rr][0] : 0u;
        }
        // ---- apply sub-step s-1 to the images (one history bit, or a freshly dealt board), then
        //      round trip 2: image dwords + legal masks.  No wait in between: same-wave LDS order.
        uint32_t a[GPW];
        uint64_t H[GPW], la[GPW], lb[GPW];
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          const int g = (wave - 3) + k * NE;
          if (left[k] <= 0) continue;
          uint8_t *img_g = img + 4 * g * TABLE_BYTES;
          const bool is_head = head && (gl.r < left[k]);
          if (is_head && !(w0[k] & 0x200u) && (w0[k] & 0x1FFu)) {
            int hb = (int)(w0[k] & 0x1FFu) - 1;
            atomicOr(reinterpret_cast<uint32_t *>(img_g + gl.r * TABLE_BYTES) + (hb >> 5), 1u << (hb & 31));
          }
          uint64_t dealm = __ballot(is_head && (w0[k] & 0x200u));
          if (dealm) {  // rare: ~1 table in 25 per sub-step
            do {
              const int l = __ffsll((unsigned long long)dealm) - 1;  // lane 15*q holds row q's command
              dealm &= dealm - 1ull;
              const int q = l / 15;
              const uint32_t wq = __builtin_amdgcn_readlane(w0[k], l);
              deal_hands(img_g + q * TABLE_BYTES, &ring[4 * g + q][(wq >> 16) & 15u][0], c);
            } while (dealm);
          }
          wave_lds_order();
          if (emit) {
            obs_chunk_load(img_g, (int)((w0[k] >> 10) & 3u), gl, a[k], H[k]);
            la[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
            lb[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
          }
        }
        if (emit) {
#pragma unroll
          for (int k = 0; k < GPW; k++) {
            const int g = (wave - 3) + k * NE;
            if (left[k] <= 0) continue;
            // ---- the 4 observation rows: two 16-B stores per lane
            if (gl.r < left[k] && obs_base)
              obs_chunk_store(a[k], H[k], (int)((w0[k] >> 10) & 3u), (w0[k] >> 12) & 15u,
                              obs_base + (rowb + 4 * g) * BRL_OBS_SIZE, gl);
            // ---- the 4 mask rows
            if (mask_base) {
              uint8_t *mdst = mask_base + (rowb + 4 * g) * BRL_NUM_ACTIONS;
              if (left[k] >= 4) {  // 152 contiguous bytes, one dword per lane
                if (ml.active) reinterpret_cast<uint32_t *>(mdst)[c.lane] = mask_dword(la[k], lb[k], ml);
              } else {  // ragged tail of the batch: row by row
                for (int q = 0; q < left[k]; q++) {
                  uint64_t lq = *reinterpret_cast<const uint64_t *>(&cs[4 * g + q][2]);
                  emit_mask_row(lq, mdst + q * BRL_NUM_ACTIONS, c);
                }
              }
            }
          }
        }
        if (++sub == A.substeps) {
          sub = 0;
          row0 += (A.debug & 32) ? 0 : A.n;
        }
      }
    }
  }
#ifdef BRL_TIMING
  if (c.lane == 0 && A.terminated_count) {  // timing build only: terminated_count doubles as a dump buffer
    unsigned long long *d = A.terminated_count + ((size_t)blockIdx.x * NW + wave) * 2;
    d[0] = __builtin_amdgcn_s_memtime() - t_begin;
    d[1] = t_wait;
    if (A.debug & 256) {  // timeline: per-barrier arrival / release stamps after the per-wave summary area
      unsigned long long *tl = A.terminated_count + (size_t)gridDim.x * NW * 2 + ((size_t)blockIdx.x * NW + wave) * 32;
      for (int q = 0; q < 16; q++) { tl[2 * q] = (q < t_nb) ? t_arr[q] : 0; tl[2 * q + 1] = (q < t_nb) ? t_rel[q] : 0; }
      if ((A.debug & 1024) || wave == 2) { tl[29] = t_probe0; tl[30] = t_probe1; tl[31] = t_probe_n; }
    }
  }
#endif
  __syncthreads();
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    if (tb < A.n) A.state[table0 * 16 + i] = img64[i];
  }
}

#include "rollout_fs.hpp"  // k_rollout_fs: the flag-synchronised form of the same launch (default for the BASELINE shape)

// ---- policy sub-step: masked categorical over logits + auto_reset(step) -------------------
// What brl_set_rng / brl_set_lut change, mirrored in device memory: the policy sub-step reads it from there instead
// of taking it by value, so that a hipGraph replay of a captured launch follows a later re-seed or LUT rotation
// (ppo.py:525-549) instead of reading freed tables / a stale key.
struct DevCtx {
  LutRef lut;
  Rng g;
  uint64_t env_offset;
};

struct PolicyArgs {
  const uint64_t *state_in;
  uint64_t *state_out;
  int64_t n;
  const float *logits;
  int64_t logits_stride;  // elements between the rows of two consecutive tables (>= 38)
  int mode;
  uint32_t draw;
  const uint32_t *draw_dev;  // optional: the draw index is draw + *draw_dev (hipGraph-captured loops)
  int autoreset;
  const DevCtx *ctx;  // device-resident (see DevCtx)
  int32_t *action;
  float *log_prob;
  StepOut o;  // o.rewards / o.terminated are ACCUMULATED
  brl_macro_ext x;  // optional per-macro-step bookkeeping (brl_policy_step_ex); all-zero otherwise
};

// Masked categorical of one table on the 64 / K lanes that share it (lane l: table l % K, slot l / K): slot s holds the
// NI consecutive actions [s * NI, s * NI + NI).  Returns the chosen action and its log-probability on every lane of the
// table.  `cand`: the actions the distribution ranges over — the legal ones (masked policy, src/roll_out.py:27-29) or
// all 38 (unmasked / illegal-action-penalty policy, src/roll_out.py:33-39).
//   mode bit 0: 0 = pi.sample (inverse CDF in action order with the 24-bit uniform of `u32`), 1 = pi.mode (first max)
// network outputs as the GEMM wrote them: float (fmt 0), bf16 (1) or fp16 (2) -> float (exact conversions)
__device__ __forceinline__ float net_out(const void *base, int64_t idx, int fmt) {
  if (fmt == 0) return reinterpret_cast<const float *>(base)[idx];
  const uint16_t h = reinterpret_cast<const uint16_t *>(base)[idx];
  if (fmt == 1) return __uint_as_float((uint32_t)h << 16);
  _Float16 f16;
  __builtin_memcpy(&f16, &h, 2);
  return (float)f16;
}

__device__ __forceinline__ float net_cvt(uint32_t raw, int fmt) {  // what net_out makes of the bits it loaded
  if (fmt == 0) return __uint_as_float(raw);
  if (fmt == 1) return __uint_as_float(raw << 16);
  const uint16_t h = (uint16_t)raw;
  _Float16 f16;
  __builtin_memcpy(&f16, &h, 2);
  return (float)f16;
}

template <int K>
__device__ __forceinline__ int categorical(const void *logits, int64_t row_off, int fmt, bool valid, uint64_t cand, int mode,
                                           uint32_t u32, int lane, float &log_prob) {
  constexpr int LPT = 64 / K;
  constexpr int NI = (BRL_NUM_ACTIONS + LPT - 1) / LPT;
  const int tl = lane % K, slot = lane / K;
  float lg[NI], e[NI];
  bool ok[NI];
  float mx = -INFINITY;
  int amax = 64;
  // the lane's NI logits: unconditional loads (clamped index), the format decided ONCE around all of them — a select or a
  // format branch per element makes hipcc branch around every load and wait for each one (NI memory round trips)
  uint32_t raw[NI];
  const int64_t ro = valid ? row_off : 0;
  if (fmt == 0) {
#pragma unroll
    for (int i = 0; i < NI; i++) raw[i] = reinterpret_cast<const uint32_t *>(logits)[ro + min(slot * NI + i, BRL_NUM_ACTIONS - 1)];
  } else {
#pragma unroll
    for (int i = 0; i < NI; i++) raw[i] = reinterpret_cast<const uint16_t *>(logits)[ro + min(slot * NI + i, BRL_NUM_ACTIONS - 1)];
  }
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int a = slot * NI + i;
    const bool in = a < BRL_NUM_ACTIONS;
    lg[i] = (in && valid) ? net_cvt(raw[i], fmt) : 0.0f;
    ok[i] = in && ((cand >> (a & 63)) & 1ull);
    if (ok[i] && lg[i] > mx) {  // first maximum wins, like argmax
      mx = lg[i];
      amax = a;
    }
  }
#pragma unroll
  for (int off = K; off < 64; off <<= 1) {
    const float omx = __shfl_xor(mx, off, 64);
    const int oam = __shfl_xor(amax, off, 64);
    const bool take = (omx > mx) || (omx == mx && oam < amax);
    mx = take ? omx : mx;
    amax = take ? oam : amax;
  }
  amax = (amax >= BRL_NUM_ACTIONS) ? 0 : amax;  // (no finite candidate logit: NaN / -inf everywhere)
  float own = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; i++) {
    e[i] = ok[i] ? expf(lg[i] - mx) : 0.0f;
    own += e[i];
  }
  // inclusive scan of the slots' sums in action order
  float incl = own;
#pragma unroll
  for (int off = 1; off < LPT; off <<= 1) {
    const float v = __shfl_up(incl, off * K, 64);
    incl += (slot >= off) ? v : 0.0f;
  }
  const float total = __shfl(incl, (LPT - 1) * K + tl, 64);
  float excl = __shfl_up(incl, K, 64);
  excl = (slot == 0) ? 0.0f : excl;
  int act = amax;
  if (!(mode & 1)) {
    const float target = (float)(u32 >> 8) * (1.0f / 16777216.0f) * total;  // inverse CDF, u in [0,1)
    float cum = excl;
    int first = 64, last = -1;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      cum += e[i];
      if (ok[i]) {
        last = slot * NI + i;
        first = (first == 64 && cum > target) ? slot * NI + i : first;
      }
    }
#pragma unroll
    for (int off = K; off < 64; off <<= 1) {
      first = min(first, __shfl_xor(first, off, 64));
      last = max(last, __shfl_xor(last, off, 64));
    }
    act = (first < 64) ? first : max(last, 0);
  }
  // the chosen action's logit lives on slot act / NI
  const int ai = act % NI;
  float sel = lg[0];
#pragma unroll
  for (int i = 1; i < NI; i++) sel = (ai == i) ? lg[i] : sel;
  const float la = __shfl(sel, (act / NI) * K + tl, 64);
  log_prob = (la - mx) - logf(total);
  return act;
}

// HEADS: the 39 head outputs (38 logits + value) of the workgroup's 16 tables are formed HERE from the last hidden layer's
// activations (A.x.head_h, bf16 / fp16) and the head weights: one 16-row MFMA tile, the K = hidden sum split over the four waves
// (v_mfma_f32_16x16x32_bf16 / _f16, fp32 accumulation), partial sums through LDS.  Replaces the N = 39 library GEMM in front of
// every policy sub-step (9.9 us at n = 8192 — as long as the four 1024-wide layers' share of a forward) and the round trip of
// its output.  K == 4 only (4 waves x 4 tables = the tile's 16 rows).
typedef short pol_b16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pol_f16x8 __attribute__((ext_vector_type(8)));
typedef float pol_f32x4 __attribute__((ext_vector_type(4)));
constexpr int POL_HD = BRL_NUM_ACTIONS + 1;   // 39

// HEADS == 2: the heads come as PARTIAL products — head_part[p][table][head], p = the column tile of the last hidden layer
// whose launch (brl_linear_act_heads) multiplied its 128 columns with the head weights while it held them: logits = head_b +
// the parts in order.  The last hidden layer is then never written to memory, and this launch reads 1.2 KB per table
// instead of the 2 KB row + its share of the head weights.
template <int K, int HEADS = 0>
__global__ __launch_bounds__(BLOCK_THREADS) void k_policy_step(PolicyArgs A) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  __shared__ float hd_red[HEADS == 1 ? 4 : 1][3][4][64];
  __shared__ float hd_logits[HEADS ? 16 : 1][POL_HD + 1];
  static_assert(!HEADS || K == 4, "the head tile is the workgroup's 16 tables");
  // (HEADS) every operand of the heads product — this wave's quarter of K: 8 + 24 16-byte loads at hidden = 1024 — is requested
  // BEFORE the table images are fetched, so that the two memory round trips overlap; the MFMAs follow wave_begin
  constexpr int HGP = HEADS == 1 ? 8 : 1;
  pol_b16x8 hav[HGP], hbv[HGP][3];
  bool hbok[3] = {false, false, false};
  int hsteps = 0;
  constexpr int HPE = (16 * POL_HD + BLOCK_THREADS - 1) / BLOCK_THREADS;   // head values per thread (3)
  float hpv[HEADS == 2 ? HPE : 1][8];
  if (HEADS == 2) {   // (requested before the table images are fetched, like the operands of HEADS == 1)
    const int64_t row0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * 16;
#pragma unroll
    for (int i = 0; i < HPE; i++) {
      const int e = (int)threadIdx.x + BLOCK_THREADS * i;
      const int row = (e < 16 * POL_HD) ? e / POL_HD : 0, col = (e < 16 * POL_HD) ? e - row * POL_HD : 0;
      const int64_t tb = (row0 + row < A.n) ? row0 + row : A.n - 1;
      const float *bp = A.x.head_part + tb * A.x.head_part_ld + col;
#pragma unroll
      for (int p = 0; p < 8; p++) hpv[i][p] = (p < A.x.head_nparts) ? bp[(int64_t)p * A.x.head_part_stride] : 0.0f;
    }
  }
  if (HEADS == 1) {
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    const int64_t row0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * 16;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t arow = (row0 + r < A.n) ? row0 + r : A.n - 1;
    const uint16_t *ap = reinterpret_cast<const uint16_t *>(A.x.head_h) + arow * A.x.head_ldh + 8 * kq;
    hsteps = A.x.head_hidden / 32;   // 32-deep K steps; wave wv takes steps wv, wv + 4, ...
#pragma unroll
    for (int nb = 0; nb < 3; nb++) hbok[nb] = 16 * nb + r < POL_HD;
#pragma unroll
    for (int u = 0; u < HGP; u++) {
      const int st = (wv + 4 * u < hsteps) ? wv + 4 * u : wv;
      hav[u] = *reinterpret_cast<const pol_b16x8 *>(ap + 32 * st);
#pragma unroll
      for (int nb = 0; nb < 3; nb++)
        hbv[u][nb] = *reinterpret_cast<const pol_b16x8 *>(reinterpret_cast<const uint16_t *>(A.x.head_w) +
                                                         (int64_t)(hbok[nb] ? 16 * nb + r : 0) * A.x.head_hidden + 8 * kq + 32 * st);
    }
  }
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, A.state_in, A.n, t);
  if (HEADS == 2) {
    const int64_t row0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * 16;
#pragma unroll
    for (int i = 0; i < HPE; i++) {
      const int e = (int)threadIdx.x + BLOCK_THREADS * i;
      if (e < 16 * POL_HD) {
        const int row = e / POL_HD, col = e - row * POL_HD;
        float v = A.x.head_b[col];
#pragma unroll
        for (int p = 0; p < 8; p++) v += hpv[i][p];   // fixed order (absent parts are exact zeros)
        if (A.x.head_nparts > 8) {
          const int64_t tb = (row0 + row < A.n) ? row0 + row : A.n - 1;
          for (int p = 8; p < A.x.head_nparts; p++) v += A.x.head_part[(int64_t)p * A.x.head_part_stride + tb * A.x.head_part_ld + col];
        }
        hd_logits[row][col] = v;
      }
    }
    __syncthreads();
  }
  if (HEADS == 1) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    pol_f32x4 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; nb++) acc[nb] = pol_f32x4{0.f, 0.f, 0.f, 0.f};
    const pol_b16x8 zero8 = pol_b16x8{0, 0, 0, 0, 0, 0, 0, 0};
    auto mma = [&](const pol_b16x8 &a, const pol_b16x8 &b, pol_f32x4 c) {
      return (A.x.head_fmt == 1) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pol_f16x8, a), __builtin_bit_cast(pol_f16x8, b), c, 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < HGP; u++) {
      if (wv + 4 * u >= hsteps) break;
#pragma unroll
      for (int nb = 0; nb < 3; nb++) acc[nb] = mma(hav[u], hbok[nb] ? hbv[u][nb] : zero8, acc[nb]);
    }
    if (hsteps > 4 * HGP) {   // hidden > 1024: the remaining steps, four in flight
      const int64_t row0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * 16;
      const int64_t arow = (row0 + r < A.n) ? row0 + r : A.n - 1;
      const uint16_t *ap = reinterpret_cast<const uint16_t *>(A.x.head_h) + arow * A.x.head_ldh + 8 * kq;
      for (int st = wv + 4 * HGP; st < hsteps; st += 4) {
        const pol_b16x8 a = *reinterpret_cast<const pol_b16x8 *>(ap + 32 * st);
#pragma unroll
        for (int nb = 0; nb < 3; nb++) {
          const pol_b16x8 b = *reinterpret_cast<const pol_b16x8 *>(reinterpret_cast<const uint16_t *>(A.x.head_w) +
                                                                   (int64_t)(hbok[nb] ? 16 * nb + r : 0) * A.x.head_hidden + 8 * kq + 32 * st);
          acc[nb] = mma(a, hbok[nb] ? b : zero8, acc[nb]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < 3; nb++)
#pragma unroll
      for (int q = 0; q < 4; q++) hd_red[wv][nb][q][lane] = acc[nb][q];   // D[table 4 (lane >> 4) + q][head 16 nb + (lane & 15)]
    __syncthreads();
    for (int e = tid; e < 16 * POL_HD; e += BLOCK_THREADS) {
      const int row = e / POL_HD, col = e - row * POL_HD;
      const int nb = col >> 4, c = col & 15, q = row & 3, rq = row >> 2;
      float v = A.x.head_b[col];
#pragma unroll
      for (int k = 0; k < 4; k++) v += hd_red[k][nb][q][16 * rq + c];   // fixed order
      hd_logits[row][col] = v;
    }
    __syncthreads();
  }
  const DevCtx cx = *A.ctx;
  const uint64_t legal = legal_mask(t);
  const uint64_t cand = (A.mode & 2) ? ALL_ACTIONS : legal;  // bit 1: the unmasked policy
  // what the epilogue adds to / copies (accumulators of the macro-step, the critic's value, the acting player): fetched
  // now, next to the logits, instead of as three more memory round trips in front of the last stores
  const bool owner = w.c.lane < K && w.valid;
  const int64_t otab = owner ? w.table : 0;
  float4 old_rw = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t old_term = 0, val_raw = 0, actor_id = 0;
  if (A.o.rewards && !A.x.first) old_rw = reinterpret_cast<const float4 *>(A.o.rewards)[otab];
  if (A.o.terminated && !A.x.first) old_term = A.o.terminated[otab];
  const int in_fmt = HEADS ? 0 : A.x.in_fmt;   // (the in-kernel heads are fp32)
  if (A.x.value_out) {
    if (HEADS) val_raw = __float_as_uint(hd_logits[(int)(threadIdx.x >> 6) * K + (w.c.lane < K ? w.c.lane : 0)][BRL_NUM_ACTIONS]);
    else if (A.x.in_fmt == 0) val_raw = reinterpret_cast<const uint32_t *>(A.x.value_in)[otab * A.x.value_stride];
    else val_raw = reinterpret_cast<const uint16_t *>(A.x.value_in)[otab * A.x.value_stride];
  }
  if (A.x.last && A.x.reward_out) actor_id = (uint32_t)A.x.actor[otab];
  uint32_t u32 = 0;
  if (!(A.mode & 1)) {
    uint32_t r[4];
    const uint64_t env_id = cx.env_offset + (uint64_t)w.table;
    const uint32_t draw = A.draw + (A.draw_dev ? *A.draw_dev : 0u);
    philox4x32_10((uint32_t)env_id, draw >> 2, STREAM_ACTION, (uint32_t)(env_id >> 32), cx.g.k0, cx.g.k1, r);
    const uint32_t sel = draw & 3u;
    u32 = (sel == 0) ? r[0] : ((sel == 1) ? r[1] : ((sel == 2) ? r[2] : r[3]));
  }
  float lp;
  const int act = HEADS ? categorical<K>(&hd_logits[0][0], (int64_t)((int)(threadIdx.x >> 6) * K + w.tl) * (POL_HD + 1), 0, w.valid, cand,
                                         A.mode, u32, w.c.lane, lp)
                        : categorical<K>(A.logits, (w.valid ? w.table : 0) * A.logits_stride, A.x.in_fmt, w.valid, cand, A.mode, u32,
                                         w.c.lane, lp);
  if (A.autoreset) auto_reset_clear(t);
  int hb = table_step(t, act);
  wave_or_hist<K>(w, hb);
  wave_lds_fence();
  uint32_t term = bits(t.sc, SC_TERM, 1);
  float4 rw = rewards_f32(t);
  if (A.autoreset) wave_reset<K>(w, t, w.valid && term, cx.g, cx.env_offset, cx.lut, t.bctr + 1u);
  int oseat = cur_seat(t);
  wave_emit<K>(w, A.n, oseat, vul_nibble(t, oseat), legal_mask(t), A.o.obs, A.o.mask, w.table0);
  if (A.x.obs_cast != nullptr) {  // the same observation as the next forward's input (float / bf16 / fp16)
    const uint32_t pack = (uint32_t)oseat | (vul_nibble(t, oseat) << 2);
    const int esz = (A.x.obs_fmt == 0) ? 4 : 2;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (w.table0 + j < A.n) {
        const uint32_t p = __builtin_amdgcn_readlane(pack, j);
        emit_obs_row_cast(w.wimg + j * TABLE_BYTES, (int)(p & 3u), p >> 2,
                          reinterpret_cast<uint8_t *>(A.x.obs_cast) + (w.table0 + j) * BRL_OBS_SIZE * esz, A.x.obs_fmt, w.c);
      }
    }
  }
  uint32_t tacc = 0;
  if (w.c.lane < K && w.valid) {
    if (A.action) A.action[w.table] = act;
    if (A.log_prob) A.log_prob[w.table] = lp;
    float4 tot = rw;
    if (A.o.rewards) {
      if (!A.x.first) tot = make_float4(old_rw.x + rw.x, old_rw.y + rw.y, old_rw.z + rw.z, old_rw.w + rw.w);
      reinterpret_cast<float4 *>(A.o.rewards)[w.table] = tot;
    }
    tacc = term;
    if (A.o.terminated) {
      if (!A.x.first) tacc |= old_term;
      A.o.terminated[w.table] = (uint8_t)tacc;
    }
    if (A.o.current_player) A.o.current_player[w.table] = cur_player(t);
    if (A.x.value_out) A.x.value_out[w.table] = net_cvt(val_raw, in_fmt);                    // src/roll_out.py:76
    if (A.x.last) {
      if (A.x.done_out) A.x.done_out[w.table] = (uint8_t)tacc;                                // G2
      if (A.x.reward_out) {                                                                  // G1, src/roll_out.py:90
        const int a = (int)(actor_id & 3u);
        const float r = (a == 0) ? tot.x : ((a == 1) ? tot.y : ((a == 2) ? tot.z : tot.w));
        A.x.reward_out[w.table] = r / A.x.reward_scale;
      }
    }
  }
  if (A.x.last && A.x.terminated_count) {                                                    // src/roll_out.py:85
    // one atomic per WORKGROUP: every wave adding to the one address cost this launch 20 us at 8192 tables (2048 same-address
    // atomics, ~10 ns each); callers that can, sum `done_out` afterwards instead (brl_amd/roll_out.py does)
    __shared__ uint32_t tc_s[WAVES_PER_BLOCK];
    const uint64_t m = __ballot(tacc != 0u);
    if (w.c.lane == 0) tc_s[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t c = 0;
#pragma unroll
      for (int k = 0; k < WAVES_PER_BLOCK; k++) c += tc_s[k];
      if (c) atomicAdd(reinterpret_cast<unsigned long long *>(A.x.terminated_count), (unsigned long long)c);
    }
  }
  wave_end<K>(w, t, A.state_out, A.n);
}

// ---- A12 duplicate_step ---------------------------------------------------------------------
__device__ __forceinline__ float4 imp_vector(float a0, float b0) {
  const float th[24] = {20, 50, 90, 130, 170, 220, 270, 320, 370, 430, 500, 600,
                        750, 900, 1100, 1300, 1500, 1750, 2000, 2250, 2500, 3000, 3500, 4000};
  float d = a0 + b0;
  float win = (d >= 0.0f) ? 1.0f : -1.0f;  // src/duplicate.py:52-54
  float ad = fabsf(d);
  int imp = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) imp += (ad >= th[i]) ? 1 : 0;  // src/duplicate.py:46-69
  float v = (float)imp * win;
  return make_float4(v, v, -v, -v);
}

// Probability mass the UNMASKED softmax of a table's logits puts on illegal actions — `jnp.dot(pi.probs, ~mask)` of the
// evaluators' step log (src/evaluation.py:664-665).  Same lane layout as categorical<K>.
template <int K>
__device__ __forceinline__ float illegal_mass(const float *logits_row, bool valid, uint64_t legal, int lane) {
  constexpr int LPT = 64 / K;
  constexpr int NI = (BRL_NUM_ACTIONS + LPT - 1) / LPT;
  const int slot = lane / K;
  float lg[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int a = slot * NI + i;
    lg[i] = (a < BRL_NUM_ACTIONS && valid) ? logits_row[a] : -INFINITY;
    mx = fmaxf(mx, lg[i]);
  }
#pragma unroll
  for (int off = K; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float all = 0.0f, ill = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int a = slot * NI + i;
    const float e = (a < BRL_NUM_ACTIONS && valid) ? expf(lg[i] - mx) : 0.0f;
    all += e;
    ill += ((legal >> (a & 63)) & 1ull) ? 0.0f : e;
  }
#pragma unroll
  for (int off = K; off < 64; off <<= 1) {
    all += __shfl_xor(all, off, 64);
    ill += __shfl_xor(ill, off, 64);
  }
  return ill / all;
}

// One iteration of the evaluators' loops (src/evaluation.py:87-204 simple duplicate, :583-1032 duplicate with bidding
// statistics, :229-582 single table): the action — given, or the greedy call of the network whose team is to act
// (players {0,1} = team 1, src/evaluation.py:146-151) —, the step log, duplicate_step (src/duplicate.py:147-192) or a
// plain env.step, and the return accumulators.
struct EvalArgs {
  const uint64_t *state_in;
  uint64_t *state_out;
  int64_t n;
  const int32_t *action;  // the calls to make, or NULL: arg-max of logits1 / logits2 by team
  const float *logits1, *logits2;
  int64_t stride1, stride2;
  int duplicate;          // 1: duplicate_step with TA / TB; 0: env.step
  brl_table_info TA, TB;
  brl_eval_stats S;       // any member may be NULL
  int bid_set;            // the single-table evaluator marks a bid made (.set(1)), the duplicate one counts it
  float *cum_return;      // [n] += rewards[0] of the step (src/evaluation.py:167-169)
  float *rewards_sum;     // [n,4] += rewards (src/evaluation.py:400; single-table evaluator)
  int32_t *action_out;    // [n] the call made
  StepOut o;
  int acting_team;        // -1: every board acts (the reference's loop); 0 / 1: only the boards whose turn it is for THAT team
                          // act, the others wait (finished boards always take their no-op step) — brl_eval_step_team
  float *obs_f32;         // optional [n,480]: the new observation as the next forward's input (`.astype(jnp.float32)`) too
};

template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_eval_step(EvalArgs A) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES_PER_BLOCK * K * TABLE_BYTES];
  Tbl t;
  Wave<K> w = wave_begin<K>(lds, A.state_in, A.n, t);
  const bool was_term = bits(t.sc, SC_TERM, 1);
  const int team = cur_player(t) >> 1;  // 0: players {0,1}
  const bool idle = (A.acting_team >= 0) && !was_term && (team != A.acting_team);  // waits for its team's iteration
  const uint64_t legal = legal_mask(t);
  uint32_t bad = 0;
  int a;
  float mass = 0.0f;
  if (A.action != nullptr) {
    a = sanitize_action(w.valid ? A.action[w.table] : 0, bad);
  } else {
    const int64_t tb = w.valid ? w.table : 0;
    const float *row = team ? A.logits2 + tb * A.stride2 : A.logits1 + tb * A.stride1;
    float lp;
    a = categorical<K>(row, 0, 0, w.valid, legal, 1, 0u, w.c.lane, lp);  // masked_pi.mode()
    if (A.S.illegal_prob_sum) mass = illegal_mass<K>(row, w.valid, legal, w.c.lane);
  }
  if (w.c.lane < K && w.valid && !was_term && !idle) {  // make_step_log: finished boards log nothing (src/evaluation.py:736-748)
    const int64_t st = w.table * 2 + team;
    if (A.S.illegal_prob_sum) A.S.illegal_prob_sum[st] += mass;
    if (A.S.step_count) A.S.step_count[st] += 1;
    if (A.S.pass_count && a == 0) A.S.pass_count[st] += 1;
    if (A.S.bid_count && a >= 3) {
      int32_t *bc = A.S.bid_count + st * 35 + (a - 3);
      *bc = A.bid_set ? 1 : *bc + 1;
    }
  }
  int hb = idle ? -1 : table_step(t, a);  // src/duplicate.py:149
  if (bad && !was_term && !idle) t.sc |= (1u << SC_TERM) | (1u << SC_ILLEGAL) | (1u << SC_MASKALL);
  wave_or_hist<K>(w, hb);
  wave_lds_fence();
  if (A.duplicate) {
    bool term = !idle && bits(t.sc, SC_TERM, 1);
    bool a_done = w.valid ? (A.TA.terminated[w.table] != 0) : true;
    bool b_done = w.valid ? (A.TB.terminated[w.table] != 0) : true;
    bool to_b = w.valid && !a_done && term;                // table A just ended -> replay the board seat-swapped
    bool emit_imp = w.valid && a_done && term && !b_done;  // table B just ended -> IMP once (G8)
    float4 rw = rewards_f32(t);
    // snapshots (src/duplicate.py:165-188) of the state as stepped
    if (w.c.lane < K && (to_b || emit_imp)) {
      const brl_table_info &T = to_b ? A.TA : A.TB;
      T.terminated[w.table] = 1;
      reinterpret_cast<float4 *>(T.rewards)[w.table] = rw;
      T.last_bid[w.table] = (int)bits(t.sc, SC_LB1, 6) - 1;
      T.last_bidder[w.table] = bits(t.sc, SC_LB1, 6) ? player_at(t, (int)bits(t.sc, SC_LBSEAT, 2)) : -1;
      T.call_x[w.table] = (uint8_t)bits(t.sc, SC_X, 1);
      T.call_xx[w.table] = (uint8_t)bits(t.sc, SC_XX, 1);
    }
    if (emit_imp) {
      float4 ar = reinterpret_cast<const float4 *>(A.TA.rewards)[w.table];
      float4 v = imp_vector(ar.x, rw.x);  // src/duplicate.py:157-160
      set_rewards(t, (int)v.x, (int)v.y, (int)v.z, (int)v.w);
    } else if (!idle) {
      t.r01 = 0;  // src/duplicate.py:162
      t.r23 = 0;
    }
    // _duplicate_init (src/duplicate.py:113-128): same hands / dealer / vulnerabilities,
    // seats [1,0,3,2], everything else back to defaults
    uint64_t tobm = __ballot(to_b) & ((1ull << K) - 1ull);
    if (to_b) {
      uint32_t sh = bits(t.sc, SC_SHUF, 8);
      uint32_t sw = ((sh >> 2) & 0x03u) | ((sh & 0x03u) << 2) | ((sh >> 2) & 0x30u) | ((sh & 0x30u) << 2);
      t.sc = (t.sc & 0xFu) | (sw << SC_SHUF);
      t.sch = 0;
      t.fd = 0;
    }
    if (tobm) {
      uint64_t *wimg64 = reinterpret_cast<uint64_t *>(w.wimg);
      for (int j = 0; j < K; j++)
        if (((tobm >> j) & 1ull) && w.c.lane < 7) wimg64[j * 16 + w.c.lane] = 0ull;
      wave_lds_fence();
    }
  }
  if (w.c.lane < K && w.valid) {
    if (A.action_out) A.action_out[w.table] = idle ? -1 : a;
    if (A.cum_return && !idle) A.cum_return[w.table] += (float)reward_of(t, 0);
    if (A.rewards_sum && !idle) {
      float4 *p = reinterpret_cast<float4 *>(A.rewards_sum) + w.table;
      const float4 old = *p, rw = rewards_f32(t);
      *p = make_float4(old.x + rw.x, old.y + rw.y, old.z + rw.z, old.w + rw.w);
    }
  }
  wave_step_outputs<K>(w, t, A.n, A.o);
  if (A.obs_f32 != nullptr) {  // (as k_policy_step's obs_cast: the cast launch in front of a full-batch forward disappears)
    const int oseat = cur_seat(t);
    const uint32_t pack = (uint32_t)oseat | (vul_nibble(t, oseat) << 2);
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (w.table0 + j < A.n) {
        const uint32_t p = __builtin_amdgcn_readlane(pack, j);
        emit_obs_row_cast(w.wimg + j * TABLE_BYTES, (int)(p & 3u), p >> 2,
                          reinterpret_cast<uint8_t *>(A.obs_f32) + (w.table0 + j) * BRL_OBS_SIZE * 4, 0, w.c);
      }
    }
  }
  wave_end<K>(w, t, A.state_out, A.n);
}

// The evaluators' end-of-run statistics (src/evaluation.py:841-1031 make_terminated_log / make_contract_log and the
// sums behind log_info): one thread per board, integer histograms in LDS, one atomic per bin and block.
//   out[tb * EV_TABLE + ...], tb = 0 (table A) / 1 (table B):
//     +0 pass-outs  +1/+2 doubled / redoubled contracts of team 1  +3/+4 of team 2  +5 team-1 "make"  +6 team-2 "make"
//     +7 team-1 "down"  +8 team-2 "down" (the reference's labels: rewards[0] >= 0 x declaring team, :951-984)
//     +9 sum of rewards[0] (table score of player 0)  +10..+44 team-1 contracts by bid  +45..+79 team-2 contracts
//   out[2 * EV_TABLE + 35 * team + bid] = how often the team made the bid (sum of bid_count over boards)
//   out[2 * EV_TABLE + 70] = sum of the final states' _step_count
constexpr int EV_TABLE = 80, EV_TOTAL = 2 * EV_TABLE + 71;
__global__ __launch_bounds__(256) void k_eval_reduce(int64_t n, brl_table_info TA, brl_table_info TB, int two_tables,
                                                     const int32_t *bid_count, const uint64_t *state,
                                                     long long *out) {
  __shared__ long long h[EV_TOTAL];
  for (int i = threadIdx.x; i < EV_TOTAL; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto add = [&](int i, long long v) { atomicAdd(reinterpret_cast<unsigned long long *>(&h[i]), (unsigned long long)v); };
  if (e < n) {
    for (int tb = 0; tb < (two_tables ? 2 : 1); tb++) {
      const brl_table_info &T = tb ? TB : TA;
      const int base = tb * EV_TABLE;
      const int lb = T.last_bid[e], who = T.last_bidder[e];
      const float r0 = T.rewards[e * 4];
      add(base + 9, (long long)r0);
      if (who == -1 && lb == -1) {  // passed out (G13, src/evaluation.py:841-842)
        add(base + 0, 1);
      } else {
        const int team = (who < 2) ? 0 : 1;
        add(base + 10 + 35 * team + lb, 1);
        if (T.call_x[e]) add(base + 1 + 2 * team, 1);
        if (T.call_xx[e]) add(base + 2 + 2 * team, 1);
        add(base + ((r0 >= 0.0f) ? 5 : 7) + team, 1);
      }
    }
    if (bid_count) {
      // 70 counters per board = 35 aligned 8-byte pairs, fetched 7 pairs at a time and only then added (a load + a
      // conditional add per element compiles to 70 dependent memory round trips)
      const int2 *bc = reinterpret_cast<const int2 *>(bid_count + e * 70);
      for (int i0 = 0; i0 < 35; i0 += 7) {
        int2 c[7];
#pragma unroll
        for (int k = 0; k < 7; k++) c[k] = bc[i0 + k];
#pragma unroll
        for (int k = 0; k < 7; k++) {
          if (c[k].x) add(2 * EV_TABLE + 2 * (i0 + k), c[k].x);
          if (c[k].y) add(2 * EV_TABLE + 2 * (i0 + k) + 1, c[k].y);
        }
      }
    }
    if (state) add(2 * EV_TABLE + 70, (long long)bits((uint32_t)(state[e * BRL_STATE_WORDS + W_SC] >> 32), SCH_STEP, 10));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < EV_TOTAL; i += blockDim.x)
    if (h[i]) atomicAdd(reinterpret_cast<unsigned long long *>(&out[i]), (unsigned long long)h[i]);
}

// ---- A9 GAE reverse scan (src/gae.py:20-39): one lane per env, coalesced over envs --------
// The recurrence is serial in t but its INPUTS are not: chunks of GAE_CHUNK steps are loaded up
// front (3 x GAE_CHUNK independent loads in flight per lane) and then scanned from registers.
constexpr int GAE_CHUNK = 32;  // all of a typical rollout's steps (ppo.py:36 num_steps=32) in flight at once
__global__ __launch_bounds__(64) void k_gae(const uint8_t *done, const float *value, const float *reward,
                                            const float *last_val, float gamma, float gamma_lambda, int T, int64_t n,
                                            float *adv, float *tgt) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float gae = 0.0f, next_value = last_val[e];
  for (int t1 = T; t1 > 0; t1 -= GAE_CHUNK) {
    float dn[GAE_CHUNK], vl[GAE_CHUNK], rw[GAE_CHUNK];
#pragma unroll
    for (int k = 0; k < GAE_CHUNK; k++) {
      int t = t1 - 1 - k;
      int64_t i = (int64_t)(t >= 0 ? t : 0) * n + e;
      dn[k] = (float)done[i];
      vl[k] = value[i];
      rw[k] = reward[i];
    }
#pragma unroll
    for (int k = 0; k < GAE_CHUNK; k++) {
      int t = t1 - 1 - k;
      if (t >= 0) {
        int64_t i = (int64_t)t * n + e;
        float nd = 1.0f - dn[k];
        float delta = rw[k] + gamma * next_value * nd - vl[k];  // src/gae.py:28
        gae = delta + gamma_lambda * nd * gae;                   // src/gae.py:29
        adv[i] = gae;
        tgt[i] = gae + vl[k];  // src/gae.py:39
        next_value = vl[k];
      }
    }
  }
}

// ---- A10 _imp_reward -----------------------------------------------------------------------
__global__ void k_imp_reward(const float *a, const float *b, float *out, int64_t n) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  reinterpret_cast<float4 *>(out)[e] = imp_vector(a[e * 4], b[e * 4]);
}

// ---- State attribute access (one thread per table; test / host-mirror path) ------------------
__global__ void k_get_fields(const uint64_t *state, int64_t n, brl_fields F) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const uint64_t *s = state + e * BRL_STATE_WORDS;
  Tbl t;
  t.sc = (uint32_t)s[W_SC]; t.sch = (uint32_t)(s[W_SC] >> 32);
  t.fd = (uint32_t)s[W_FD]; t.t2 = (uint32_t)(s[W_FD] >> 32);
  t.t0 = (uint32_t)s[W_TR]; t.t1 = (uint32_t)(s[W_TR] >> 32);
  t.lut = (uint32_t)s[W_CTR]; t.bctr = (uint32_t)(s[W_CTR] >> 32);
  t.r01 = (uint32_t)s[W_REW]; t.r23 = (uint32_t)(s[W_REW] >> 32);
  uint32_t lb1 = bits(t.sc, SC_LB1, 6);
  if (F.current_player) F.current_player[e] = cur_player(t);
  if (F.terminated) F.terminated[e] = (uint8_t)bits(t.sc, SC_TERM, 1);
  if (F.rewards) reinterpret_cast<float4 *>(F.rewards)[e] = rewards_f32(t);
  if (F.step_count) F.step_count[e] = (int)bits(t.sch, SCH_STEP, 10);
  if (F.turn) F.turn[e] = (int)bits(t.sch, SCH_TURN, 9);
  if (F.dealer) F.dealer[e] = (int)bits(t.sc, SC_DEALER, 2);
  if (F.vul_ns) F.vul_ns[e] = (uint8_t)bits(t.sc, SC_VULNS, 1);
  if (F.vul_ew) F.vul_ew[e] = (uint8_t)bits(t.sc, SC_VULEW, 1);
  if (F.shuffled_players)
    for (int k = 0; k < 4; k++) F.shuffled_players[e * 4 + k] = player_at(t, k);
  if (F.last_bid) F.last_bid[e] = (int)lb1 - 1;
  if (F.last_bidder) F.last_bidder[e] = lb1 ? player_at(t, (int)bits(t.sc, SC_LBSEAT, 2)) : -1;
  if (F.call_x) F.call_x[e] = (uint8_t)bits(t.sc, SC_X, 1);
  if (F.call_xx) F.call_xx[e] = (uint8_t)bits(t.sc, SC_XX, 1);
  if (F.pass_num) F.pass_num[e] = (int)bits(t.sc, SC_PASS, 3);
  for (int d = 0; d < 5; d++) {
    if (F.first_denomination_ns) F.first_denomination_ns[e * 5 + d] = (int)bits(t.fd, 3 * d, 3) - 1;
    if (F.first_denomination_ew) F.first_denomination_ew[e * 5 + d] = (int)bits(t.fd, 15 + 3 * d, 3) - 1;
  }
  if (F.hand) {
    // invert obs index rank*4+suit back to the pgx card id; ascending ids per seat
    for (int seat = 0; seat < 4; seat++) {
      uint64_t m = s[W_HAND + seat] >> 4;
      int k = 0;
      for (int card = 0; card < 52; card++) {
        int suit = card / 13, rank = card % 13;
        int idx = ((rank + 12) % 13) * 4 + (3 - suit);
        if ((m >> idx) & 1ull) F.hand[e * 52 + seat * 13 + (k++)] = card;
      }
    }
  }
  if (F.tricks)
    for (int seat = 0; seat < 4; seat++)
      for (int d = 0; d < 5; d++) F.tricks[e * 20 + seat * 5 + d] = (uint8_t)trick_nibble(t, seat, d);
  if (F.lut_idx) F.lut_idx[e] = (int32_t)t.lut;
  if (F.board_ctr) F.board_ctr[e] = t.bctr;
  if (F.illegal) F.illegal[e] = (uint8_t)bits(t.sc, SC_ILLEGAL, 1);
}

#include "mlp_infer.hpp"   // k_linear16: one bf16 / fp16 layer of the policy MLP (inference)

// =====================================================================================
// C-ABI
// =====================================================================================
struct brl_handle {
  int device;
  int4 *lut_keys;
  int4 *lut_values;
  uint4 *lut_hands;
  int64_t lut_len;
  float *neg_log_n;
  uint64_t seed;
  uint64_t env_offset;
  DevCtx *ctx_dev;  // device mirror of (LUT, seed, env_offset), read by the policy sub-step
  int tables_per_wave;
  int ws;  // 1: wave-specialised fused rollout k_rollout_ws<32,12,1> (default); 0: k_rollout_random<K> (BRL_ROLLOUT_WS=0)
  int fs;  // 1: flag-synchronised k_rollout_fs where it applies (default); 0: always k_rollout_ws (BRL_ROLLOUT_FS=0)
};

static thread_local char g_err[512] = "";

// (shared by every translation unit of the library through abi_common.hpp)
int brl_fail(int code, const char *fmt, const char *detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
  return code;
}
static int fail(int code, const char *fmt, const char *detail) { return brl_fail(code, fmt, detail); }

#include "abi_common.hpp"   // HIP_TRY, NEED

extern "C" const char *brl_last_error(void) { return g_err; }
extern "C" int brl_version(void) { return 5; }   // include/brl_hip.h: the round the exported set last changed in

static inline Rng rng_of(const brl_handle *h) { return Rng{(uint32_t)h->seed, (uint32_t)(h->seed >> 32)}; }
static inline LutRef lut_of(const brl_handle *h) { return LutRef{h->lut_keys, h->lut_values, (uint32_t)h->lut_len, h->lut_hands}; }

// Refresh the device mirror.  Callers have synchronised the device: nothing in flight reads the old contents.
static int sync_ctx(brl_handle *h) {
  DevCtx c{lut_of(h), rng_of(h), h->env_offset};
  HIP_TRY(hipMemcpy(h->ctx_dev, &c, sizeof(c), hipMemcpyHostToDevice));
  return BRL_OK;
}

static int upload_lut(brl_handle *h, const int32_t *keys, const int32_t *values, int64_t len) {
  if (len != h->lut_len) {  // same-size rotation (ppo.py:128: every file holds hash_size rows) reuses the allocations
    if (h->lut_keys) HIP_TRY(hipFree(h->lut_keys));
    if (h->lut_values) HIP_TRY(hipFree(h->lut_values));
    if (h->lut_hands) HIP_TRY(hipFree(h->lut_hands));
    h->lut_keys = nullptr;
    h->lut_values = nullptr;
    h->lut_hands = nullptr;
    h->lut_len = 0;
  }
  if (len > 0) {
    NEED(keys && values, "lut_keys / lut_values are NULL with lut_len > 0");
    NEED(len < (1ll << 32), "lut_len must be < 2^32");
    if (!h->lut_keys) {
      HIP_TRY(hipMalloc(&h->lut_keys, (size_t)len * 16));
      HIP_TRY(hipMalloc(&h->lut_values, (size_t)len * 16));
      HIP_TRY(hipMalloc(&h->lut_hands, (size_t)len * 32));
    }
    HIP_TRY(hipMemcpy(h->lut_keys, keys, (size_t)len * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->lut_values, values, (size_t)len * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_lut_hands, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, h->lut_keys, h->lut_hands, len);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    h->lut_len = len;
  }
  return sync_ctx(h);
}

extern "C" int brl_create(int device, const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len,
                          brl_handle **out) {
  NEED(out != nullptr, "out");
  NEED(lut_len >= 0, "lut_len");
  HIP_TRY(hipSetDevice(device));
  brl_handle *h = (brl_handle *)calloc(1, sizeof(brl_handle));
  NEED(h != nullptr, "out of host memory");
  h->device = device;
  h->tables_per_wave = 4;
  const char *env = getenv("BRL_TABLES_PER_WAVE");
  if (env) {
    int k = atoi(env);
    if (k == 1 || k == 2 || k == 4 || k == 8) h->tables_per_wave = k;
  }
  h->ws = 1;
  const char *ws = getenv("BRL_ROLLOUT_WS");  // "0": the K-tables-per-wave fused rollout (A/B baseline)
  if (ws && ws[0] == '0' && ws[1] == 0) h->ws = 0;
  h->fs = 1;
  const char *fs = getenv("BRL_ROLLOUT_FS");  // "0": the barrier-synchronised k_rollout_ws for every shape (A/B, tests)
  if (fs && fs[0] == '0' && fs[1] == 0) h->fs = 0;
  float tab[BRL_NUM_ACTIONS + 1];
  tab[0] = 0.0f;
  for (int i = 1; i <= BRL_NUM_ACTIONS; i++) tab[i] = (float)(-log((double)i));
  hipError_t e = hipMalloc(&h->neg_log_n, sizeof(tab));
  if (e == hipSuccess) e = hipMemcpy(h->neg_log_n, tab, sizeof(tab), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&h->ctx_dev, sizeof(DevCtx));
  if (e != hipSuccess) {
    free(h);
    return fail(BRL_E_HIP, "brl_create: %s", hipGetErrorString(e));
  }
  int rc = upload_lut(h, lut_keys, lut_values, lut_len);
  if (rc != BRL_OK) {
    (void)hipFree(h->neg_log_n);
    (void)hipFree(h->ctx_dev);
    free(h);
    return rc;
  }
  *out = h;
  return BRL_OK;
}

extern "C" int brl_set_lut(brl_handle *h, const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len) {
  NEED(h != nullptr, "handle");
  NEED(lut_len >= 0, "lut_len");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());  // in-flight kernels may still read the old table
  return upload_lut(h, lut_keys, lut_values, lut_len);
}

extern "C" int brl_destroy(brl_handle *h) {
  if (!h) return BRL_OK;
  (void)hipSetDevice(h->device);
  if (h->lut_keys) (void)hipFree(h->lut_keys);
  if (h->lut_values) (void)hipFree(h->lut_values);
  if (h->lut_hands) (void)hipFree(h->lut_hands);
  if (h->neg_log_n) (void)hipFree(h->neg_log_n);
  if (h->ctx_dev) (void)hipFree(h->ctx_dev);
  free(h);
  return BRL_OK;
}

extern "C" int brl_set_rng(brl_handle *h, uint64_t seed, uint64_t env_offset) {
  NEED(h != nullptr, "handle");
  if (seed == h->seed && env_offset == h->env_offset) return BRL_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());  // launches in flight (and captured graphs being replayed) may still read the old key
  h->seed = seed;
  h->env_offset = env_offset;
  return sync_ctx(h);
}

static inline unsigned wave_grid(int64_t n, int K) {
  int64_t per_block = (int64_t)WAVES_PER_BLOCK * K;
  return (unsigned)((n + per_block - 1) / per_block);
}
static inline unsigned thread_grid(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

#define LAUNCH_K(h, kernel, n, stream, ...)                                                                  \
  do {                                                                                                       \
    hipStream_t _s = (hipStream_t)(stream);                                                                  \
    switch ((h)->tables_per_wave) {                                                                          \
      case 1: hipLaunchKernelGGL(kernel<1>, dim3(wave_grid(n, 1)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      case 2: hipLaunchKernelGGL(kernel<2>, dim3(wave_grid(n, 2)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      case 8: hipLaunchKernelGGL(kernel<8>, dim3(wave_grid(n, 8)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL(kernel<4>, dim3(wave_grid(n, 4)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
    }                                                                                                        \
    HIP_TRY(hipGetLastError());                                                                              \
  } while (0)

#define COMMON(h, n)                 \
  NEED((h) != nullptr, "handle");    \
  NEED((n) >= 0, "n");               \
  if ((n) == 0) return BRL_OK;       \
  HIP_TRY(hipSetDevice((h)->device))

extern "C" int brl_init_random(brl_handle *h, uint64_t *state, int64_t n, uint32_t board_ctr0, void *stream) {
  COMMON(h, n);
  NEED(state != nullptr, "state");
  if (h->lut_len == 0) return fail(BRL_E_NOLUT, "brl_init_random needs a LUT%s", "");
  LAUNCH_K(h, k_init_random, n, stream, state, n, rng_of(h), h->env_offset, lut_of(h), board_ctr0);
  return BRL_OK;
}

extern "C" int brl_init_from_deals(brl_handle *h, uint64_t *state, int64_t n, const int32_t *hand,
                                   const int32_t *dealer, const uint8_t *vul_ns, const uint8_t *vul_ew,
                                   const int32_t *shuffled_players, const uint8_t *tricks, void *stream) {
  COMMON(h, n);
  NEED(state && hand && dealer && vul_ns && vul_ew && shuffled_players && tricks, "NULL input array");
  hipLaunchKernelGGL(k_init_explicit, dim3(thread_grid(n, 128)), dim3(128), 0, (hipStream_t)stream, state, n, hand,
                     dealer, vul_ns, vul_ew, shuffled_players, tricks);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                        const int32_t *action, int autoreset, uint8_t *obs, uint8_t *mask, float *rewards,
                        uint8_t *terminated, int32_t *current_player, void *stream) {
  COMMON(h, n);
  NEED(state_in && state_out && action, "NULL state / action");
  if (autoreset && h->lut_len == 0) return fail(BRL_E_NOLUT, "auto-reset needs a LUT%s", "");
  StepOut o{obs, mask, rewards, terminated, current_player};
  LAUNCH_K(h, k_step, n, stream, state_in, state_out, n, action, autoreset, rng_of(h), h->env_offset, lut_of(h), o);
  return BRL_OK;
}

extern "C" int brl_observe(brl_handle *h, const uint64_t *state, int64_t n, const int32_t *player_id, uint8_t *obs,
                           uint8_t *mask, void *stream) {
  COMMON(h, n);
  NEED(state != nullptr, "state");
  LAUNCH_K(h, k_observe, n, stream, state, n, player_id, obs, mask);
  return BRL_OK;
}

extern "C" int brl_get_fields(brl_handle *h, const uint64_t *state, int64_t n, const brl_fields *out, void *stream) {
  COMMON(h, n);
  NEED(state && out, "state / out");
  hipLaunchKernelGGL(k_get_fields, dim3(thread_grid(n, 128)), dim3(128), 0, (hipStream_t)stream, state, n, *out);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

// The fused rollout kernels store 16 bytes per lane (observation pieces, mask chunks, four tables of a scalar column) and
// 4 bytes of `done` at a time: every output array must start on such a boundary (any hipMalloc / torch allocation does;
// an odd view into one does not).
static inline bool aligned_to(const void *p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) == 0; }
static bool transition_aligned(const brl_transition *o, const void *last_obs, const void *last_mask, const void *adv, const void *tgt) {
  return aligned_to(o->obs, 16) && aligned_to(o->legal_action_mask, 16) && aligned_to(o->action, 16) && aligned_to(o->value, 16) &&
         aligned_to(o->reward, 16) && aligned_to(o->log_prob, 16) && aligned_to(o->done, 4) && aligned_to(last_obs, 16) &&
         aligned_to(last_mask, 16) && aligned_to(adv, 16) && aligned_to(tgt, 16);
}

extern "C" int brl_rollout_random(brl_handle *h, uint64_t *state, int64_t n, int num_steps, int substeps,
                                  uint32_t draw_base, float reward_scale, const brl_transition *out,
                                  uint8_t *last_obs, uint8_t *last_mask, int64_t *terminated_count, void *stream) {
  COMMON(h, n);
  NEED(state && out, "state / out");
  NEED(aligned_to(state, 16) && transition_aligned(out, last_obs, last_mask, nullptr, nullptr),
       "output arrays must be 16-byte aligned (done: 4-byte)");
  NEED(num_steps >= 0, "num_steps");
  NEED(substeps >= 1 && substeps <= 16, "substeps");
  if (h->lut_len == 0) return fail(BRL_E_NOLUT, "brl_rollout_random auto-resets and needs a LUT%s", "");
  RolloutArgs A;
  A.state = state; A.n = n; A.T = num_steps; A.substeps = substeps; A.draw_base = draw_base;
  A.reward_scale = reward_scale; A.g = rng_of(h); A.env_offset = h->env_offset; A.lut = lut_of(h);
  A.neg_log_n = h->neg_log_n; A.out = *out; A.terminated_count = (unsigned long long *)terminated_count;
  A.last_obs = last_obs; A.last_mask = last_mask;
  A.gae_last_val = nullptr; A.gae_gamma = 0.0f; A.gae_gamma_lambda = 0.0f; A.gae_adv = nullptr; A.gae_tgt = nullptr;
#ifdef BRL_TIMING  // experiment switches (some of them change the outputs): timing builds only (scripts/timing.py)
  A.debug = getenv("BRL_DEBUG") ? atoi(getenv("BRL_DEBUG")) : 0;
#else
  A.debug = 0;
#endif
  // the wave-specialised kernel serves a macro-step that spans <= 2 command batches; longer ones (and BRL_ROLLOUT_WS=0)
  // take the K-tables-per-wave kernel
  const bool all_cols = out->obs && out->legal_action_mask && out->done && out->action && out->value && out->reward && out->log_prob;
  if (h->ws && h->fs && substeps == 1 && n % FS_TPB == 0 && all_cols) {
    // the BASELINE shape: flag-synchronised kernel (rollout_fs.hpp).  It holds a whole launch's commands in LDS (<= 40 steps):
    // a longer rollout is the same thing in pieces — every piece continues from the state, the draw counter and the
    // terminated count the one before left (the pieces are of near-equal length: 64 -> 32 + 32, 100 -> 34 + 33 + 33)
    const int pieces = (num_steps + FS_MAX_TOTAL - 1) / FS_MAX_TOTAL;
    int t0 = 0;
    for (int i = 0; i < pieces || (pieces == 0 && i == 0); i++) {
      const int len = (pieces > 0) ? (num_steps - t0 + (pieces - i) - 1) / (pieces - i) : 0;
      const int64_t rows = (int64_t)t0 * n;
      RolloutArgs P = A;
      P.T = len;
      P.draw_base = draw_base + (uint32_t)t0;
      P.out.obs = out->obs + rows * BRL_OBS_SIZE;
      P.out.legal_action_mask = out->legal_action_mask + rows * BRL_NUM_ACTIONS;
      P.out.done = out->done + rows;
      P.out.action = out->action + rows;
      P.out.value = out->value + rows;
      P.out.reward = out->reward + rows;
      P.out.log_prob = out->log_prob + rows;
      const bool last = (i + 1 >= pieces);
      P.last_obs = last ? last_obs : nullptr;
      P.last_mask = last ? last_mask : nullptr;
      hipLaunchKernelGGL(k_rollout_fs, dim3((unsigned)(n / FS_TPB)), dim3(FS_NW * 64), 0, (hipStream_t)stream, P);
      t0 += len;
    }
  } else if (h->ws && substeps <= WS_BATCH) {
    hipLaunchKernelGGL((k_rollout_ws<32, 12, 1>), dim3((unsigned)((n + 31) / 32)), dim3(12 * 64), 0, (hipStream_t)stream, A);
  } else {
    LAUNCH_K(h, k_rollout_random, n, stream, A);
    return BRL_OK;
  }
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_rollout_random_gae(brl_handle *h, uint64_t *state, int64_t n, int num_steps, uint32_t draw_base,
                                      float reward_scale, const brl_transition *out, uint8_t *last_obs, uint8_t *last_mask,
                                      int64_t *terminated_count, const float *last_val, float gamma, float gamma_lambda,
                                      float *advantages, float *targets, void *stream) {
  COMMON(h, n);
  NEED(state && out && last_val && advantages && targets, "state / out / last_val / advantages / targets");
  NEED(num_steps >= 1, "num_steps");
  NEED(out->done && out->value && out->reward, "the done / value / reward columns");
  NEED(aligned_to(state, 16) && transition_aligned(out, last_obs, last_mask, advantages, targets),
       "output arrays must be 16-byte aligned (done: 4-byte)");
  if (h->lut_len == 0) return fail(BRL_E_NOLUT, "brl_rollout_random_gae auto-resets and needs a LUT%s", "");
  const bool all_cols = out->obs && out->legal_action_mask && out->action && out->log_prob;
  if (!(h->ws && h->fs && num_steps <= FS_MAX_TOTAL && n % FS_TPB == 0 && all_cols)) {
    // shapes the one-launch kernel does not serve (more than 40 steps, n not a multiple of 32, a column left out): the same
    // results from the rollout launch(es) followed by the scan of their columns
    const int rc = brl_rollout_random(h, state, n, num_steps, 1, draw_base, reward_scale, out, last_obs, last_mask,
                                      terminated_count, stream);
    if (rc != BRL_OK) return rc;
    return brl_gae(h, out->done, out->value, out->reward, last_val, gamma, gamma_lambda, num_steps, n, advantages, targets, stream);
  }
  RolloutArgs A;
  A.state = state; A.n = n; A.T = num_steps; A.substeps = 1; A.draw_base = draw_base;
  A.reward_scale = reward_scale; A.g = rng_of(h); A.env_offset = h->env_offset; A.lut = lut_of(h);
  A.neg_log_n = h->neg_log_n; A.out = *out; A.terminated_count = (unsigned long long *)terminated_count;
  A.last_obs = last_obs; A.last_mask = last_mask; A.debug = 0;
  A.gae_last_val = last_val; A.gae_gamma = gamma; A.gae_gamma_lambda = gamma_lambda; A.gae_adv = advantages; A.gae_tgt = targets;
  hipLaunchKernelGGL(k_rollout_fs, dim3((unsigned)(n / FS_TPB)), dim3(FS_NW * 64), 0, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

static int policy_step_impl(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                            const float *logits, int64_t logits_stride, int mode, const uint32_t *draw_dev, uint32_t draw,
                            int autoreset,
                            int32_t *action, float *log_prob, uint8_t *obs, uint8_t *mask, float *rewards_acc,
                            uint8_t *terminated_acc, int32_t *current_player, void *stream, const brl_macro_ext *ext = nullptr) {
  COMMON(h, n);
  const bool heads = ext != nullptr && (ext->head_h != nullptr || ext->head_part != nullptr);
  NEED(state_in && state_out && (logits || heads), "NULL state / logits");
  NEED(mode >= 0 && mode <= 3, "mode");
  NEED(heads || logits_stride >= BRL_NUM_ACTIONS, "logits_stride");
  if (autoreset && h->lut_len == 0) return fail(BRL_E_NOLUT, "auto-reset needs a LUT%s", "");
  PolicyArgs A;
  A.state_in = state_in; A.state_out = state_out; A.n = n; A.logits = logits; A.mode = mode; A.draw = draw;
  A.logits_stride = logits_stride;
  A.draw_dev = draw_dev;
  A.autoreset = autoreset; A.ctx = h->ctx_dev;
  A.action = action; A.log_prob = log_prob;
  A.o = StepOut{obs, mask, rewards_acc, terminated_acc, current_player};
  memset(&A.x, 0, sizeof(A.x));
  if (ext != nullptr) {
    A.x = *ext;
    NEED(!ext->value_out || heads || (ext->value_in && ext->value_stride >= 1), "ext: value_in / value_stride");
    NEED(!ext->last || !ext->reward_out || (ext->actor && rewards_acc && ext->reward_scale != 0.0f), "ext: reward_out needs actor, rewards_acc, reward_scale");
    NEED(!ext->last || !(ext->done_out || ext->terminated_count) || terminated_acc, "ext: done_out / terminated_count need terminated_acc");
    NEED(!ext->obs_cast || (ext->obs_fmt >= 0 && ext->obs_fmt <= 2), "ext: obs_fmt");
    NEED(ext->in_fmt >= 0 && ext->in_fmt <= 2, "ext: in_fmt");
    if (ext->head_part != nullptr) {
      NEED(ext->head_b && ext->head_nparts >= 1 && ext->head_part_ld >= BRL_NUM_ACTIONS + 1 && ext->head_part_stride >= n * ext->head_part_ld,
           "ext: head_b / head_nparts / head_part_ld (>= 39) / head_part_stride (>= n * head_part_ld)");
      NEED(h->tables_per_wave == 4, "ext: head_part needs BRL_TABLES_PER_WAVE=4 (the head tile is a workgroup's 16 tables)");
      hipLaunchKernelGGL((k_policy_step<4, 2>), dim3(wave_grid(n, 4)), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, A);
      HIP_TRY(hipGetLastError());
      return BRL_OK;
    }
    if (heads) {
      NEED(ext->head_w && ext->head_b && (ext->head_fmt == 1 || ext->head_fmt == 2), "ext: head_w / head_b / head_fmt (1 bf16, 2 fp16)");
      NEED(ext->head_hidden > 0 && ext->head_hidden % 32 == 0 && ext->head_ldh >= ext->head_hidden && ext->head_ldh % 8 == 0,
           "ext: head_hidden (a multiple of 32) / head_ldh (a multiple of 8)");
      NEED(h->tables_per_wave == 4, "ext: head_h needs BRL_TABLES_PER_WAVE=4 (the head tile is a workgroup's 16 tables)");
      hipLaunchKernelGGL((k_policy_step<4, 1>), dim3(wave_grid(n, 4)), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, A);
      HIP_TRY(hipGetLastError());
      return BRL_OK;
    }
  }
  LAUNCH_K(h, k_policy_step, n, stream, A);
  return BRL_OK;
}

extern "C" int brl_policy_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                               const float *logits, int mode, uint32_t draw, int autoreset, int32_t *action,
                               float *log_prob, uint8_t *obs, uint8_t *mask, float *rewards_acc,
                               uint8_t *terminated_acc, int32_t *current_player, void *stream) {
  return policy_step_impl(h, state_in, state_out, n, logits, BRL_NUM_ACTIONS, mode, nullptr, draw, autoreset, action,
                          log_prob, obs, mask, rewards_acc, terminated_acc, current_player, stream);
}

extern "C" int brl_policy_step_at(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                                  const float *logits, int64_t logits_stride, int mode, const uint32_t *draw_base,
                                  uint32_t draw_offset, int autoreset, int32_t *action, float *log_prob, uint8_t *obs,
                                  uint8_t *mask, float *rewards_acc, uint8_t *terminated_acc, int32_t *current_player,
                                  void *stream) {
  return policy_step_impl(h, state_in, state_out, n, logits, logits_stride, mode, draw_base, draw_offset, autoreset,
                          action, log_prob, obs, mask, rewards_acc, terminated_acc, current_player, stream);
}

extern "C" int brl_policy_step_ex(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                                  const float *logits, int64_t logits_stride, int mode, const uint32_t *draw_base,
                                  uint32_t draw_offset, int autoreset, int32_t *action, float *log_prob, uint8_t *obs,
                                  uint8_t *mask, float *rewards_acc, uint8_t *terminated_acc, int32_t *current_player,
                                  const brl_macro_ext *ext, void *stream) {
  return policy_step_impl(h, state_in, state_out, n, logits, logits_stride, mode, draw_base, draw_offset, autoreset,
                          action, log_prob, obs, mask, rewards_acc, terminated_acc, current_player, stream, ext);
}

// observation bytes (0/1) -> the network's input dtype: 16 bytes in, 16 elements out per thread
// (src/roll_out.py:75 `last_obs.astype(jnp.float32)`; torch's generic bool->bf16 copy takes 15 us for 3.9 MB)
template <int FMT>  // 0: f32, 1: bf16 (0x3F80), 2: f16 (0x3C00)
__device__ __forceinline__ void obs_cast16(const uint4 v, void *out, int64_t i) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  if (FMT == 0) {
    float4 *o = reinterpret_cast<float4 *>(out) + 4 * i;
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = make_float4((w[k] & 1u) ? 1.0f : 0.0f, (w[k] & 0x100u) ? 1.0f : 0.0f, (w[k] & 0x10000u) ? 1.0f : 0.0f,
                         (w[k] & 0x1000000u) ? 1.0f : 0.0f);
  } else {
    const uint32_t one = (FMT == 1) ? 0x3F80u : 0x3C00u;
    uint4 *o = reinterpret_cast<uint4 *>(out) + 2 * i;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // bytes (b0,b1,b2,b3) of a dword -> halves (b0,b1) and (b2,b3)
      h[2 * k] = ((w[k] & 1u) ? one : 0u) | ((w[k] & 0x100u) ? (one << 16) : 0u);
      h[2 * k + 1] = ((w[k] & 0x10000u) ? one : 0u) | ((w[k] & 0x1000000u) ? (one << 16) : 0u);
    }
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
  }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_obs_cast(const uint4 *in, void *out, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  obs_cast16<FMT>(in[i], out, i);
}

// the same for the rows rows[0..m) of `in` only (out row r = in row rows[r]): the forwards of an evaluator run on the boards
// that are still playing
template <int FMT>
__global__ __launch_bounds__(256) void k_obs_cast_rows(const uint4 *in, const int64_t *rows, void *out, int64_t m16) {
  constexpr int PER_ROW = BRL_OBS_SIZE / 16;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m16) return;
  const int64_t r = i / PER_ROW;
  obs_cast16<FMT>(in[rows[r] * PER_ROW + (i - r * PER_ROW)], out, i);
}

extern "C" int brl_obs_cast(brl_handle *h, const uint8_t *obs, int64_t n, void *out, int fmt, void *stream) {
  COMMON(h, n);
  NEED(obs && out, "NULL obs / out");
  NEED(fmt >= 0 && fmt <= 2, "fmt");
  const int64_t n16 = n * (BRL_OBS_SIZE / 16);
  const dim3 grid((unsigned)((n16 + 255) / 256)), block(256);
  if (fmt == 0) hipLaunchKernelGGL(k_obs_cast<0>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  else if (fmt == 1) hipLaunchKernelGGL(k_obs_cast<1>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  else hipLaunchKernelGGL(k_obs_cast<2>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_obs_cast_rows(brl_handle *h, const uint8_t *obs, const int64_t *rows, int64_t m, void *out, int fmt,
                                 void *stream) {
  COMMON(h, m);
  NEED(obs && rows && out, "NULL obs / rows / out");
  NEED(fmt >= 0 && fmt <= 2, "fmt");
  const int64_t m16 = m * (BRL_OBS_SIZE / 16);
  const dim3 grid((unsigned)((m16 + 255) / 256)), block(256);
  if (fmt == 0) hipLaunchKernelGGL(k_obs_cast_rows<0>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  else if (fmt == 1) hipLaunchKernelGGL(k_obs_cast_rows<1>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  else hipLaunchKernelGGL(k_obs_cast_rows<2>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

// The loop condition of the evaluators, `~state.terminated.all()` (src/evaluation.py:120-122), as data: how many boards are
// finished, and the indices of the others in ascending order at the front of `live` (the entries behind them are left as they
// are: the caller initialises the list with 0..n-1 once, so they stay valid board indices).  One workgroup: a thread counts
// its run of boards, the runs' offsets come from a scan in LDS — deterministic order, no atomics.
__global__ __launch_bounds__(1024) void k_live_index(const uint8_t *terminated, int64_t n, int64_t *live, int64_t *finished, int64_t tag) {
  __shared__ int64_t part[1024];
  const int tid = (int)threadIdx.x;
  // (a thread's run of boards, rounded up to 8 so that the flags can be read 8 at a time: one load instead of a chain of byte loads)
  const int64_t per = ((n + 1023) / 1024 + 7) / 8 * 8, a0 = (int64_t)tid * per, a = (a0 < n) ? a0 : n, b = (a + per < n) ? a + per : n;
  const bool wide = (reinterpret_cast<uintptr_t>(terminated) & 7u) == 0;
  int64_t c = 0;
  for (int64_t i = a; i < b; i += 8) {
    if (wide && i + 8 <= b) {
      const uint64_t f = *reinterpret_cast<const uint64_t *>(terminated + i);
      // bytes are 0 / 1 (bool) or any non-zero: count the zero bytes
      uint64_t nz = f | (f >> 4); nz |= nz >> 2; nz |= nz >> 1; nz &= 0x0101010101010101ull;
      c += 8 - __popcll(nz);
    } else {
      for (int64_t k = i; k < b && k < i + 8; k++) c += terminated[k] ? 0 : 1;
    }
  }
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan
    const int64_t v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int64_t pos = part[tid] - c;
  if (live != nullptr)
    for (int64_t i = a; i < b; i++)
      if (!terminated[i]) live[pos++] = i;
  if (tid == 1023 && finished != nullptr) {
    const int64_t count = n - part[1023];
    // tag >= 0: the word is read by the HOST while the stream runs on (pinned memory, no event): tag and count arrive as one
    // 64-bit store, released at system scope
    if (tag >= 0) __hip_atomic_store(finished, (tag << 32) | count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else *finished = count;
  }
}

extern "C" int brl_live_index(brl_handle *h, const uint8_t *terminated, int64_t n, int64_t *live, int64_t *finished, int64_t tag,
                              void *stream) {
  COMMON(h, n);
  NEED(terminated && (live || finished), "NULL terminated / outputs");
  NEED(tag < ((int64_t)1 << 31) && n < ((int64_t)1 << 32), "tag below 2^31, n below 2^32");
  if (finished != nullptr) {
    // `finished` may be PINNED HOST memory: the launch then stores the count where the host reads it (behind an event) — no copy
    // launch, no copy engine between two iterations of an evaluator.  Translated to the address the device uses.
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, finished) == hipSuccess && at.type == hipMemoryTypeHost) {
      void *dp = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dp, finished, 0));
      finished = (int64_t *)dp;
    } else {
      (void)hipGetLastError();   // (an address the runtime does not know: left as it is)
    }
  }
  hipLaunchKernelGGL(k_live_index, dim3(1), dim3(1024), 0, (hipStream_t)stream, terminated, n, live, finished, tag);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

static int lin16_attr(int fmt) {
  static bool done[3] = {false, false, false};
  if (!done[fmt]) {   // 144 KB of dynamic LDS: above the default 64 KB limit
    if (fmt == 1) HIP_TRY(hipFuncSetAttribute((const void *)lin16::k_linear16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::LDS_BYTES));
    else HIP_TRY(hipFuncSetAttribute((const void *)lin16::k_linear16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::LDS_BYTES));
    done[fmt] = true;
  }
  return BRL_OK;
}

static int lin16_store_mode() {
  static int store_mode = -1;
  if (store_mode < 0) {
    // y leaves write-through by default: measured in the bf16 graph rollout, 8192 tables: 12.47 ms against 13.30 (plain
    // stores: the dirty lines are written back when the kernel ends) and 13.01 (non-temporal); BRL_LIN16_STORE=0/1/2 for A/B
    const char *e = getenv("BRL_LIN16_STORE");
    store_mode = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : 2;
  }
  return store_mode;
}

#ifdef LIN16_TIMING   // scripts/time_linear16.py --stamps: 4 shader-clock stamps per workgroup
static unsigned long long *g_lin16_dbg = nullptr;
extern "C" void brl_lin16_set_dbg(void *p) { g_lin16_dbg = (unsigned long long *)p; }
#endif
static int linear_act_impl(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                           int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w, int64_t ld_head_w,
                           int n_heads, float *head_part, int64_t head_part_ld, int64_t head_part_stride, void *stream) {
  COMMON(h, m);
  NEED(x && w && (y || head_part), "NULL x / w / y");
  if (head_part) {
    NEED(head_w && n_heads >= 1 && n_heads <= 48 && ld_head_w >= n_out && ld_head_w % 8 == 0 && (((uintptr_t)head_w) & 15) == 0,
         "head_w / n_heads (<= 48) / ld_head_w");
    NEED(head_part_ld >= ((n_heads + 3) & ~3) && head_part_ld % 4 == 0 && head_part_stride >= m * head_part_ld && head_part_stride % 4 == 0
         && (((uintptr_t)head_part) & 15) == 0, "head_part (16-byte aligned) / head_part_ld (% 4, >= n_heads rounded up to 4) / head_part_stride");
  }
  if (!y) ldy = n_out;
  NEED(fmt == 1 || fmt == 2, "fmt (1 = bf16, 2 = fp16)");
  NEED(n_out > 0 && n_out % lin16::BN == 0, "n_out % 128");
  NEED(k >= 8 && k % 8 == 0, "k % 8");
  NEED(ldx >= k && ldw >= k && ldy >= n_out && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "ldx / ldw / ldy");
  NEED((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "x / w / y not 16-byte aligned");
  NEED(m * ldx * 2 < ((int64_t)1 << 32) && (int64_t)n_out * ldw * 2 < ((int64_t)1 << 32), "operand larger than 4 GB");
  NEED(m <= (int64_t)1 << 30, "m");
  if (int rc = lin16_attr(fmt)) return rc;
  lin16::Args A;
  memset(&A, 0, sizeof(A));
  A.x = (const uint16_t *)x; A.ldx = ldx;
  A.w = (const uint16_t *)w; A.ldw = ldw;
  A.bias = bias;
  A.y = (uint16_t *)y; A.ldy = ldy;
  A.M = (int)m; A.N = n_out; A.K = k;
  A.relu = relu;
  A.head_w = (const uint16_t *)head_w; A.ld_head_w = ld_head_w; A.n_heads = n_heads;
  A.head_part = head_part; A.head_part_ld = head_part_ld; A.head_part_stride = head_part_stride;
  A.store_mode = lin16_store_mode();
#ifdef LIN16_TIMING
  A.dbg = g_lin16_dbg;
#endif
  const int tiles = (int)((m + lin16::BM - 1) / lin16::BM) * (n_out / lin16::BN);
  if (fmt == 1) hipLaunchKernelGGL(lin16::k_linear16<1>, dim3(tiles), dim3(lin16::THREADS), lin16::LDS_BYTES, (hipStream_t)stream, A);
  else hipLaunchKernelGGL(lin16::k_linear16<2>, dim3(tiles), dim3(lin16::THREADS), lin16::LDS_BYTES, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_linear_act(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                            int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, void *stream) {
  NEED(y != nullptr, "NULL y");
  return linear_act_impl(h, x, ldx, w, ldw, bias, y, ldy, m, n_out, k, relu, fmt, nullptr, 0, 0, nullptr, 0, 0, stream);
}

extern "C" int brl_linear_act_heads(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias,
                                  void *y, int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w,
                                  int64_t ld_head_w, int n_heads, float *head_part, int64_t head_part_ld,
                                  int64_t head_part_stride, void *stream) {
  NEED(head_part != nullptr, "NULL head_part");
  return linear_act_impl(h, x, ldx, w, ldw, bias, y, ldy, m, n_out, k, relu, fmt, head_w, ld_head_w, n_heads, head_part,
                         head_part_ld, head_part_stride, stream);
}

extern "C" int brl_gae(brl_handle *h, const uint8_t *done, const float *value, const float *reward,
                       const float *last_val, float gamma, float gamma_lambda, int T, int64_t n, float *advantages,
                       float *targets, void *stream) {
  COMMON(h, n);
  NEED(done && value && reward && last_val && advantages && targets, "NULL array");
  NEED(T >= 0, "T");
  hipLaunchKernelGGL(k_gae, dim3(thread_grid(n, 64)), dim3(64), 0, (hipStream_t)stream, done, value, reward, last_val,
                     gamma, gamma_lambda, T, n, advantages, targets);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_imp_reward(brl_handle *h, const float *a, const float *b, float *out, int64_t n, void *stream) {
  COMMON(h, n);
  NEED(a && b && out, "NULL array");
  hipLaunchKernelGGL(k_imp_reward, dim3(thread_grid(n, 128)), dim3(128), 0, (hipStream_t)stream, a, b, out, n);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

static bool table_info_ok(const brl_table_info *t) {
  return t && t->terminated && t->rewards && t->last_bid && t->last_bidder && t->call_x && t->call_xx;
}

static int eval_step_impl(brl_handle *h, EvalArgs &A, void *stream) {
  LAUNCH_K(h, k_eval_step, A.n, stream, A);
  return BRL_OK;
}

extern "C" int brl_duplicate_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                                  const int32_t *action, const brl_table_info *table_a,
                                  const brl_table_info *table_b, uint8_t *obs, uint8_t *mask, float *rewards,
                                  uint8_t *terminated, int32_t *current_player, void *stream) {
  COMMON(h, n);
  NEED(state_in && state_out && action, "NULL state / action");
  NEED(table_info_ok(table_a) && table_info_ok(table_b), "table_a / table_b has NULL members");
  EvalArgs A{};
  A.state_in = state_in; A.state_out = state_out; A.n = n; A.action = action; A.duplicate = 1;
  A.TA = *table_a; A.TB = *table_b;
  A.o = StepOut{obs, mask, rewards, terminated, current_player};
  A.acting_team = -1;
  return eval_step_impl(h, A, stream);
}

extern "C" int brl_eval_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                             const float *logits_team1, int64_t stride1, const float *logits_team2, int64_t stride2,
                             const brl_table_info *table_a, const brl_table_info *table_b, const brl_eval_stats *stats,
                             int bid_set, float *cum_return, float *rewards_sum, int32_t *action_out, uint8_t *obs,
                             uint8_t *mask, float *rewards, uint8_t *terminated, int32_t *current_player, void *stream) {
  COMMON(h, n);
  NEED(state_in && state_out && logits_team1 && logits_team2, "NULL state / logits");
  NEED(stride1 >= BRL_NUM_ACTIONS && stride2 >= BRL_NUM_ACTIONS, "logits stride");
  NEED((table_a == nullptr) == (table_b == nullptr), "table_a and table_b go together");
  if (table_a) NEED(table_info_ok(table_a) && table_info_ok(table_b), "table_a / table_b has NULL members");
  EvalArgs A{};
  A.state_in = state_in; A.state_out = state_out; A.n = n;
  A.logits1 = logits_team1; A.logits2 = logits_team2; A.stride1 = stride1; A.stride2 = stride2;
  A.duplicate = table_a != nullptr;
  if (table_a) { A.TA = *table_a; A.TB = *table_b; }
  if (stats) A.S = *stats;
  A.bid_set = bid_set; A.cum_return = cum_return; A.rewards_sum = rewards_sum; A.action_out = action_out;
  A.o = StepOut{obs, mask, rewards, terminated, current_player};
  A.acting_team = -1;
  return eval_step_impl(h, A, stream);
}

extern "C" int brl_eval_step_team(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n, const float *logits,
                                  int64_t stride, int acting_team, const brl_table_info *table_a,
                                  const brl_table_info *table_b, const brl_eval_stats *stats, int bid_set, float *cum_return,
                                  float *rewards_sum, int32_t *action_out, uint8_t *obs, uint8_t *mask, float *rewards,
                                  uint8_t *terminated, int32_t *current_player, float *obs_f32, void *stream) {
  COMMON(h, n);
  NEED(state_in && state_out && logits, "NULL state / logits");
  NEED(stride >= BRL_NUM_ACTIONS, "logits stride");
  NEED(acting_team == 0 || acting_team == 1, "acting_team");
  NEED((((uintptr_t)obs_f32) & 15) == 0, "obs_f32 not 16-byte aligned");
  NEED((table_a == nullptr) == (table_b == nullptr), "table_a and table_b go together");
  if (table_a) NEED(table_info_ok(table_a) && table_info_ok(table_b), "table_a / table_b has NULL members");
  EvalArgs A{};
  A.state_in = state_in; A.state_out = state_out; A.n = n;
  A.logits1 = logits; A.logits2 = logits; A.stride1 = stride; A.stride2 = stride;
  A.duplicate = table_a != nullptr;
  if (table_a) { A.TA = *table_a; A.TB = *table_b; }
  if (stats) A.S = *stats;
  A.bid_set = bid_set; A.cum_return = cum_return; A.rewards_sum = rewards_sum; A.action_out = action_out;
  A.o = StepOut{obs, mask, rewards, terminated, current_player};
  A.acting_team = acting_team;
  A.obs_f32 = obs_f32;
  return eval_step_impl(h, A, stream);
}

extern "C" int brl_eval_reduce(brl_handle *h, int64_t n, const brl_table_info *table_a, const brl_table_info *table_b,
                               const int32_t *bid_count, const uint64_t *state, int64_t *out, void *stream) {
  COMMON(h, n);
  NEED(table_info_ok(table_a) && out, "table_a / out");
  if (table_b) NEED(table_info_ok(table_b), "table_b has NULL members");
  HIP_TRY(hipMemsetAsync(out, 0, sizeof(int64_t) * EV_TOTAL, (hipStream_t)stream));
  hipLaunchKernelGGL(k_eval_reduce, dim3(thread_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, *table_a,
                     table_b ? *table_b : *table_a, table_b ? 1 : 0, bid_count, state, (long long *)out);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
