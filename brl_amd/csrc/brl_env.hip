// brl_env.hip — translation unit of libbrl_hip.so: the handle (LUT upload, RNG key), the environment's own entry points
// (init / step / observe / State fields: pgx.bridge_bidding.{init,step,observe}), calc_gae's scan and the IMP conversion
// (include/brl_hip.h).  The fused rollouts and the policy sub-step live in brl_rollout.hip, the evaluators' step in brl_eval.hip,
// the 16-bit inference layer in brl_infer16.hip, the PPO update in brl_ppo.hip, the step's own fp32 GEMM in brl_mlp_gemm.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC (see brl_amd/build.py)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "handle.hpp"
#include "imp.hpp"

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

// =====================================================================================
// C-ABI
// =====================================================================================
static thread_local char g_err[512] = "";

// (shared by every translation unit of the library through abi_common.hpp)
int brl_fail(int code, const char *fmt, const char *detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
  return code;
}


extern "C" const char *brl_last_error(void) { return g_err; }
extern "C" int brl_version(void) { return 6; }   // include/brl_hip.h: the round the exported set last changed in
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
