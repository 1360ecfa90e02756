// brl_eval.hip — translation unit of libbrl_hip.so: one iteration of the evaluators' loops (src/evaluation.py:87-204, 229-1032:
// greedy call of the team to act, duplicate_step / env.step, the step log, return accumulators), their end-of-run histograms and
// the live-board index (include/brl_hip.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "handle.hpp"
#include "imp.hpp"
#include "policy_common.hpp"

// ---- A12 duplicate_step ---------------------------------------------------------------------
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

// =====================================================================================
// C-ABI
// =====================================================================================
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
