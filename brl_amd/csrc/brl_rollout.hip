// brl_rollout.hip — translation unit of libbrl_hip.so: the fused random-policy rollouts (k_rollout_fs — the BASELINE step —,
// k_rollout_ws, k_rollout_random<K>: src/roll_out.py:49-108 with the uniform-random masked policy) and the policy sub-step of the
// MLP-in-the-loop rollout (k_policy_step: sample / arg-max + auto_reset(step) + macro-step bookkeeping, src/utils.py:69-128).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "handle.hpp"
#include "policy_common.hpp"

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

// =====================================================================================
// C-ABI
// =====================================================================================
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
