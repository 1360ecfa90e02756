// rollout_fs.hpp — k_rollout_fs: the fused random-policy rollout (A7, src/roll_out.py:63-107 with a uniform random legal policy),
// flag-synchronised.  Included by brl_rollout.hip after k_rollout_ws (shares RolloutArgs, the command format and LutRef).
//
// Same roles as k_rollout_ws — one workgroup owns 32 consecutive tables; a LOGIC wave runs the per-table dependency chain,
// a LOADER fetches boards, a SCORER writes the scalar Transition columns, EMIT waves write observations — but NO workgroup
// barrier between the prologue's and the one before the state write-back:
//   * the whole launch's commands fit in LDS (one 16-byte command per table and slot, <= 41 slots = 21 KB), so the logic
//     wave never waits for a follower: it posts slot s and publishes `posted = s + 1` with a plain LDS store (LDS
//     operations of one wave are performed in order, so a reader that sees the counter sees the commands);
//   * every follower consumes slot s as soon as it is posted, at its own pace (it polls `posted` only when it has caught up);
//   * <= 11 boards per table can be dealt in <= 40 sub-steps, so the 12-entry board ring never wraps: the loader fetches
//     all 12 boards of every table up front (two per pass, one in each half of the wave), publishes `ring_count`, and
//     then turns into the wave that writes the legal-mask rows;
//   * the logic wave runs nothing but the chain (draw -> call -> scalars, ~90 instructions per sub-step): what the
//     followers need besides (legal mask, n_legal, history bit, observer seat) is recomputed from its raw posts by a
//     PREP wave, two slots per pass;
//   * scoring is split by what is serial: SCORER A (lane = table) follows the commands slot by slot (first denominations,
//     who acted, which boards ended) and queues the finished boards; SCORER B scores them one lane per board and writes
//     the scalar columns 8 slots at a time.  B is done ~6 k cycles after the logic wave, so the optional calc_gae scan
//     (brl_rollout_random_gae: on the logic wave, once B has every slot's reward in LDS) ends before the emit waves do.
// Why (profiles/r02/r02_rollout_experiments.txt §5): with the output going to HBM the launch is  T = (time until the
// stores start and are never starved) + (store time of 140 MB).  A store-only probe paced like this hand-off
// (scripts/micro/store_test4.hip) needs 24-25 us; paced like k_rollout_ws's batches (1,3,4,8,8,..) 29-31 us.
// The probe also says HOW to store: fully contiguous 960-byte wave instructions, non-temporal (each instruction writes
// whole 64-byte pieces, nothing to merge in L2): 25.7 us for the launch's bytes against 30.7 us for 2 x 16 B per lane at a
// 32-byte stride through L2.  So an emit lane here expands 16 observation bits into ONE 16-byte piece of two rows
// (rows r and r + 2 of its group: the two instructions cover rows 0-1 and rows 2-3).  Everything else the launch writes
// (mask rows, scalar columns) goes out write-through (store_wt16): plain stores would leave 15 MB of dirty L2 lines to be
// written back after the last wave has ended (27.4 -> 25.1 us).
//
// Serves substeps == 1, n % 32 == 0 with every Transition column requested (the BASELINE configuration), <= 40 steps per
// launch (brl_rollout_random runs a longer rollout as pieces);
// everything else takes k_rollout_ws / k_rollout_random.  Bit-identical outputs (tests/test_gpu_parity.py).
#pragma once

constexpr int FS_TPB = 32;
constexpr int FS_NW = 13;          // logic, loader/mask, scorer A, 8 emit, scorer B, prep
constexpr int FS_MAX_TOTAL = 40;   // sub-steps per launch
#ifndef FS_RING_N        // (timing experiments only: fewer prefetched boards give WRONG results for tables that deal more)
#define FS_RING_N 12
#endif
constexpr int FS_RING = FS_RING_N;
constexpr int FS_CHUNK = 8;        // scorer: slots per pass
#ifndef FS_EXP
#define FS_EXP 0                   // timing experiments (scripts/timing_fs.py, FLAGS=-DFS_EXP=8): 8 = no observation stores
#endif

// (explicit LDS address space: a volatile access through a generic pointer becomes a FLAT load whose wait, vmcnt(0), also
// waits for every store the wave has in flight)
typedef __attribute__((address_space(3))) volatile int fs_lds_int;
__device__ __forceinline__ int fs_flag_read(const int *p) {
  int v = *(fs_lds_int *)p;
  asm volatile("" ::: "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void fs_flag_write(int *p, int v) {
  asm volatile("" ::: "memory");  // everything written before stays before (same-wave LDS order does the rest)
  *(fs_lds_int *)p = v;
}
// wait until slot s is posted; `avail` caches the last value seen
__device__ __forceinline__ void fs_wait(const int *flag, int s, int &avail) {
  if (s < avail) return;
  for (;;) {
    avail = fs_flag_read(flag);
    if (s < avail) return;
    __builtin_amdgcn_s_sleep(1);
  }
}

#ifdef BRL_TIMING
#define FS_STAMP(k) do { if (c.lane == 0 && A.terminated_count) fs_dump[2 * (k)] = __builtin_amdgcn_s_memtime() - t_begin; } while (0)
#else
#define FS_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(FS_NW * 64) void k_rollout_fs(RolloutArgs A) {
  constexpr int TPB = FS_TPB, NW = FS_NW;
#ifdef BRL_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_wait = 0;
#endif
  __shared__ __attribute__((aligned(16))) uint8_t img[TPB * TABLE_BYTES];
  __shared__ __attribute__((aligned(16))) uint32_t cmd[FS_MAX_TOTAL + 1][TPB][CMD_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t ring[TPB][FS_RING][RING_WORDS];
  __shared__ uint32_t udraw[FS_MAX_TOTAL + 4][TPB];
  __shared__ float s_neglog[BRL_NUM_ACTIONS + 2];
  __shared__ __attribute__((aligned(16))) uint32_t raw[FS_MAX_TOTAL + 1][TPB][4];  // logic -> prep (see the logic wave)
  __shared__ int posted, raw_posted, ring_count, gae_ready;
  // emit waves: the packed observation of each table AS SEEN BY each of the four seats (15 dwords: vulnerability nibble,
  // history rotated to that observer, its hand), kept up to date call by call: a row is a copy of one of them
  __shared__ __attribute__((aligned(16))) uint32_t oimg[TPB][4][16];
  // scorer A -> scorer B: the launch's finished boards in slot order (<= 11 per table), per-slot info (actor, action,
  // n_legal, done), and B's results: reward of the acting player per slot and table (zero unless a board ended there)
  __shared__ __attribute__((aligned(16))) uint32_t evq[FS_RING * TPB][4];
  // (8 rows in front of each: the GAE scan works in blocks of 8 steps and lets the last block run over the start)
  __shared__ __attribute__((aligned(16))) uint32_t minfo_s[8 + FS_MAX_TOTAL][TPB];
  __shared__ __attribute__((aligned(16))) float frew_s[8 + FS_MAX_TOTAL][TPB];
  uint32_t(*const minfo)[TPB] = minfo_s + 8;
  float(*const frew)[TPB] = frew_s + 8;
  __shared__ __attribute__((aligned(16))) uint32_t last_rw[TPB][4];  // final (fd, ring entry) from A; rewards words of a board that ended in the LAST slot
  __shared__ int scored, ev_count;

  const int tid = (int)threadIdx.x;
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Hardware wave w runs on SIMD w % 4, and within a SIMD the older wave wins the issue arbitration.  SIMD 0 (hardware
  // waves 0, 4, 8, 12) hosts the logic wave and the three light followers (prep, scorer A, scorer B): an emit wave there lost
  // ~15 % of its pace to the logic wave (priority 3) and ended 5-8 k cycles after the others.  Loader / mask = hardware
  // wave 3 (it must start early); the 8 emit waves are 3 + 3 + 2 on SIMDs 1, 2, 3.
  //                                  hw: 0  1  2  3  4   5  6  7  8  9  10 11  12
  constexpr uint64_t ROLE_OF_HW = 0x0ull | (3ull << 4) | (4ull << 8) | (1ull << 12) | (12ull << 16) | (5ull << 20) | (6ull << 24) |
                                  (7ull << 28) | (2ull << 32) | (8ull << 36) | (9ull << 40) | (10ull << 44) | (11ull << 48);
  const int wave = (int)((ROLE_OF_HW >> (4 * hw_wave)) & 15ull);
  const LaneConst c = make_lane_const();
#ifdef BRL_TIMING  // stamps go straight to the dump area behind the per-wave summary (terminated_count doubles as dump buffer)
  unsigned long long *fs_dump = A.terminated_count + (size_t)gridDim.x * NW * 2 + ((size_t)blockIdx.x * NW + wave) * 32;
#endif
  const int64_t table0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * TPB;
  const int total = A.T;  // substeps == 1: sub-step s == macro-step s; command slots 0..total
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  // Prologue: every global load of the workgroup is issued before anything waits (one memory round trip), the Philox
  // draws run while the loads are in flight.
  static_assert(TPB * 16 <= NW * 64 && TPB * 64 <= 3 * NW * 64, "one table word and <= 3 image dwords per thread");
  const uint64_t st_word = (tid < TPB * 16) ? A.state[table0 * 16 + tid] : 0ull;
  const int lt = c.lane & (TPB - 1);
  uint64_t ctr_word = 0;
  if (wave == 1) ctr_word = A.state[(table0 + lt) * 16 + W_CTR];
  // observer images from the packed tables (words 0..6 history with absolute seats, 7..10 hand words, 11 scalars):
  // image dword (table t, observer o, dword q) = task 64 t + 16 o + q (q = 15: padding)
  uint32_t oi_a[3], oi_sc[3];
  uint64_t oi_h[3];
  {
    const uint32_t *st32 = reinterpret_cast<const uint32_t *>(A.state + table0 * 16);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int task = tid + k * NW * 64;
      const int tk = (task < TPB * 64) ? task : 0;
      const int t = tk >> 6, o = (tk >> 4) & 3, q = tk & 15;
      oi_a[k] = st32[t * 32 + ((q < 13) ? q : 13)];
      oi_h[k] = A.state[(table0 + t) * 16 + W_HAND + o];
      oi_sc[k] = st32[t * 32 + 2 * W_SC];
    }
  }
  if (tid <= BRL_NUM_ACTIONS) s_neglog[tid] = A.neg_log_n[tid];
  // (loader) board pass i: lanes 0..31 fetch board nb0 + 2 i of their table, lanes 32..63 board nb0 + 2 i + 1 (Philox ->
  // LUT row -> packed hand words + DDS values, 48 B) into ring entries 2 i, 2 i + 1.  Passes 0 and 1 are ISSUED HERE, before
  // the workgroup's barrier: the logic wave needs the first two entries of every table at its first deal, a few hundred
  // cycles after the barrier, and waits for them otherwise (~1.7 k cycles in its first 8 sub-steps).
  constexpr int LP = FS_RING / 2;
  brl_u32x4 ld_ha[LP], ld_hb[LP], ld_v[LP];
  uint32_t ld_idx[LP], ld_scb[LP];
  const int ld_half = c.lane >> 5;
  auto ld_issue = [&](int i) {
    const uint64_t eid = A.env_offset + (uint64_t)(table0 + lt);
    const uint32_t nb0 = (uint32_t)(ctr_word >> 32) + 1u;
    board_params(A.g, eid, nb0 + (uint32_t)(2 * i + ld_half), A.lut.len, ld_idx[i], ld_scb[i]);
    ld_ha[i] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)ld_idx[i]];
    ld_hb[i] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)ld_idx[i] + 1];
    ld_v[i] = reinterpret_cast<const brl_u32x4 *>(A.lut.values)[ld_idx[i]];
  };
  auto ld_commit = [&](int i) {
    uint4 *dst = reinterpret_cast<uint4 *>(&ring[lt][2 * i + ld_half][0]);  // entry j = the j-th board dealt in this launch
    brl_u32x4 *dv = reinterpret_cast<brl_u32x4 *>(dst);
    dv[0] = ld_ha[i];
    dv[1] = ld_hb[i];
    dv[2] = ld_v[i];
    dst[3] = make_uint4(ld_idx[i], ld_scb[i], 0u, 0u);
    if (c.lane == 0) fs_flag_write(&ring_count, 2 * i + 2);
  };
  if (wave == 1) {
    ld_issue(0);
    ld_issue(1);
  }
  {
    // every action draw of the launch (Philox is state-independent): draw d = draw_base + s lives in word d & 3 of block
    // d >> 2 (counter arithmetic mod 2^32, like k_rollout_random); one (table, block) per thread
    const uint32_t fb = A.draw_base >> 2;
    const int nblk = (total > 0) ? (int)(((A.draw_base & 3u) + (uint32_t)total + 3u) >> 2) : 0;
    // (tasks start at hardware wave 4: one Philox wave per SIMD before any SIMD gets a second one, none on hardware
    //  waves 0..3 (logic, loader, the two oldest emit waves) — the multiplies are quarter-rate and the barrier waits
    //  for the slowest wave)
    for (int task = (hw_wave >= 4) ? tid - 4 * 64 : TPB * nblk; task < TPB * nblk; task += (NW - 4) * 64) {
      const int tb = task & (TPB - 1);
      const uint32_t blk = (fb + (uint32_t)(task >> 5)) & 0x3FFFFFFFu;
      const uint64_t eid = A.env_offset + (uint64_t)(table0 + tb);
      uint32_t rb[4];
      philox4x32_10((uint32_t)eid, blk, STREAM_ACTION, (uint32_t)(eid >> 32), A.g.k0, A.g.k1, rb);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t s = ((blk << 2) | (uint32_t)k) - A.draw_base;
        if (s < (uint32_t)total) udraw[s][tb] = rb[k];
      }
    }
  }
  FS_STAMP(9);   // Philox done
  if (tid < TPB * 16) img64[tid] = st_word;
  FS_STAMP(10);  // first global load back
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int task = tid + k * NW * 64;
    if (task < TPB * 64) {
      const int t = task >> 6, o = (task >> 4) & 3, q = task & 15;
      const uint32_t a = oi_a[k];
      const uint64_t H = oi_h[k];
      const uint32_t m16 = __umul24(0xFu >> o, 0x1111u), m1 = m16 | (m16 << 16);  // nibble mask 0xF >> o in every nibble
      const uint32_t rot = ((a >> o) & m1) | ((a << (4 - o)) & ~m1);  // relative seat = (caller - observer) mod 4
      uint32_t v = (q < 13) ? rot : ((q == 13) ? ((rot & 0xFFFu) | (uint32_t)(H << 8)) : ((q == 14) ? (uint32_t)(H >> 24) : 0u));
      v |= (q == 0) ? vul_nibble_sc(oi_sc[k], o) : 0u;
      oimg[t][o][q] = v;
    }
  }
  for (int i = tid; i < FS_MAX_TOTAL * TPB; i += NW * 64) (&frew[0][0])[i] = 0.0f;
  if (tid < TPB * 4) (&last_rw[0][0])[tid] = 0u;
  if (tid == 0) {
    posted = 0;
    raw_posted = 0;
    ring_count = 0;
    gae_ready = 0;
    scored = 0;
    ev_count = 0;
  }
  FS_STAMP(11);  // images built
  __syncthreads();  // images, draws and the two counters are in LDS
  FS_STAMP(12);

  if (wave == 0) {
    // ------------------------------------------------------------------ logic wave: lane = table, lanes 32..63 mirror 0..31
    // Only what feeds the chain: state s -> call -> state s+1 (+ re-deal).  It posts raw[s][table] = (sc, sch of state s,
    // call of sub-step s-1 | dealt << 8 | ring entry << 16, sc right after sub-step s-1) and publishes raw_posted; the prep
    // wave turns that into the followers' command (legal mask, n_legal, history bit, observer seat, vulnerability).
    const int tl = lt;
    uint32_t sc, sch, lut, bctr;
    {
      const uint2 *p = reinterpret_cast<const uint2 *>(img + tl * TABLE_BYTES);
      uint2 a = p[W_SC], d = p[W_CTR];
      sc = a.x; sch = a.y; lut = d.x; bctr = d.y;
    }
    __builtin_amdgcn_s_setprio(3);
    uint2 nxt = make_uint2(0u, 0u);  // (LUT row, fresh scalars) of this table's next board, read one deal ahead
    bool have_nxt = false;
    int ring_seen = 0, kub = 0;      // ring passes known complete; deals of any one table so far (upper bound)
    uint32_t kt = 0;                 // boards this table has dealt in this launch = ring entry of its next board
    uint32_t pa = 0, psc = 0;
    uint32_t un = udraw[0][tl];
    for (int s = 0;; s++) {
      if (c.lane < TPB) *reinterpret_cast<uint4 *>(&raw[s][tl][0]) = make_uint4(sc, sch, pa, psc);
      if (c.lane == 0) fs_flag_write(&raw_posted, s + 1);
      if ((s & 7) == 0) FS_STAMP(s >> 3);
      if (s == total) break;
      const uint32_t u = un;
      un = udraw[(s + 2 < total) ? s + 1 : total - 1][tl];  // next sub-step's draw, off the chain
      const uint32_t seat = lean_seat(sc, sch), lb1 = bits(sc, SC_LB1, 6);
      const uint32_t a = lean_pick(sc, seat, lb1, u);
      const uint32_t term = lean_apply(sc, sch, a, seat, lb1);
      psc = sc;
      pa = a | (term << 8) | (kt << 16);
      if (__any(term != 0u)) {
        // the j-th sub-step with a deal reads ring entries <= j (entry of the deal + the one read ahead)
        kub++;
        const int need = (kub + 1 < FS_RING) ? kub + 1 : FS_RING;
        while (ring_seen < need) {
          ring_seen = fs_flag_read(&ring_count);
          if (ring_seen < need) __builtin_amdgcn_s_sleep(1);
        }
        if (!have_nxt) {  // (no table has dealt yet: kt == 0 everywhere)
          nxt = *reinterpret_cast<const uint2 *>(&ring[tl][0][12]);
          have_nxt = true;
        }
        if (term) {  // A5 post-step half of auto_reset (src/utils.py:45-55): next board from the ring
          sc = nxt.y | (sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL)));
          sch = 0;
          lut = nxt.x;
          bctr += 1u;
          kt += 1u;
          nxt = *reinterpret_cast<const uint2 *>(&ring[tl][(kt < FS_RING) ? kt : FS_RING - 1][12]);
        }
      }
    }
    if (c.lane < TPB) {
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      p[W_SC] = make_uint2(sc, sch);
      p[W_CTR] = make_uint2(lut, bctr);
    }
    if (A.gae_adv != nullptr) {
      // optional: calc_gae of this trajectory (src/gae.py:20-39; the same operations in the same order as k_gae, with the
      // value column this launch writes: 0).  This wave (highest issue priority) is done ~10 k cycles before the emit waves,
      // so the scan (lane = table, reverse over the rewards / dones the scorer left in LDS) hides in their tail.
      const float gae_lv = A.gae_last_val[table0 + lt];
      while (fs_flag_read(&gae_ready) == 0) __builtin_amdgcn_s_sleep(1);
      FS_STAMP(5);  // rewards complete
      if (c.lane < TPB) {
        float gae = 0.0f, next_value = gae_lv;
        const float g0 = A.gae_gamma * 0.0f;  // gamma * next_value for every step but the last one (value column == 0)
        for (int t1 = total; t1 > 0; t1 -= 8) {
          // blocks of 8 steps, their 16 LDS values fetched together; no branch and no index clamp per step: the steps t < 0
          // of the last block come AFTER every real step of the chain and live in the rows in front of the arrays
          float rr[8];
          uint32_t mi[8];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            rr[k] = frew[t1 - 1 - k][lt];
            mi[k] = minfo[t1 - 1 - k][lt];
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const float nd = ((mi[k] >> 14) & 1u) ? 0.0f : 1.0f;                                   // 1 - done, src/gae.py:27
            const float gnv = (k == 0 && t1 == total) ? A.gae_gamma * next_value : g0;
            const float delta = rr[k] + gnv * nd - 0.0f;                                           // src/gae.py:28 (value == 0)
            gae = delta + A.gae_gamma_lambda * nd * gae;                                           // src/gae.py:29
            frew[t1 - 1 - k][lt] = gae;  // advantages in place (scorer B read its last chunk's rewards BEFORE raising gae_ready); targets = gae + value: in the store loop
          }
        }
      }
      FS_STAMP(6);  // scan done
      wave_lds_order();
      // the whole wave writes them: lane l — 4 consecutive tables of step l / 8 (+ 8, 16, ..), 16-byte write-through stores
      const int q = c.lane >> 3, t4 = 4 * (c.lane & 7);
      for (int t = q; t < total; t += 8) {
        const int64_t i = (int64_t)t * A.n + table0 + t4;
        const float4 a = *reinterpret_cast<const float4 *>(&frew[t][t4]);
        const float4 b = make_float4(a.x + 0.0f, a.y + 0.0f, a.z + 0.0f, a.w + 0.0f);  // targets = advantages + value, src/gae.py:39
        store_wt16(A.gae_adv + i, brl_u32x4{__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)});
        store_wt16(A.gae_tgt + i, brl_u32x4{__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)});
      }
    }
  } else if (wave == 12) {
    // ------------------------------------------------------------------ prep wave: raw -> command, two slots per pass
    // (lanes 0..31: slot s, lanes 32..63: slot s + 1 when it is already there).  Command format: brl_rollout.hip (k_rollout_ws).
    int avail = 0;
    int s = 0;
    while (s <= total) {
      fs_wait(&raw_posted, s, avail);
      const bool two = (s + 1 < avail);  // (avail <= total + 1)
      const int ms = s + ((two && c.lane >= TPB) ? 1 : 0);
      if (two || c.lane < TPB) {
        const uint4 cur = *reinterpret_cast<const uint4 *>(&raw[ms][lt][0]);
        const uint2 prv = *reinterpret_cast<const uint2 *>(&raw[(ms > 0) ? ms - 1 : 0][lt][0]);
        const uint32_t seat = lean_seat(cur.x, cur.y);
        uint32_t nl_cur, nl_prv;
        const uint64_t legal = lean_legal(cur.x, seat, nl_cur);
        const uint32_t seatp = lean_seat(prv.x, prv.y);
        (void)lean_legal(prv.x, seatp, nl_prv);
        const uint32_t a = cur.z & 63u;
        const uint32_t hb1 = lean_hb1(bits(prv.x, SC_LB1, 6), seatp, a);
        uint32_t pend = hb1 | (((cur.z >> 8) & 1u) << 9) | (((cur.z >> 16) & 15u) << 16) | (seatp << 21) | (nl_prv << 23);
        pend = (ms > 0) ? pend : 0u;
        const uint32_t w0 = pend | (seat << 10);
        const uint32_t w3 = ((uint32_t)(legal >> 32) & 63u) | (a << 8);
        *reinterpret_cast<uint4 *>(&cmd[ms][lt][0]) = make_uint4(w0, cur.w, (uint32_t)legal, w3);
      }
      s += two ? 2 : 1;
      if (c.lane == 0) fs_flag_write(&posted, s);
    }
  } else if (wave == 1) {
    // ------------------------------------------------------------------ loader, then mask wave
    // (passes 0 and 1 were issued in the prologue; each commit publishes two more ring entries per table)
#pragma unroll
    for (int i = 0; i < LP; i++) {
      ld_commit(i);
      if (i + 2 < LP) ld_issue(i + 2);
    }
    FS_STAMP(8);
    // The 32 legal-mask rows of a slot are 1216 contiguous bytes = 76 chunks of 16 B: lane l writes chunk l, lanes < 12
    // also chunk 64 + l.  A chunk holds the bytes of table ta (from action `off` on) and possibly of ta + 1.
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
    int avail = 0;
    for (int s = 0; s <= total; s++) {
      fs_wait(&posted, s, avail);
      uint8_t *dstrow = (s < total) ? mrow : ((A.last_mask != nullptr) ? A.last_mask + table0 * BRL_NUM_ACTIONS : nullptr);
      mrow += mstep;
      if (dstrow == nullptr) continue;
      const uint32_t(*cs)[CMD_WORDS] = cmd[s];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if (q == 1 && c.lane >= 12) break;
        const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[ta[q]][2]) & ALL_ACTIONS;
        const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[tb[q]][2]) & ALL_ACTIONS;
        const uint32_t bits16 = (uint32_t)((la >> off[q]) | (lb << (BRL_NUM_ACTIONS - off[q])));
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = __umul24((bits16 >> (4 * i)) & 0xFu, 0x204081u) & 0x01010101u;
        // (write-through: -1.6 us against plain stores, see fs_store_wt; non-temporal: +1.0 us)
        store_wt16(dstrow + 16 * (c.lane + 64 * q), brl_u32x4{d[0], d[1], d[2], d[3]});
      }
    }
  } else if (wave == 2) {
    // ------------------------------------------------------------------ scorer A: lane = table (lanes 32..63 idle).  The
    // part of the scoring that is SERIAL per table: first denominations, who acted, which boards ended (queued for
    // scorer B with the scalars they ended on), which ring entry a table plays.  ~600 cycles per slot when it has a SIMD's
    // issue slots; beside the logic wave (SIMD 0) it falls ~5 k cycles behind and catches up when that wave is done.
    // Publishes ev_count, then scored (slots done).
    const int tl = lt;
    const bool mine = c.lane < TPB;
    Tbl ts;
    load_scalars(ts, img + tl * TABLE_BYTES);
    uint32_t vslot = NO_SLOT;  // ring slot of the table's current board; NO_SLOT: the board it came in with
    int avail = 0, nev = 0;
    uint4 wn = make_uint4(0u, 0u, 0u, 0u);
    for (int s = 1; s <= total; s++) {  // slot s describes sub-step s - 1 = macro-step s - 1
      if (s >= avail) {
        fs_wait(&posted, s, avail);
        wn = *reinterpret_cast<const uint4 *>(&cmd[s][tl][0]);
      }
      const uint4 w = wn;
      if (s + 1 < avail) wn = *reinterpret_cast<const uint4 *>(&cmd[s + 1][tl][0]);  // (uniform) next command, off the chain
      const int a = (int)((w.w >> 8) & 63u);
      const int seat = (int)((w.x >> 21) & 3u);
      ts.sc = w.y;
      // the acting player (src/roll_out.py:72), its action, n_legal
      uint32_t info = (uint32_t)player_at(ts, seat) | ((uint32_t)a << 2) | (((w.x >> 23) & 63u) << 8);
      note_first_denomination(ts.fd, seat, a);
      const bool fin = mine && bits(ts.sc, SC_TERM, 1);
      const uint64_t fm = __ballot(fin);
      if (fm) {  // queue the finished boards, compacted over the tables: one lane per board in scorer B
        if (fin) {
          const int pos = nev + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
          *reinterpret_cast<uint4 *>(&evq[pos][0]) = make_uint4(ts.sc, ts.fd, (uint32_t)(s - 1) | ((uint32_t)tl << 8), vslot);
          info |= 1u << 14;  // done (G2)
        }
        nev += __popcll(fm);
      }
      const bool dealt = (w.x & 0x200u) != 0u;  // re-dealt: no strain named yet; DDS values stay in the ring entry
      vslot = dealt ? ((w.x >> 16) & 15u) : vslot;
      ts.fd = dealt ? 0u : ts.fd;
      if (mine) {
        minfo[s - 1][tl] = info;
        if (s == total) *reinterpret_cast<uint2 *>(&last_rw[tl][0]) = make_uint2(ts.fd, vslot);  // for B's write-back
      }
      if (c.lane == 0) {
        fs_flag_write(&ev_count, nev);
        fs_flag_write(&scored, s);
      }
      if ((s & 7) == 0) FS_STAMP(s >> 3);
    }
  } else if (wave == 11) {
    // ------------------------------------------------------------------ scorer B: what is PARALLEL — one lane per finished
    // board: contract -> DDS tricks -> score -> reward of the acting player (A4, G1) into frew[slot][table]; then the scalar
    // Transition columns of up to 8 slots, 16-byte write-through stores (lane l: 4 consecutive tables of slot l / 8).
    // A chunk costs ~3 k cycles whatever its length, so this wave is the one that runs in chunks.
    const int tl = lt;
    uint32_t tcount = 0;
    int b0 = 0, seen = 0, ev_done = 0;  // slots [0, b0) written; scorer A's counters as last read
    while (b0 < total) {
      fs_wait(&scored, b0, seen);
      const int evs = fs_flag_read(&ev_count);  // (read AFTER scored: every board that ended in a slot < seen is queued)
      const int c1 = (seen < b0 + FS_CHUNK) ? seen : b0 + FS_CHUNK;  // slots [b0, c1)
      const int m = c1 - b0;
      for (int e0 = ev_done; e0 < evs; e0 += 64) {
        const int e = e0 + c.lane;
        if (e < evs) {
          const uint4 q = *reinterpret_cast<const uint4 *>(&evq[e][0]);
          const uint32_t tt = (q.z >> 8) & 63u, sl = q.z & 0xFFu;
          Tbl tb;
          tb.sc = q.x; tb.fd = q.y;
          if (q.w != NO_SLOT) {  // a board dealt in this launch: DDS values from its ring entry
            const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tt][q.w][8]);
            pack_tricks(tb, vv.x, vv.y, vv.z, vv.w);
          } else {  // the board the table came in with: its tricks are in the packed image
            const uint2 *ip = reinterpret_cast<const uint2 *>(img + tt * TABLE_BYTES);
            const uint2 tr = ip[W_TR], fdw = ip[W_FD];
            tb.t0 = tr.x; tb.t1 = tr.y; tb.t2 = fdw.y;
          }
          terminal_reward(tb);  // A4
          const int actor = (int)(minfo[sl][tt] & 3u);
          frew[sl][tt] = (float)reward_of(tb, actor) / A.reward_scale;  // G1, src/roll_out.py:90
          if ((int)sl == total - 1) *reinterpret_cast<uint2 *>(&last_rw[tt][2]) = make_uint2(tb.r01, tb.r23);
        }
      }
      ev_done = evs;
      wave_lds_order();
      // (frew of every slot < c1 is final: a board that ended in slot sl was queued before `scored` passed sl)
      {
        const int q = c.lane >> 3, t4 = 4 * (c.lane & 7);
        // This chunk's rewards / infos are READ before gae_ready is raised: the GAE scan overwrites frew in place (advantages),
        // starting with the last slots — the ones this chunk still has to store as Transition.reward.  LDS operations of
        // one wave are performed in issue order, so reads issued before the flag store see the rewards, not the scan's values.
        uint4 info4 = make_uint4(0u, 0u, 0u, 0u);
        float4 rew = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < m) {
          info4 = *reinterpret_cast<const uint4 *>(&minfo[b0 + q][t4]);
          rew = *reinterpret_cast<const float4 *>(&frew[b0 + q][t4]);
        }
        wave_lds_order();
        if (A.gae_adv != nullptr && c1 == total && c.lane == 0) fs_flag_write(&gae_ready, 1);
        if (q < m) {
          const uint32_t inf[4] = {info4.x, info4.y, info4.z, info4.w};
          float lgp[4];
          uint32_t act[4], dn = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            lgp[k] = s_neglog[(inf[k] >> 8) & 63u];
            act[k] = (inf[k] >> 2) & 63u;
            const uint32_t done = (inf[k] >> 14) & 1u;
            dn |= done << (8 * k);
            tcount += done;
          }
          const int64_t rw = (int64_t)(b0 + q) * A.n + table0 + t4;
          store_wt16(A.out.action + rw, brl_u32x4{act[0], act[1], act[2], act[3]});
          store_wt16(A.out.value + rw, brl_u32x4{0u, 0u, 0u, 0u});
          store_wt16(A.out.reward + rw, brl_u32x4{__float_as_uint(rew.x), __float_as_uint(rew.y), __float_as_uint(rew.z), __float_as_uint(rew.w)});
          store_wt16(A.out.log_prob + rw, brl_u32x4{__float_as_uint(lgp[0]), __float_as_uint(lgp[1]), __float_as_uint(lgp[2]), __float_as_uint(lgp[3])});
          *reinterpret_cast<uint32_t *>(A.out.done + rw) = dn;  // G2 (4-byte pieces: plain; write-through no faster)
        }
      }
      b0 = c1;
      if ((b0 & 7) == 0) FS_STAMP(b0 >> 3);
    }
    if (A.gae_adv != nullptr && total == 0 && c.lane == 0) fs_flag_write(&gae_ready, 1);  // (nothing to scan)
    if (A.terminated_count != nullptr) {  // src/roll_out.py:85
      uint32_t v = tcount;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
#ifndef BRL_TIMING
      if (c.lane == 0 && v) atomicAdd(A.terminated_count, (unsigned long long)v);
#endif
    }
    if (total > 0 && c.lane < TPB) {
      // scalar words of the table's final state: first denominations and the DDS values of its current board (scorer A
      // left fd and the ring entry), rewards of the last macro-step (src/utils.py:126): its board's, or zero
      const uint4 f = *reinterpret_cast<const uint4 *>(&last_rw[tl][0]);
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      if (f.y != NO_SLOT) {
        Tbl tb;
        const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tl][f.y][8]);
        pack_tricks(tb, vv.x, vv.y, vv.z, vv.w);
        p[W_FD] = make_uint2(f.x, tb.t2);
        p[W_TR] = make_uint2(tb.t0, tb.t1);
      } else {
        p[W_FD] = make_uint2(f.x, p[W_FD].y);
      }
      p[W_REW] = make_uint2(f.z, f.w);
    }
  } else if (wave <= 10) {
    // ------------------------------------------------------------------ emit waves: wave 3 + g owns tables 4 g .. 4 g + 3
    // lane L < 60 holds 16-byte piece L of rows 0-1 (960 contiguous bytes) and piece L of rows 2-3: piece L is half
    // L & 1 of packed dword q = (L >> 1) % 15 of row r = (L >> 1) / 15 (and of row r + 2): 16 bits of the acting seat's
    // observer image, one ds_read_u16.  Lanes 0..7 (rows 0, 2) and 30..37 (rows 1, 3) also keep the images up to date:
    // lane (row, observer o) sets the call's history bit at relative seat (caller - o) mod 4.
    const int g = wave - 3;
    const int L = c.lane;
    const bool active = L < 60;
    const int pq = active ? (L >> 1) : 29;
    const int r = pq / 15, q = pq - 15 * r;
    const int half = L & 1;
    const int hl = (L < 30) ? L : L - 30;
    const bool head = active && hl < 8;
    const int hsel = head ? (hl >> 2) : 0;  // 0: the lane's row r, 1: its row r + 2
    const uint32_t ho = (uint32_t)(hl & 3);
    uint8_t *og = reinterpret_cast<uint8_t *>(&oimg[4 * g][0][0]);  // this group's 4 x 4 images, 64 B each
    uint32_t *oh = reinterpret_cast<uint32_t *>(og + ((r + 2 * hsel) * 4 + (int)ho) * 64);
    const int rd_off = r * 256 + q * 4 + half * 2;
    const int dl_o = active ? L / 15 : 3, dl_q = active ? L - 15 * (L / 15) : 15;  // deals: lane = (observer, dword)
    uint8_t *optr = A.out.obs + (table0 + 4 * g) * BRL_OBS_SIZE + 16 * L;
    const int64_t ostep = A.n * BRL_OBS_SIZE;
    int avail = 0;
    for (int s = 0; s <= total; s++) {
#ifdef BRL_TIMING
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      fs_wait(&posted, s, avail);
#ifdef BRL_TIMING
      t_wait += __builtin_amdgcn_s_memtime() - tw0;
      if ((s & 7) == 0) FS_STAMP(s >> 3);
#endif
      if (s == (total >> 1)) {
        // The emit waves of a SIMD do the same work, but the oldest is served first and ends ~8 k cycles before the
        // youngest — while the stores of the last ones no longer fill HBM.  From here on the younger go first: they all
        // end within ~1 k cycles of each other (launch -0.5 us).
        if (hw_wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (hw_wave >= 4) __builtin_amdgcn_s_setprio(1);
      }
      const uint32_t(*cs)[CMD_WORDS] = cmd[s];
      const uint32_t w0a = cs[4 * g + r][0], w0b = cs[4 * g + r + 2][0];
      const uint32_t wh = hsel ? w0b : w0a;
      // apply sub-step s-1 to the images: one history bit per observer, or a freshly dealt board
      if (head && !(wh & 0x200u) && (wh & 0x1FFu)) {
        const uint32_t hb = (wh & 0x1FFu) - 1u;
        const uint32_t bit = (hb & ~3u) | ((hb - ho) & 3u);
        atomicOr(oh + (bit >> 5), 1u << (bit & 31u));
      }
      uint64_t dealm = __ballot(head && ho == 0u && (wh & 0x200u));
      while (dealm) {  // rare: ~1 table in 25 per sub-step
        const int l = __ffsll((unsigned long long)dealm) - 1;  // lanes 0, 4, 30, 34: rows 0, 2, 1, 3
        dealm &= dealm - 1ull;
        const int row = (l >= 30) ? 1 + ((l - 30) >> 2) * 2 : (l >> 2) * 2;
        const uint32_t wq = __builtin_amdgcn_readlane(wh, l);
        const uint32_t *re = &ring[4 * g + row][(wq >> 16) & 15u][0];
        const uint64_t H = *reinterpret_cast<const uint64_t *>(re + 2 * dl_o);
        const uint32_t v = (dl_q == 0) ? vul_nibble_sc(re[13], dl_o)
                                       : ((dl_q == 13) ? (uint32_t)(H << 8) : ((dl_q == 14) ? (uint32_t)(H >> 24) : 0u));
        if (active) oimg[4 * g + row][dl_o][dl_q] = v;
      }
      wave_lds_order();
      const uint32_t v0 = *reinterpret_cast<const uint16_t *>(og + rd_off + (int)((w0a >> 10) & 3u) * 64);
      const uint32_t v1 = *reinterpret_cast<const uint16_t *>(og + rd_off + 512 + (int)((w0b >> 10) & 3u) * 64);
      uint8_t *dst = (s < total) ? optr : ((A.last_obs != nullptr) ? A.last_obs + (table0 + 4 * g) * BRL_OBS_SIZE + 16 * L : nullptr);
      optr += ostep;
      if (active && dst != nullptr) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t word = k ? v1 : v0;
          brl_u32x4 d;
          d.x = __umul24(word & 0xFu, 0x204081u) & 0x01010101u;
          d.y = __umul24((word >> 4) & 0xFu, 0x204081u) & 0x01010101u;
          d.z = __umul24((word >> 8) & 0xFu, 0x204081u) & 0x01010101u;
          d.w = __umul24(word >> 12, 0x204081u) & 0x01010101u;
          if (!(FS_EXP & 8)) __builtin_nontemporal_store(d, reinterpret_cast<brl_u32x4 *>(dst + 960 * k));
          else if (d.x == 0x12345678u) __builtin_nontemporal_store(d, reinterpret_cast<brl_u32x4 *>(dst + 960 * k));  // timing experiment: no stores
        }
      }
    }
  }
#ifdef BRL_TIMING
  if (c.lane == 0 && A.terminated_count) {  // timing build only: terminated_count doubles as a dump buffer
    unsigned long long *d = A.terminated_count + ((size_t)blockIdx.x * NW + wave) * 2;
    d[0] = __builtin_amdgcn_s_memtime() - t_begin;
    d[1] = t_wait;
    fs_dump[30] = __builtin_amdgcn_s_memrealtime() - rt_begin;  // 100 MHz ticks
  }
#endif
  __syncthreads();
  // packed tables back: history (words 0..6) = the image of observer 0 without its vulnerability nibble and hand bits,
  // hand words (7..10) from each observer's image, words 11..15 from the scalar image
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    const int t = i >> 4, w = i & 15;
    uint64_t v;
    if (w < 7) {
      uint32_t lo = oimg[t][0][2 * w], hi = oimg[t][0][2 * w + 1];
      lo = (w == 0) ? (lo & ~0xFu) : lo;
      hi = (w == 6) ? (hi & 0xFFFu) : hi;
      v = (uint64_t)lo | ((uint64_t)hi << 32);
    } else if (w < 11) {
      const uint32_t d13 = oimg[t][w - 7][13], d14 = oimg[t][w - 7][14];
      v = ((uint64_t)d14 << 24) | (uint64_t)((d13 >> 8) & 0xFFFFF0u);
    } else {
      v = img64[i];
    }
    A.state[table0 * 16 + i] = v;
  }
}
