// rollout_pipe.hpp — the fused random-policy rollout as a three-stage batch pipeline.
// Same roles, LDS images, commands and emit path as k_rollout_ws (brl_kernels.hip), with the per-table
// dependency chain cut down to its minimum: the logic wave runs fast_step (rollout_common.hpp) on a packed
// word and posts only that state; NP prep waves, one batch behind, re-run the full step slot-parallel and
// build the 16-byte commands; the loader / scorer / emit waves work two batches behind the logic wave.
// One s_barrier per batch as in k_rollout_ws — waiting waves cost no issue slots.
//   stage t:  logic = batch t   |  prep = batch t-1  |  scorer, emit, loader bookkeeping = batch t-2
// substeps > 1 or a caller-supplied finished table (all-True mask): the logic wave runs the full step and
// posts the commands itself (legacy mode), the prep waves only keep the barrier count.
#pragma once

constexpr int PR_RING = 16;  // boards kept ahead per table (k_rollout_pipe: the logic wave leads the followers by two batches)

template <int TPB, int NW, int NP>
__global__ __launch_bounds__(NW * 64) void k_rollout_pipe(RolloutArgs A) {
  static_assert(TPB <= 32 && NP >= 1 && NW >= 4 + NP, "logic + loader + scorer + NP prep + >=1 emit wave");
  static_assert(TPB % 4 == 0, "emit waves write 4 consecutive tables per instruction");
  constexpr int NE = NW - 3 - NP;
  constexpr int E0 = 3 + NP;  // first emit wave
  constexpr int B = WS_BATCH;
#ifdef BRL_TIMING
  unsigned long long t_wait = 0, t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_arr[16], t_rel[16];
  int t_nb = 0;
#endif
  __shared__ __attribute__((aligned(16))) uint8_t img[TPB * TABLE_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t bimg[TPB * BROW];  // byte images (emit waves)
  __shared__ __attribute__((aligned(16))) uint32_t cmd[4][B][TPB][CMD_WORDS];  // batch b -> cmd[b & 3]
  __shared__ __attribute__((aligned(8))) uint2 spost[2][B + 1][TPB];  // fast mode: (d, static word) of the states of batch b; entry 0 = the state before the batch
  __shared__ __attribute__((aligned(16))) uint32_t ring[TPB][PR_RING][RING_WORDS];
  __shared__ int ring_ready;  // set by the loader wave once the first two boards of every table are in the ring
  __shared__ uint32_t udraw[4][WS_BATCH][TPB];  // action draws of batch b -> udraw[b & 3], precomputed by the loader wave
  __shared__ float s_neglog[BRL_NUM_ACTIONS + 2];
  const int tid = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LaneConst c = make_lane_const();
  const int64_t table0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * TPB;
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    img64[i] = (tb < A.n) ? A.state[table0 * 16 + i] : 0ull;
  }
  if (tid <= BRL_NUM_ACTIONS) s_neglog[tid] = A.neg_log_n[tid];
  const int total = A.T * A.substeps;   // sub-steps; command slots are s = 0..total
  const int nbatch = ws_nbatch(total);  // batches of command slots (ws_bstart / ws_blen)
  const int tl = c.lane;                // logic / loader / scorer: lane = table
  const int tls = (tl < TPB) ? tl : 0;
  const bool valid = (tl < TPB) && (table0 + tl < A.n);
  const uint64_t env_id = A.env_offset + (uint64_t)(table0 + tl);
  // loader state (wave 1): next board to fetch, boards in flight
  uint32_t nb = 0, nb0 = 0, pbase = 0, pidx[3] = {0, 0, 0}, pscb[3] = {0, 0, 0};
  brl_u32x4 pha[3], phb[3], pv[3];  // (native vectors: HIP's uint4 struct arrays are not promoted to registers here)
  uint64_t ctr_word = 0;
  if (wave == 1 && valid) ctr_word = A.state[(table0 + tl) * 16 + W_CTR];  // issued now, needed after the barrier
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
      if (tl < TPB) udraw[b & 3][j][tl] = (sel == 0) ? rbk[0] : ((sel == 1) ? rbk[1] : ((sel == 2) ? rbk[2] : rbk[3]));
    }
  };
  if (wave == 1) draws(0);
  if (tid == 0) ring_ready = 0;
  __syncthreads();  // images and the draws of batch 0 are in LDS; the ring follows (ring_ready)
  uint32_t pcount = 0;  // (loader) boards whose loads are in flight
  if (wave == 1 && valid) {
    // the first two boards of every table: loads ISSUED here, committed to the ring after the first batch
    // barrier (loader loop) — everybody else has already started; only a DEAL needs the ring, and the logic
    // wave checks ring_ready before its first one
    nb0 = (uint32_t)(ctr_word >> 32) + 1u;
    nb = nb0;
    pbase = nb;
    pcount = 2;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      board_params(A.g, env_id, nb + (uint32_t)k, A.lut.len, pidx[k], pscb[k]);
      pha[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k]];
      phb[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k] + 1];
      pv[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.values)[pidx[k]];
    }
    nb += 2u;
  }

  // fast mode (substeps == 1, no caller-supplied finished table): the logic wave runs the minimal transition
  // and the prep waves build the commands; otherwise the logic wave runs the full step and posts the commands
  // itself (one stage earlier than the prep waves would — 4 command buffers cover both)
  const bool fastmode = (A.substeps == 1) && !(A.debug & 4) &&
                        !__any((tl < TPB) && bits(reinterpret_cast<const uint2 *>(img + tls * TABLE_BYTES)[W_SC].x, SC_MASKALL, 1));

  if (wave == 1) {
    // ------------------------------------------------------------------ loader wave
    // stage t (between barriers t-1 and t): draws of batch t+1; after barrier t: commit the boards fetched during
    // stage t, count the boards the tables consumed in batch t-1 (its commands are complete now) and fetch.
    // Boards dealt in batch b are read by the scorer / emit waves during stage b+2, so their ring slots may be
    // rewritten after barrier b+2: a fetch issued in stage t+1 (committed after barrier t+1) may reuse the slots
    // of boards dealt in batches <= t-1.  The logic wave, then at most in batch t+2, has consumed <= 3 boards per
    // batch since: <= 9 + the one it reads ahead < PR_RING.
    uint32_t dealt_total = 0;
    for (int t = 0; t < nbatch + 2; t++) {
      if (t + 1 < nbatch) draws(t + 1);  // the logic wave starts batch t+1 right after this barrier
      LDS_BARRIER();
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if ((uint32_t)k < pcount) {
          uint4 *dst = reinterpret_cast<uint4 *>(&ring[tls][(pbase + (uint32_t)k) % PR_RING][0]);
          brl_u32x4 *dv = reinterpret_cast<brl_u32x4 *>(dst);
          dv[0] = pha[k];
          dv[1] = phb[k];
          dv[2] = pv[k];
          dst[3] = make_uint4(pidx[k], pscb[k], 0u, 0u);
        }
      }
      pcount = 0;
      if (t == 0) {  // the first two boards are in the ring now
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c.lane == 0) __hip_atomic_store(&ring_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const int bq = t - 1;  // its commands are complete in both modes
      if (bq >= 0 && bq < nbatch) {
        uint32_t dealt = 0;
        for (int j = 0; j < ws_blen(bq); j++) {
          const int s = ws_bstart(bq) + j;
          if (s <= total) dealt += (cmd[bq & 3][j][tls][0] >> 9) & 1u;
        }
        dealt_total += dealt;
      }
      const uint32_t want = nb0 + (uint32_t)PR_RING - 1u + dealt_total;  // (-1: a board's entry lives until the NEXT deal: the scorer reads its DDS values when it ends)
      if (valid && nb < want && t + 1 < nbatch + 2) {
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
    uint2 nxt = make_uint2(0u, 0u);
    bool have_nxt = false;
    if (fastmode) {
      // ---- minimal transition (fast_step); the prep waves turn the posted states into commands
      uint32_t d = fast_from_legacy(sc, sch);
      uint32_t stw = (sc & 0x0A000FFFu) | ((bctr % PR_RING) << 28);  // board constants | TERM | ILLEGAL | ring slot
      uint32_t rslot = (bctr + 1u) % PR_RING;
      uint2 prev = make_uint2(d, stw);
      for (int bi = 0; bi < nbatch; bi++) {
        const int blen = ws_blen(bi);
        uint32_t un = udraw[bi & 3][0][tls];
        if (tl < TPB) spost[bi & 1][0][tl] = prev;  // the state before the batch, for the prep wave of its first slot
        for (int j = 0; j < blen; j++) {
          const int s = ws_bstart(bi) + j;
          if (s > total) break;
          const uint32_t u = un;
          un = udraw[bi & 3][(j + 1 < blen) ? j + 1 : j][tls];  // next sub-step's draw, off the chain
          prev = make_uint2(d, stw);
          if (tl < TPB) spost[bi & 1][1 + j][tl] = prev;
          if (j == blen - 1 || s == total) LDS_BARRIER();
          if (s == total) break;
          d = fast_step(d, u);
          const uint32_t term = (d >> FD_TERM) & 1u;
          stw = (stw & ~(1u << SC_TERM)) | (term << SC_TERM);
          const bool deal = valid && term;
          if (!have_nxt && __any(deal)) {  // uniform: first deal of the wave
            while (__hip_atomic_load(&ring_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            nxt = *reinterpret_cast<const uint2 *>(&ring[tls][rslot][12]);
            have_nxt = true;
          }
          if (deal) {  // A5 post-step half of auto_reset (src/utils.py:45-55): next board from the ring
            stw = (nxt.y & 0xFFFu) | (stw & ((1u << SC_TERM) | (1u << SC_ILLEGAL))) | (rslot << 28);
            d = (nxt.y & 3u) | (35u << FD_REM) | (2u << FD_E);
            lut = nxt.x;
            bctr += 1u;
            rslot = (rslot + 1u == (uint32_t)PR_RING) ? 0u : rslot + 1u;
            nxt = *reinterpret_cast<const uint2 *>(&ring[tl][rslot][12]);
          }
        }
      }
      fast_to_legacy(d, stw, sc, sch);
    } else {
      uint32_t pend = 0, pend_act = 0, pend_sc = 0, term_any = 0;
      int sub = 0;
      for (int bi = 0; bi < nbatch; bi++) {
        const int blen = ws_blen(bi);
        uint32_t un = udraw[bi & 3][0][tls];
        for (int j = 0; j < blen; j++) {
          const int s = ws_bstart(bi) + j;
          if (s > total) break;
          const uint32_t u = un;
          un = udraw[bi & 3][(j + 1 < blen) ? j + 1 : j][tls];
          uint32_t nsc = sc, nsch = sch;
          const LeanStep st = lean_random_step(nsc, nsch, u);
          if (tl < TPB) {  // command slot s: what sub-step s-1 did + how state s looks
            uint32_t w0 = pend | ((uint32_t)st.seat << 10) | (vul_nibble_sc(sc, st.seat) << 12);
            uint32_t w3 = ((uint32_t)(st.legal >> 32) & 63u) | (pend_act << 8);
            *reinterpret_cast<uint4 *>(&cmd[bi & 3][j][tl][0]) = make_uint4(w0, pend_sc, (uint32_t)st.legal, w3);
          }
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
          const uint32_t slot = (bctr + 1u) % PR_RING;
          pend = st.hb1 | ((uint32_t)deal << 9) | (slot << 16) | ((uint32_t)st.seat << 21) | ((uint32_t)st.n_legal << 23);
          if (!have_nxt && __any(deal)) {  // uniform: first deal of the wave
            while (__hip_atomic_load(&ring_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            nxt = *reinterpret_cast<const uint2 *>(&ring[tls][(bctr + 1u) % PR_RING][12]);
            have_nxt = true;
          }
          if (deal) {
            sc = nxt.y | (sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL)));
            sch = 0;
            lut = nxt.x;
            bctr += 1u;
            nxt = *reinterpret_cast<const uint2 *>(&ring[tl][(bctr + 1u) % PR_RING][12]);
          }
          if (last && A.substeps > 1) sc = (sc & ~(1u << SC_TERM)) | (term_any << SC_TERM);  // src/utils.py:127
        }
      }
    }
    LDS_BARRIER();  // the two stages in which the followers drain the pipeline
    LDS_BARRIER();
    if (tl < TPB) {
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      p[W_SC] = make_uint2(sc, sch);
      p[W_CTR] = make_uint2(lut, bctr);
    }
  } else if (wave >= 3 && wave < E0) {
    // ------------------------------------------------------------------ prep waves (fast mode)
    // Stage b+1: the commands of batch b from the states the logic wave posted for it.  Slot-parallel: prep wave p
    // takes the slots j = p, p + NP, ... of the batch; each one re-runs the full step on the state BEFORE the slot
    // (what sub-step s-1 did: history bit, action, n_legal, acting seat, deal) and looks at the state of the slot
    // itself (observer seat, vulnerability nibble, legal mask).
    __builtin_amdgcn_s_setprio(2);
    LDS_BARRIER();
    for (int bi = 0; bi < nbatch; bi++) {
      if (fastmode) {
        const int blen = ws_blen(bi), bstart = ws_bstart(bi);
        for (int j = wave - 3; j < blen; j += NP) {
          const int s = bstart + j;
          if (s > total) break;
          const uint2 cur = spost[bi & 1][1 + j][tls];
          uint32_t pend = 0, pend_sc = 0, pend_act = 0;
          if (s > 0) {
            const uint2 bef = spost[bi & 1][j][tls];
            const uint32_t u = (j > 0) ? udraw[bi & 3][j - 1][tls] : udraw[(bi - 1) & 3][ws_blen(bi - 1) - 1][tls];
            uint32_t psc, psch;
            fast_to_legacy(bef.x, bef.y, psc, psch);
            const LeanStep st = lean_random_step(psc, psch, u);
            pend_sc = psc;
            pend_act = (uint32_t)st.action;
            const bool deal = valid && st.term;
            pend = st.hb1 | ((uint32_t)deal << 9) | ((cur.y >> 28) << 16) | ((uint32_t)st.seat << 21) | ((uint32_t)st.n_legal << 23);
          }
          uint32_t csc, csch;
          fast_to_legacy(cur.x, cur.y, csc, csch);
          const uint32_t seat = cur.x & 3u;
          const uint32_t lb1 = bits(csc, SC_LB1, 6);
          const uint32_t own = ((bits(csc, SC_LBSEAT, 2) ^ seat) & 1u) ^ 1u;
          const uint32_t x = bits(csc, SC_X, 1), xx = bits(csc, SC_XX, 1), has = lb1 != 0;
          const uint32_t can_x = has & (own ^ 1u) & (x ^ 1u) & (xx ^ 1u);
          const uint32_t can_xx = has & own & x & (xx ^ 1u);
          const uint64_t legal = ((ALL_ACTIONS >> (3 + lb1)) << (3 + lb1)) | (uint64_t)(1u | (can_x << 1) | (can_xx << 2));
          if (tl < TPB) {
            const uint32_t w0 = pend | (seat << 10) | (vul_nibble_sc(csc, (int)seat) << 12);
            const uint32_t w3 = ((uint32_t)(legal >> 32) & 63u) | (pend_act << 8);
            *reinterpret_cast<uint4 *>(&cmd[bi & 3][j][tl][0]) = make_uint4(w0, pend_sc, (uint32_t)legal, w3);
          }
        }
      }
      LDS_BARRIER();
    }
    LDS_BARRIER();
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
    LDS_BARRIER();  // stage 0: nothing to score yet (the commands of batch b are complete after barrier b+1)
    for (int bi = 0; bi < nbatch; bi++) {
      LDS_BARRIER();
      // ---- pass 1
      int nev = 0, m = 0;  // nev: boards queued (uniform); m: macro-steps completed so far in this batch
      // acc[0] carries the partial sums of a macro-step that straddles the batch boundary
#pragma unroll
      for (int q = 1; q < B; q++) *reinterpret_cast<int4 *>(&acc[q][tl][0]) = make_int4(0, 0, 0, 0);
      const int blen = ws_blen(bi);
      uint4 wn = *reinterpret_cast<const uint4 *>(&cmd[bi & 3][0][tls][0]);
      for (int j = 0; j < blen; j++) {
        const int s = ws_bstart(bi) + j;
        if (s > total) break;
        const uint4 w = wn;  // the next command is fetched while this one is processed
        wn = *reinterpret_cast<const uint4 *>(&cmd[bi & 3][(j + 1 < blen) ? j + 1 : j][tls][0]);
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
      wave_lds_order();
      for (int e0 = 0; e0 < nev; e0 += 64) {  // (a table finishes at most one board per macro-step: no two lanes share a cell)
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
      wave_lds_order();
      if (m > 0) last_acc = *reinterpret_cast<const int4 *>(&acc[m - 1][tl][0]);
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
    LDS_BARRIER();
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
  } else {
    // ------------------------------------------------------------------ emit waves
    // Each wave owns the images of its groups of 4 tables: the packed ones (img, what goes back to HBM) and BYTE
    // images (bimg: one byte per observation bit, seats in absolute order — rollout_common.hpp), so that a lane's 32
    // output bytes are two 16-byte LDS reads and one rotate per dword instead of 8 nibble extractions, 8 multiplies
    // and 8 masks.  Per sub-step: apply the command (one call = one bit + one byte, or a re-deal), then copy.
    const GroupLane gl = make_group_lane();
    const MaskLane ml = make_mask_lane();
    const ByteLane bl = make_byte_lane();
    const DealLane dl = make_deal_lane();
    constexpr int NG = TPB / 4;              // groups of 4 consecutive tables
    constexpr int GPW = (NG + NE - 1) / NE;  // groups per emit wave
    const bool head = (gl.r < 4) && (gl.ch == 0);  // one lane per row does the row's bookkeeping
    const int rr = (gl.r < 4) ? gl.r : 3;
    int sub = 0;            // s % substeps
    int64_t row0 = table0;  // first Transition row of this workgroup at macro-step s / substeps
    int left[GPW];          // rows of each group that exist (0..4)
#pragma unroll
    for (int k = 0; k < GPW; k++) {
      const int g = (wave - E0) + k * NE;
      int64_t rem = (g < NG && !(A.debug & 1)) ? A.n - (table0 + 4 * g) : 0;
      left[k] = (int)max((int64_t)0, min((int64_t)4, rem));
      if (g < NG) bimg_build(img + 4 * g * TABLE_BYTES, bimg + 4 * g * BROW, gl, bl);
    }
    wave_lds_order();
    // sub-step s-1 applied to group g's images: one call per row, or a freshly dealt board
    auto apply = [&](int g, uint32_t w0, int rows) {
      uint8_t *img_g = img + 4 * g * TABLE_BYTES;
      uint8_t *bimg_g = bimg + 4 * g * BROW;
      const bool is_head = head && (gl.r < rows);
      if (is_head && !(w0 & 0x200u) && (w0 & 0x1FFu)) {
        const int hb = (int)(w0 & 0x1FFu) - 1;
        atomicOr(reinterpret_cast<uint32_t *>(img_g + gl.r * TABLE_BYTES) + (hb >> 5), 1u << (hb & 31));
        uint8_t *brow = bimg_g + gl.r * BROW;
        if (hb < BTAIL) {
          brow[hb] = 1;
        } else {  // the last bid's 12 bytes live in the four observer tails
#pragma unroll
          for (int q = 0; q < 4; q++) brow[hb + 64 * q] = 1;
        }
      }
      uint64_t dealm = __ballot(is_head && (w0 & 0x200u));
      while (dealm) {  // ~1 table in 11 per sub-step
        const int l = __ffsll((unsigned long long)dealm) - 1;  // lane 15*q holds row q's command
        dealm &= dealm - 1ull;
        const int q = l / 15;
        const uint32_t wq = __builtin_amdgcn_readlane(w0, l);
        const uint32_t *e = &ring[4 * g + q][(wq >> 16) & 15u][0];
        deal_hands(img_g + q * TABLE_BYTES, e, c);
        deal_bytes_hands(bimg_g + q * BROW, e, c, dl);
      }
      wave_lds_order();
    };
    // FAST PATH (substeps == 1, every group of this wave complete, obs + mask requested — the BASELINE
    // configuration): per-lane output pointers advanced by a constant, no per-step emit / tail / pointer
    // selection.  The slot of the post-rollout state (s == total) is left to the general code.
    LDS_BARRIER();  // stage 0
    bool fast = (A.substeps == 1) && A.out.obs && A.out.legal_action_mask && !(A.debug & ~256);
#pragma unroll
    for (int k = 0; k < GPW; k++) fast = fast && (left[k] == 4 || left[k] == 0);
    int bi0 = 0;  // first batch the general loop still has to process
    if (fast) {
      uint8_t *optr[GPW];
      uint32_t *mptr[GPW];
#pragma unroll
      for (int k = 0; k < GPW; k++) {
        const int g = (wave - E0) + k * NE;
        optr[k] = A.out.obs + (table0 + 4 * g) * BRL_OBS_SIZE + bl.out_off;
        mptr[k] = reinterpret_cast<uint32_t *>(A.out.legal_action_mask + (table0 + 4 * g) * BRL_NUM_ACTIONS) + c.lane;
      }
      const int64_t ostep = A.n * BRL_OBS_SIZE, mstep = A.n * BRL_NUM_ACTIONS;
      const bool olane = gl.r < 4;
      for (; bi0 < nbatch; bi0++) {
        const int bstart = ws_bstart(bi0), blen = ws_blen(bi0);
        if (bstart + blen > total) break;  // the batch holding slot `total` goes through the general loop
        LDS_BARRIER();
        for (int j = 0; j < blen; j++) {
          const uint32_t(*cs)[CMD_WORDS] = cmd[bi0 & 3][j];
#pragma unroll
          for (int k = 0; k < GPW; k++) {
            if (left[k] == 0) continue;
            const int g = (wave - E0) + k * NE;
            const uint32_t w0 = cs[4 * g + rr][0];
            apply(g, w0, 4);
            uint4 q0, q1;
            byte_chunk_load(bimg + 4 * g * BROW, (w0 >> 10) & 3u, bl, q0, q1);
            const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
            const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
            if (olane) byte_chunk_store(q0, q1, (w0 >> 10) & 3u, (w0 >> 12) & 15u, optr[k], bl);
            if (ml.active) *mptr[k] = mask_dword(la, lb, ml);
            optr[k] += ostep;
            mptr[k] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(mptr[k]) + mstep);
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
        const uint32_t(*cs)[CMD_WORDS] = cmd[bi & 3][j];
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          const int g = (wave - E0) + k * NE;
          if (left[k] <= 0) continue;
          const uint32_t w0 = cs[4 * g + rr][0];
          apply(g, w0, left[k]);
          if (!emit) continue;
          uint4 q0, q1;
          byte_chunk_load(bimg + 4 * g * BROW, (w0 >> 10) & 3u, bl, q0, q1);
          const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
          const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
          // ---- the 4 observation rows: two 16-B stores per lane
          if (gl.r < left[k] && obs_base)
            byte_chunk_store(q0, q1, (w0 >> 10) & 3u, (w0 >> 12) & 15u,
                             obs_base + (rowb + 4 * g) * BRL_OBS_SIZE + bl.out_off, bl);
          // ---- the 4 mask rows
          if (mask_base) {
            uint8_t *mdst = mask_base + (rowb + 4 * g) * BRL_NUM_ACTIONS;
            if (left[k] >= 4) {  // 152 contiguous bytes, one dword per lane
              if (ml.active) reinterpret_cast<uint32_t *>(mdst)[c.lane] = mask_dword(la, lb, ml);
            } else {  // ragged tail of the batch: row by row
              for (int q = 0; q < left[k]; q++) {
                uint64_t lq = *reinterpret_cast<const uint64_t *>(&cs[4 * g + q][2]);
                emit_mask_row(lq, mdst + q * BRL_NUM_ACTIONS, c);
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
    LDS_BARRIER();
  }
#ifdef BRL_TIMING
  if (c.lane == 0 && A.terminated_count) {  // timing build only: terminated_count doubles as a dump buffer
    unsigned long long *d = A.terminated_count + ((size_t)blockIdx.x * NW + wave) * 2;
    d[0] = __builtin_amdgcn_s_memtime() - t_begin;
    d[1] = t_wait;
    if (A.debug & 256) {  // timeline: per-barrier arrival / release stamps after the per-wave summary area
      unsigned long long *tl = A.terminated_count + (size_t)gridDim.x * NW * 2 + ((size_t)blockIdx.x * NW + wave) * 32;
      for (int q = 0; q < 16; q++) { tl[2 * q] = (q < t_nb) ? t_arr[q] : 0; tl[2 * q + 1] = (q < t_nb) ? t_rel[q] : 0; }
    }
  }
#endif
  __syncthreads();
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    if (tb < A.n) A.state[table0 * 16 + i] = img64[i];
  }
}
