// rollout_flow.hpp — the fused random-policy rollout, flag-synchronised ("flow") variant.
// Included by brl_kernels.hip after k_rollout_ws (same roles, same LDS images, same commands);
// what changes is the hand-off between the waves of a workgroup:
//   k_rollout_ws  : one s_barrier per command batch — every wave waits for the slowest one, and the
//                   first stores leave only after the first batches have been published.
//   k_rollout_flow: no barrier in the loop.  The logic wave posts slot s into a ring of FL_CR command
//                   slots and bumps `f_posted`; every follower wave consumes slots at its own pace and
//                   publishes `f_done[wave]`; the logic wave only stalls when the ring is full.
// The per-table dependency chain (state(t+1) needs state(t)) is what bounds the launch: a single wave
// issues about one instruction every 4-5 cycles whatever its type, so every instruction on the logic
// wave is paid num_steps times in sequence.  With substeps == 1 the logic wave therefore runs a
// MINIMAL transition on a chain-friendly packed word (`fast_step`, ~25 VALU ops) and posts only that
// state; a PREP wave (lane = table) shadows it with the full legacy step and turns each posted state
// into the 16-byte command the loader / scorer / emit waves consume (history bit, legal mask, seat,
// vulnerability nibble, action, n_legal).  substeps > 1 or a caller-supplied finished table (all-True
// mask) use the legacy loop on the logic wave, which posts commands itself.
// LDS operations of one wave execute in order and the LDS unit of a CU is one in-order pipe, so a
// flag written after the data is seen after the data; the fences below only pin the compiler.
// Every spin is bounded (FL_SPIN_MAX polls): on a protocol bug the workgroup raises `f_abort`, all
// loops fall through and the launch ends (with garbage that the parity tests catch) instead of hanging.
#pragma once

constexpr int FL_CR = 16;            // command ring, slots (two scorer batches)
constexpr int FL_DR = 32;            // action-draw ring, slots
constexpr int FL_SPIN_MAX = 1 << 18;

__device__ __forceinline__ int fl_ld(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void fl_st(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void fl_order() { asm volatile("" ::: "memory"); }

// ---- chain-friendly auction state of the logic wave (fast mode) -----------------------------------
// d: [8:0] dealer + turn (seat = low 2 bits) | [14:9] rem = 35 - lb1 | [16:15] last bidder seat |
//    [19:17] e = doubling state: dblst | own << 2 (dblst 0 none / 1 X / 2 XX or "no bid yet"; own = the
//    player to act is on the last bidder's side); X / XX is legal iff e is 0 or 5 |
//    [22:20] pass count, +1 once a bid exists  => the auction is over iff bit 22 is set.
constexpr uint32_t FD_REM = 9, FD_LBSEAT = 15, FD_E = 17, FD_PASS = 20, FD_TERM = 22;
constexpr uint32_t FL_TAG_VALID = 0x80000u;  // ring entry word 9 = sc_bits | (valid | board & 0x7FFFF) << 12

__device__ __forceinline__ uint32_t fast_from_legacy(uint32_t sc, uint32_t sch) {
  const uint32_t lb1 = bits(sc, SC_LB1, 6), st = bits(sc, SC_DEALER, 2) + bits(sch, SCH_TURN, 9);
  const uint32_t has = lb1 != 0u, x = bits(sc, SC_X, 1), xx = bits(sc, SC_XX, 1);
  const uint32_t own = ((bits(sc, SC_LBSEAT, 2) ^ st) & 1u) ^ 1u;
  const uint32_t e = has ? ((x + xx) | (own << 2)) : 2u;
  return (st & 0x1FFu) | ((35u - lb1) << FD_REM) | (bits(sc, SC_LBSEAT, 2) << FD_LBSEAT) | (e << FD_E) |
         ((bits(sc, SC_PASS, 3) + has) << FD_PASS);
}

// one call by the player to act, drawn uniformly from the legal ones with the 32-bit draw u (same choice as
// lean_random_step: the k-th legal call in ascending order, k = mulhi(u, n_legal))
__device__ __forceinline__ uint32_t fast_step(uint32_t d, uint32_t u) {
  const uint32_t rem = __builtin_amdgcn_ubfe(d, FD_REM, 6), e = __builtin_amdgcn_ubfe(d, FD_E, 3);
  const uint32_t dbl = __builtin_amdgcn_ubfe(0x21u, e, 1);
  const uint32_t n = rem + dbl + 1u;  // pass + rem bids + at most one of X / XX
  const uint32_t k = __umulhi(u, n);
  const int kb = (int)(k - dbl);      // >= 1: the kb-th bid above the last one
  const uint32_t d1 = d + 1u;         // next seat
  const uint32_t d_pass = (d1 ^ (4u << FD_E)) + (1u << FD_PASS);
  const uint32_t d_dbl = (((d1 + (1u << FD_E)) ^ (4u << FD_E)) & ~(7u << FD_PASS)) | (1u << FD_PASS);
  const uint32_t d_bid = (d1 & 0x1FFu) | ((rem - (uint32_t)kb) << FD_REM) | ((d & 3u) << FD_LBSEAT) | (1u << FD_PASS);
  uint32_t dn = (k == 0u) ? d_pass : d_dbl;
  dn = (kb > 0) ? d_bid : dn;
  return dn;
}

// ---- byte images of the emit waves ----------------------------------------------------------------
// Per table BROW bytes: [0, 416) = observation bytes 0..415 with the seats of every 4-byte group in ABSOLUTE
// order (bytes 0..3, the vulnerability, are unused: they come with the command); then for each observer seat
// a 64-byte tail = observation bytes 416..479 as that seat sees them: the last bid's 12 history bytes
// (absolute order, replicated) followed by the seat's own 52 hand bytes.
constexpr int BTAIL = 416, BROW = BTAIL + 4 * 64;

struct ByteLane {
  uint32_t src_off;   // this lane's 32 source bytes within the group's 4 byte images (observer seat 0)
  uint32_t tail_sel;  // all-ones for chunks 13 / 14 (observer-specific tail)
  uint32_t rot_a;     // 24 where dwords 0..2 hold history (rotate by the observer's seat), else 0
  uint32_t rot_b;     // same for dwords 3..7
  bool vul;           // chunk 0: dword 0 is the vulnerability nibble of the command
  uint32_t out_off;   // this lane's 32 output bytes within the group's 4 rows
};

__device__ __forceinline__ ByteLane make_byte_lane() {
  ByteLane b;
  const int lane = (int)(threadIdx.x & 63u);
  const int r = lane / 15, ch = lane - r * 15, rr = (r < 4) ? r : 0;
  b.src_off = (uint32_t)(rr * BROW + ((ch <= 12) ? 32 * ch : BTAIL + 32 * (ch - 13)));
  b.tail_sel = (ch >= 13) ? 0xFFFFFFFFu : 0u;
  b.rot_a = (ch <= 13) ? 24u : 0u;
  b.rot_b = (ch <= 12) ? 24u : 0u;
  b.vul = (ch == 0);
  b.out_off = (uint32_t)(rr * 480 + ch * 32);
  return b;
}

__device__ __forceinline__ void expand32(uint32_t word, uint4 &lo, uint4 &hi) {  // 32 bits -> 32 bytes of 0/1
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = __umul24((word >> (4 * i)) & 0xFu, 0x204081u) & 0x01010101u;
  lo = make_uint4(d[0], d[1], d[2], d[3]);
  hi = make_uint4(d[4], d[5], d[6], d[7]);
}

// byte images of a group of 4 tables from their packed images (once per launch)
__device__ __forceinline__ void bimg_build(const uint8_t *img_group, uint8_t *bimg_group, const GroupLane &g,
                                           const ByteLane &b) {
  if (g.r >= 4) return;
  const uint32_t a = *reinterpret_cast<const uint32_t *>(img_group + g.hist_off);
  uint4 lo, hi;
  if (g.ch <= 12) {
    expand32(a, lo, hi);
    uint4 *dst = reinterpret_cast<uint4 *>(bimg_group + b.src_off);
    dst[0] = lo;
    dst[1] = hi;
  } else {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint64_t H = *reinterpret_cast<const uint64_t *>(img_group + g.hand_off + s * 8);
      const uint32_t hv = (g.ch == 13) ? (uint32_t)(H << 8) : (uint32_t)(H >> 24);
      expand32((a & g.keep_hist) | (hv & g.keep_hand), lo, hi);
      uint4 *dst = reinterpret_cast<uint4 *>(bimg_group + b.src_off + 64 * s);
      dst[0] = lo;
      dst[1] = hi;
    }
  }
}

// a freshly dealt board in one table's byte image: no history, the four hands from the LUT key
__device__ __forceinline__ void deal_bytes(uint8_t *brow, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3,
                                           const LaneConst &c) {
  if (c.lane < BTAIL / 16) *reinterpret_cast<uint4 *>(brow + 16 * c.lane) = make_uint4(0u, 0u, 0u, 0u);
  else if (c.lane < BTAIL / 16 + 4) *reinterpret_cast<uint4 *>(brow + BTAIL + 64 * (c.lane - BTAIL / 16)) = make_uint4(0u, 0u, 0u, 0u);
  wave_lds_order();
  const uint32_t ksel = (c.dsuit == 0) ? q0 : ((c.dsuit == 1) ? q1 : ((c.dsuit == 2) ? q2 : q3));
  const uint32_t owner = (ksel >> c.dshift) & 3u;
  if (c.lane < 52) {
#pragma unroll
    for (int s = 0; s < 4; s++) brow[BTAIL + 64 * s + 12 + c.lane] = (uint8_t)(owner == (uint32_t)s);
  }
}

__device__ __forceinline__ void byte_chunk_load(const uint8_t *bimg_group, uint32_t seat, const ByteLane &b, uint4 &q0,
                                                uint4 &q1) {
  const uint4 *src = reinterpret_cast<const uint4 *>(bimg_group + b.src_off + ((seat << 6) & b.tail_sel));
  q0 = src[0];
  q1 = src[1];
}

__device__ __forceinline__ void byte_chunk_store(uint4 q0, uint4 q1, uint32_t seat, uint32_t vulnib, uint8_t *dst,
                                                 const ByteLane &b) {
  const uint32_t ra = (seat << 3) & b.rot_a, rb = (seat << 3) & b.rot_b;
  // relative seat j = absolute seat (observer + j) & 3: rotate every 4-byte group right by `seat` bytes
  uint32_t d0 = __builtin_amdgcn_alignbit(q0.x, q0.x, ra);
  const uint32_t vd = __umul24(vulnib, 0x204081u) & 0x01010101u;
  d0 = b.vul ? vd : d0;
  uint4 *o = reinterpret_cast<uint4 *>(dst);
  o[0] = make_uint4(d0, __builtin_amdgcn_alignbit(q0.y, q0.y, ra), __builtin_amdgcn_alignbit(q0.z, q0.z, ra),
                    __builtin_amdgcn_alignbit(q0.w, q0.w, rb));
  o[1] = make_uint4(__builtin_amdgcn_alignbit(q1.x, q1.x, rb), __builtin_amdgcn_alignbit(q1.y, q1.y, rb),
                    __builtin_amdgcn_alignbit(q1.z, q1.z, rb), __builtin_amdgcn_alignbit(q1.w, q1.w, rb));
}

// min over f_done[first .. first+count-1] (count <= 16), uniform over the wave
__device__ __forceinline__ int fl_min_done(const int *f_done, int first, int count, int lane) {
  int v = fl_ld(&f_done[first + (lane & 15) % count]);
  v = min(v, __shfl_xor(v, 1, 64));
  v = min(v, __shfl_xor(v, 2, 64));
  v = min(v, __shfl_xor(v, 4, 64));
  v = min(v, __shfl_xor(v, 8, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

#ifdef BRL_TIMING  // scripts/timing_flow.py: terminated_count doubles as a per-wave stamp buffer (48 slots per wave)
#define FL_STAMP(k)                                                                                                   \
  do {                                                                                                                \
    if (c.lane == 0 && A.terminated_count && (k) < 48)                                                                \
      A.terminated_count[((size_t)blockIdx.x * NW + wave) * 48 + (k)] = __builtin_amdgcn_s_memtime();                 \
  } while (0)
#else
#define FL_STAMP(k) do { } while (0)
#endif

// wave roles
constexpr int FW_LOGIC = 0, FW_LOADER = 1, FW_SCORER = 2, FW_PREP = 3, FW_EMIT0 = 4;

template <int TPB, int NW>
__global__ __launch_bounds__(NW * 64) void k_rollout_flow(RolloutArgs A) {
  static_assert(TPB <= 32 && NW >= 5 && NW <= 16, "logic + loader + scorer + prep + >=1 emit wave");
  static_assert(TPB % 4 == 0, "emit waves write 4 consecutive tables per instruction");
  constexpr int NE = NW - FW_EMIT0;
  constexpr int B = WS_BATCH;
  __shared__ __attribute__((aligned(16))) uint8_t img[TPB * TABLE_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t bimg[TPB * BROW];  // byte images (emit waves)
  __shared__ __attribute__((aligned(16))) uint32_t cmd[FL_CR][TPB][CMD_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t ring[TPB][WS_RING][RING_WORDS];
  __shared__ __attribute__((aligned(8))) uint2 spost[FL_CR][TPB];  // fast mode: (d, static word) of state s
  __shared__ uint32_t udraw[FL_DR][TPB];
  __shared__ int f_state;    // states posted by the logic wave (fast mode)
  __shared__ int f_posted;   // command slots posted (prep wave; the logic wave in legacy mode)
  __shared__ int f_draws;    // action draws produced by the loader wave
  __shared__ int f_mode;     // 0 undecided, 1 fast (logic -> prep -> followers), 2 legacy (logic -> followers)
  __shared__ int f_abort;
  __shared__ int f_done[16]; // per wave: command slots consumed ([FW_PREP]: states consumed)
  __shared__ float s_neglog[BRL_NUM_ACTIONS + 2];
  const int tid = (int)threadIdx.x;
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LaneConst c = make_lane_const();
  const int64_t table0 = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x) * TPB;
  FL_STAMP(0);
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    img64[i] = (tb < A.n) ? A.state[table0 * 16 + i] : 0ull;
  }
  if (tid <= BRL_NUM_ACTIONS) s_neglog[tid] = A.neg_log_n[tid];
  const int total = A.T * A.substeps;  // sub-steps; command slots are s = 0..total
  const int tl = c.lane;               // logic / loader / scorer / prep: lane = table
  const int tls = (tl < TPB) ? tl : 0;
  const bool valid = (tl < TPB) && (table0 + tl < A.n);
  const uint64_t env_id = A.env_offset + (uint64_t)(table0 + tl);
  uint64_t ctr_word = 0;
  if (wave == FW_LOADER && valid) ctr_word = A.state[(table0 + tl) * 16 + W_CTR];
  // action draws: Philox is state-independent, the loader wave produces them ahead of the logic wave
  uint32_t rbk[4] = {0, 0, 0, 0};
  uint32_t rbk_idx = 0xFFFFFFFFu;
  auto draw_slot = [&](int d) {  // draw of command slot d -> udraw[d % FL_DR]
    const uint32_t draw = A.draw_base + (uint32_t)d;
    if ((draw >> 2) != rbk_idx) {
      rbk_idx = draw >> 2;
      philox4x32_10((uint32_t)env_id, rbk_idx, STREAM_ACTION, (uint32_t)(env_id >> 32), A.g.k0, A.g.k1, rbk);
    }
    const uint32_t sel = draw & 3u;
    if (tl < TPB) udraw[d & (FL_DR - 1)][tl] = (sel == 0) ? rbk[0] : ((sel == 1) ? rbk[1] : ((sel == 2) ? rbk[2] : rbk[3]));
  };
  int ndraw = 0;  // (loader) draws produced
  if (wave == FW_LOADER) {
    const int first = min(total + 1, 4 - (int)(A.draw_base & 3u));  // the rest of the first Philox block
    for (; ndraw < first; ndraw++) draw_slot(ndraw);
    if (tl < TPB) {  // no board in the ring yet: clear the valid bit of every entry's tag
#pragma unroll
      for (int k = 0; k < WS_RING; k++) ring[tl][k][9] = 0u;
    }
  }
  if (tid == 0) {
    f_state = 0;
    f_posted = 0;
    f_mode = 0;
    f_abort = 0;
    f_draws = min(total + 1, 4 - (int)(A.draw_base & 3u));
  }
  if (tid < 16) f_done[tid] = 0;
  __syncthreads();  // images and the first draws are in LDS; the ring follows (entry tags)
  FL_STAMP(1);

  auto aborted = [&]() { return fl_ld(&f_abort) != 0; };

  if (wave == FW_LOADER) {
    // ------------------------------------------------------------------ loader wave
    uint32_t nb = 0, nb0 = 0, pbase = 0, pcount = 0, pidx[3] = {0, 0, 0}, pscb[3] = {0, 0, 0};
    int4 pk[3], pv[3];
    if (valid) {
      nb0 = (uint32_t)(ctr_word >> 32) + 1u;
      nb = nb0;
      pbase = nb;
      pcount = 2;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        board_params(A.g, env_id, nb + (uint32_t)k, A.lut.len, pidx[k], pscb[k]);
        pk[k] = A.lut.keys[pidx[k]];
        pv[k] = A.lut.values[pidx[k]];
      }
      nb += 2u;
    }
    int lo = 0;             // command slots whose deals have been counted (= f_done[FW_LOADER])
    uint32_t released = 0;  // boards dealt in slots < lo: their ring slots are free again
    int idle = 0;
    int it = 0;
    for (;;) {
      bool progress = false;
      FL_STAMP(2 + it);
      it++;
      // 1. draws: up to 8 more, never more than FL_DR - 3 ahead of the commands posted (the prep wave reads
      //    the draw of slot s - 1 while it builds command s; the logic wave is ahead of it)
      {
        const int lim = min(total + 1, fl_ld(&f_posted) + FL_DR - 3);
        const int upto = min(lim, ndraw + 8);
        if (ndraw < upto) {
          for (; ndraw < upto; ndraw++) draw_slot(ndraw);
          fl_order();
          if (c.lane == 0) fl_st(&f_draws, ndraw);
          progress = true;
        }
      }
      // 2. commit the boards fetched in the previous round (their loads have landed); the tag goes last
      if (__any(pcount > 0)) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          if ((uint32_t)k < pcount) {
            const uint32_t b = pbase + (uint32_t)k;
            uint32_t *e = &ring[tls][b % WS_RING][0];
            uint4 *dst = reinterpret_cast<uint4 *>(e);
            dst[0] = make_uint4((uint32_t)pk[k].x, (uint32_t)pk[k].y, (uint32_t)pk[k].z, (uint32_t)pk[k].w);
            dst[1] = make_uint4((uint32_t)pv[k].x, (uint32_t)pv[k].y, (uint32_t)pv[k].z, (uint32_t)pv[k].w);
            e[8] = pidx[k];
            fl_order();
            __hip_atomic_store(&e[9], pscb[k] | ((FL_TAG_VALID | (b & 0x7FFFFu)) << 12), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        pcount = 0;
        progress = true;
      }
      // 3. boards whose slots every reader (scorer + emit waves) is done with
      {
        const int R = fl_min_done(f_done, FW_SCORER, NW - FW_SCORER, c.lane);
        if (R > lo) {
          uint32_t dealt = 0;
          for (int s = lo; s < R; s++) dealt += (cmd[s & (FL_CR - 1)][tls][0] >> 9) & 1u;
          released += dealt;
          lo = R;
          fl_order();
          if (c.lane == 0) fl_st(&f_done[FW_LOADER], lo);
          progress = true;
        }
      }
      // 4. keep WS_RING boards ahead of the released ones: at most 3 fetches in flight per table
      {
        const uint32_t want = nb0 + (uint32_t)WS_RING + released;
        const bool need = valid && (int32_t)(want - nb) > 0 && lo <= total;
        if (__any(need)) {
          if (need) {
            pbase = nb;
            pcount = min(3u, want - nb);
#pragma unroll
            for (int k = 0; k < 3; k++) {
              if ((uint32_t)k < pcount) {
                board_params(A.g, env_id, nb + (uint32_t)k, A.lut.len, pidx[k], pscb[k]);
                pk[k] = A.lut.keys[pidx[k]];
                pv[k] = A.lut.values[pidx[k]];
              }
            }
            nb += pcount;
          }
          progress = true;
        }
      }
      if (lo > total && ndraw > total) break;
      if (progress) {
        idle = 0;
      } else {
        __builtin_amdgcn_s_sleep(2);
        if (++idle > FL_SPIN_MAX) { if (c.lane == 0) fl_st(&f_abort, 1); break; }
        if (aborted()) break;
      }
    }
  } else if (wave == FW_LOGIC) {
    // ------------------------------------------------------------------ logic wave
    uint32_t sc, sch, lut, bctr;
    {
      const uint2 *p = reinterpret_cast<const uint2 *>(img + tls * TABLE_BYTES);
      uint2 a = p[W_SC], d = p[W_CTR];
      sc = a.x; sch = a.y; lut = d.x; bctr = d.y;
    }
    const bool fastmode = (A.substeps == 1) && !__any((tl < TPB) && bits(sc, SC_MASKALL, 1)) && !(A.debug & 4);
    if (c.lane == 0) fl_st(&f_mode, fastmode ? 1 : 2);
    __builtin_amdgcn_s_setprio(3);  // the critical chain wins issue arbitration on its SIMD
    bool dead = false;
    // (LUT row, fresh scalars | tag) of the NEXT board of this slot: read ahead of the deal that uses it, checked
    // against the board's tag when it is used (the loader may not have filled the entry yet)
    uint32_t rslot = (bctr + 1u) % WS_RING;
    auto ring_peek = [&]() {
      const uint64_t v = __hip_atomic_load(reinterpret_cast<const uint64_t *>(&ring[tls][rslot][8]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
      return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    uint2 nxt = ring_peek();
    auto ring_take = [&](bool deal) {  // make sure `nxt` is board bctr + 1 for every dealing lane
      int spins = 0;
      while (__any(deal && (nxt.y >> 12) != (FL_TAG_VALID | ((bctr + 1u) & 0x7FFFFu)))) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        nxt = ring_peek();
        if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
      }
    };
    int draws_seen = fl_ld(&f_draws);  // >= 1
    int space_upto = FL_CR;            // slots < space_upto may be written
    uint32_t un = udraw[0][tls];
    auto next_draw = [&](int s) {  // draw of slot s + 1, off the chain
      if (s + 1 >= draws_seen) {
        int spins = 0;
        while ((draws_seen = fl_ld(&f_draws)) <= s + 1) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
        }
      }
      un = udraw[(s + 1) & (FL_DR - 1)][tls];
    };
    if (fastmode) {
      // ---- fast mode: minimal transition, the prep wave builds the commands
      uint32_t d = fast_from_legacy(sc, sch);
      uint32_t stw = (sc & 0x0A000FFFu) | ((bctr % WS_RING) << 28);  // board constants | TERM | ILLEGAL | ring slot
      for (int s = 0; s <= total && !dead; s++) {
        const uint32_t u = un;
        if (s < total) next_draw(s);
        if (s >= space_upto) {  // the state ring is full: wait for the prep wave
          int spins = 0;
          while ((space_upto = fl_ld(&f_done[FW_PREP]) + FL_CR) <= s) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
          }
        }
        if (tl < TPB) spost[s & (FL_CR - 1)][tl] = make_uint2(d, stw);
        fl_order();
        if (c.lane == 0) fl_st(&f_state, s + 1);
        FL_STAMP(2 + s);
        if (s == total) break;
        d = fast_step(d, u);
        const uint32_t term = (d >> FD_TERM) & 1u;
        stw = (stw & ~(1u << SC_TERM)) | (term << SC_TERM);
        const bool deal = valid && term;
        if (__any(deal)) {
          ring_take(deal);
          if (deal) {  // A5 post-step half of auto_reset (src/utils.py:45-55): next board from the ring
            stw = (nxt.y & 0xFFFu) | (stw & ((1u << SC_TERM) | (1u << SC_ILLEGAL))) | (rslot << 28);
            d = (nxt.y & 3u) | (35u << FD_REM) | (2u << FD_E);
            lut = nxt.x;
            bctr += 1u;
            rslot = (rslot + 1u == (uint32_t)WS_RING) ? 0u : rslot + 1u;
            nxt = ring_peek();
          }
        }
      }
    } else {
      // ---- legacy mode: the full step on the logic wave, which posts the commands itself
      uint32_t pend = 0, pend_act = 0, pend_sc = 0, term_any = 0;
      int sub = 0;
      for (int s = 0; s <= total && !dead; s++) {
        const uint32_t u = un;
        if (s < total) next_draw(s);
        uint32_t nsc = sc, nsch = sch;
        const LeanStep st = lean_random_step(nsc, nsch, u);
        if (s >= space_upto) {  // the command ring is full: wait for the slowest follower
          int spins = 0;
          for (;;) {
            space_upto = fl_min_done(f_done, 1, NW - 1, c.lane) + FL_CR;
            if (s < space_upto) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
          }
        }
        if (tl < TPB) {  // command slot s: what sub-step s-1 did + how state s looks
          uint32_t w0 = pend | ((uint32_t)st.seat << 10) | (vul_nibble_sc(sc, st.seat) << 12);
          uint32_t w3 = ((uint32_t)(st.legal >> 32) & 63u) | (pend_act << 8);
          *reinterpret_cast<uint4 *>(&cmd[s & (FL_CR - 1)][tl][0]) = make_uint4(w0, pend_sc, (uint32_t)st.legal, w3);
        }
        fl_order();
        if (c.lane == 0) fl_st(&f_posted, s + 1);
        FL_STAMP(2 + s);
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
        pend = st.hb1 | ((uint32_t)deal << 9) | (rslot << 16) | ((uint32_t)st.seat << 21) | ((uint32_t)st.n_legal << 23);
        if (__any(deal)) {
          ring_take(deal);
          if (deal) {
            sc = (nxt.y & 0xFFFu) | (sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL)));
            sch = 0;
            lut = nxt.x;
            bctr += 1u;
            rslot = (rslot + 1u == (uint32_t)WS_RING) ? 0u : rslot + 1u;
            nxt = ring_peek();
          }
        }
        if (last && A.substeps > 1) sc = (sc & ~(1u << SC_TERM)) | (term_any << SC_TERM);  // src/utils.py:127
      }
      if (tl < TPB) reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES)[W_SC] = make_uint2(sc, sch);
    }
    if (dead && c.lane == 0) {
      fl_st(&f_abort, 1);
      fl_st(&f_state, total + 1);
      fl_st(&f_posted, total + 1);
    }
    if (tl < TPB) reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES)[W_CTR] = make_uint2(lut, bctr);
  } else if (wave == FW_PREP) {
    // ------------------------------------------------------------------ prep wave (fast mode)
    // Shadows the logic wave with the full legacy step: same draws, and on a re-deal the fresh scalars come
    // with the posted state, so it never touches the board ring.  Command slot s = what sub-step s-1 did +
    // how state s looks (legal mask, observer seat, vulnerability nibble).
    bool dead = false;
    int mode = 0;
    {
      int spins = 0;
      while ((mode = fl_ld(&f_mode)) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
      }
    }
    if (mode == 1 && !dead) {
      uint32_t sc, sch;
      {
        const uint2 a = reinterpret_cast<const uint2 *>(img + tls * TABLE_BYTES)[W_SC];
        sc = a.x; sch = a.y;
      }
      uint32_t pend = 0, pend_act = 0, pend_sc = 0;
      int seen = 0;             // states known to be posted
      int space_upto = FL_CR;   // command slots < space_upto may be written
      for (int s = 0; s <= total && !dead; s++) {
        if (s >= seen) {
          int spins = 0;
          while ((seen = fl_ld(&f_state)) <= s) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
          }
          if (aborted()) dead = true;
          if (dead) break;
          fl_order();
        }
        if (s > 0) {  // sub-step s-1 on the shadow state
          const uint32_t u = udraw[(s - 1) & (FL_DR - 1)][tls];
          const uint32_t stw = spost[s & (FL_CR - 1)][tls].y;
          const LeanStep st = lean_random_step(sc, sch, u);
          pend_sc = sc;
          pend_act = (uint32_t)st.action;
          const bool deal = valid && st.term;
          pend = st.hb1 | ((uint32_t)deal << 9) | ((stw >> 28) << 16) | ((uint32_t)st.seat << 21) | ((uint32_t)st.n_legal << 23);
          if (deal) {  // the board the logic wave dealt: fresh scalars | TERM | ILLEGAL
            sc = stw & 0x0A000FFFu;
            sch = 0;
          }
        }
        fl_order();
        if (c.lane == 0) fl_st(&f_done[FW_PREP], s + 1);  // state s has been read
        // how state s looks
        const uint32_t seat = (bits(sc, SC_DEALER, 2) + bits(sch, SCH_TURN, 9)) & 3u;
        uint64_t legal;
        {
          const uint32_t lb1 = bits(sc, SC_LB1, 6);
          const uint32_t own = ((bits(sc, SC_LBSEAT, 2) ^ seat) & 1u) ^ 1u;
          const uint32_t x = bits(sc, SC_X, 1), xx = bits(sc, SC_XX, 1), has = lb1 != 0;
          const uint32_t can_x = has & (own ^ 1u) & (x ^ 1u) & (xx ^ 1u);
          const uint32_t can_xx = has & own & x & (xx ^ 1u);
          legal = ((ALL_ACTIONS >> (3 + lb1)) << (3 + lb1)) | (uint64_t)(1u | (can_x << 1) | (can_xx << 2));
          if (bits(sc, SC_MASKALL, 1)) legal = ALL_ACTIONS;
        }
        if (s >= space_upto) {  // the command ring is full: wait for the slowest follower
          int spins = 0;
          for (;;) {
            space_upto = fl_min_done(f_done, 1, NW - 1, c.lane) + FL_CR;
            if (s < space_upto) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
          }
        }
        if (tl < TPB) {
          const uint32_t w0 = pend | (seat << 10) | (vul_nibble_sc(sc, (int)seat) << 12);
          const uint32_t w3 = ((uint32_t)(legal >> 32) & 63u) | (pend_act << 8);
          *reinterpret_cast<uint4 *>(&cmd[s & (FL_CR - 1)][tl][0]) = make_uint4(w0, pend_sc, (uint32_t)legal, w3);
        }
        fl_order();
        if (c.lane == 0) fl_st(&f_posted, s + 1);
        FL_STAMP(2 + s);
      }
      if (tl < TPB) reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES)[W_SC] = make_uint2(sc, sch);
    }
    if (c.lane == 0) {
      fl_st(&f_done[FW_PREP], total + 1 + FL_CR);
      if (dead) {
        fl_st(&f_abort, 1);
        fl_st(&f_posted, total + 1);
      }
    }
  } else if (wave == FW_SCORER) {
    // ------------------------------------------------------------------ scorer wave
    // Works in batches of WS_BATCH command slots (three passes, see k_rollout_ws).
    __shared__ __attribute__((aligned(16))) uint32_t ev[3][64][8];     // finished boards of this batch
    __shared__ __attribute__((aligned(16))) int acc[WS_BATCH][64][4];  // reward sums by player id per macro-step
    __shared__ uint32_t minfo[WS_BATCH][64];                           // per macro-step: actor, action, n_legal, done
    Tbl ts;
    load_scalars(ts, img + tls * TABLE_BYTES);
    int sub = 0;
    uint32_t cur_info = 0, tcount = 0;
    int64_t row = table0 + tl;
    int4 last_acc = make_int4(reward_of(ts, 0), reward_of(ts, 1), reward_of(ts, 2), reward_of(ts, 3));
    *reinterpret_cast<int4 *>(&acc[0][tl][0]) = make_int4(0, 0, 0, 0);
    const int nbatch = total / B + 1;
    bool dead = false;
    for (int bi = 0; bi < nbatch && !dead; bi++) {
      const int s0 = bi * B, s1 = min(s0 + B, total + 1);
      {
        int spins = 0;
        while (fl_ld(&f_posted) < s1) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
        }
        if (aborted()) dead = true;
        if (dead) break;
      }
      fl_order();
      // ---- pass 1
      int nev = 0, m = 0;
#pragma unroll
      for (int q = 1; q < B; q++) *reinterpret_cast<int4 *>(&acc[q][tl][0]) = make_int4(0, 0, 0, 0);
      uint4 wn = *reinterpret_cast<const uint4 *>(&cmd[s0 & (FL_CR - 1)][tls][0]);
      for (int s = s0; s < s1; s++) {
        const uint4 w = wn;
        wn = *reinterpret_cast<const uint4 *>(&cmd[((s + 1 < s1) ? s + 1 : s) & (FL_CR - 1)][tls][0]);
        if (s == 0) continue;  // cmd slot 0 describes no sub-step
        const int a = (int)((w.w >> 8) & 63u);
        const int seat = (int)((w.x >> 21) & 3u);
        ts.sc = w.y;
        if (sub == 0)
          cur_info = (uint32_t)player_at(ts, seat) | ((uint32_t)a << 2) | (((w.x >> 23) & 63u) << 8);
        note_first_denomination(ts.fd, seat, a);
        if (bits(ts.sc, SC_TERM, 1)) {
          uint4 *e = reinterpret_cast<uint4 *>(&ev[nev][tl][0]);
          e[0] = make_uint4(ts.sc, ts.fd, ts.t0, ts.t1);
          e[1] = make_uint4(ts.t2, (uint32_t)m, 0u, 0u);
          nev++;
          cur_info |= 1u << 14;  // done (G2)
        }
        if (w.x & 0x200u) {
          const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tls][(w.x >> 16) & 15u][4]);
          pack_tricks(ts, vv.x, vv.y, vv.z, vv.w);
          ts.fd = 0;
        }
        if (++sub == A.substeps) {
          sub = 0;
          minfo[m][tl] = cur_info;
          m++;
        }
      }
      // every cmd / ring read of this batch has been issued: release the slots
      fl_order();
      if (c.lane == 0) fl_st(&f_done[FW_SCORER], s1);
      FL_STAMP(2 + 2 * bi);
      // ---- pass 2
      for (int e = 0; e < 3; e++) {
        if (!__any(e < nev)) break;
        if (e < nev) {
          const uint4 *p = reinterpret_cast<const uint4 *>(&ev[e][tl][0]);
          const uint4 e0 = p[0], e1 = p[1];
          Tbl tb;
          tb.sc = e0.x; tb.fd = e0.y; tb.t0 = e0.z; tb.t1 = e0.w; tb.t2 = e1.x;
          terminal_reward(tb);  // A4
          int *ac = &acc[e1.y & (WS_BATCH - 1)][tl][0];
          ac[0] += reward_of(tb, 0); ac[1] += reward_of(tb, 1); ac[2] += reward_of(tb, 2); ac[3] += reward_of(tb, 3);
        }
      }
      // ---- pass 3
      wave_lds_order();
      if (m > 0) last_acc = *reinterpret_cast<const int4 *>(&acc[m - 1][tl][0]);
      {
        const int half = c.lane >> 5;
        const int tq = c.lane & 31;
        const bool vq = (tq < TPB) && (table0 + tq < A.n);
        for (int q0 = 0; q0 < m; q0 += 2) {
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
      {
        int4 carry = (sub != 0) ? *reinterpret_cast<const int4 *>(&acc[m & (WS_BATCH - 1)][tl][0]) : make_int4(0, 0, 0, 0);
        if (m >= WS_BATCH) carry = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int4 *>(&acc[0][tl][0]) = carry;
      }
      FL_STAMP(3 + 2 * bi);
    }
    if (dead && c.lane == 0) fl_st(&f_done[FW_SCORER], total + 1);
    set_rewards(ts, last_acc.x, last_acc.y, last_acc.z, last_acc.w);
    if (A.terminated_count != nullptr) {  // src/roll_out.py:85
      uint32_t v = tcount;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
#ifndef BRL_TIMING
      if (c.lane == 0 && v) atomicAdd(A.terminated_count, (unsigned long long)v);
#endif
    }
    if (tl < TPB) {
      uint2 *p = reinterpret_cast<uint2 *>(img + tl * TABLE_BYTES);
      p[W_FD] = make_uint2(ts.fd, ts.t2);
      p[W_TR] = make_uint2(ts.t0, ts.t1);
      p[W_REW] = make_uint2(ts.r01, ts.r23);
    }
  } else {
    // ------------------------------------------------------------------ emit waves
    // Observation rows are copied from BYTE images (bimg): one byte per observation bit, seats in absolute
    // order, so a lane's 32 output bytes are two 16-byte LDS reads plus one rotate per dword (the observer's
    // seat) instead of 8 nibble extractions + 8 multiplies + 8 masks.  The packed images (img) are kept up to
    // date too (one ds_or per call): they are what goes back to HBM as the table state.
    const GroupLane gl = make_group_lane();
    const MaskLane ml = make_mask_lane();
    const ByteLane bl = make_byte_lane();
    constexpr int NG = TPB / 4;
    constexpr int GPW = (NG + NE - 1) / NE;
    const bool head = (gl.r < 4) && (gl.ch == 0);
    const int rr = (gl.r < 4) ? gl.r : 3;
    int sub = 0;
    int64_t row0 = table0;
    int left[GPW];
#pragma unroll
    for (int k = 0; k < GPW; k++) {
      const int g = (wave - FW_EMIT0) + k * NE;
      int64_t rem = (g < NG && !(A.debug & 1)) ? A.n - (table0 + 4 * g) : 0;
      left[k] = (int)max((int64_t)0, min((int64_t)4, rem));
      if (g < NG) bimg_build(img + 4 * g * TABLE_BYTES, bimg + 4 * g * BROW, gl, bl);
    }
    wave_lds_order();
    int avail = 0;  // command slots known to be posted
    bool dead = false;
    auto wait_slot = [&](int s) {
      if (s < avail) return;
      int spins = 0;
      while ((avail = fl_ld(&f_posted)) <= s) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
      }
      if (aborted()) dead = true;
      fl_order();
    };
    // sub-step s-1 applied to group g's images: one call (history bit) or a freshly dealt board per row
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
      while (dealm) {  // rare: ~1 table in 25 per sub-step
        const int l = __ffsll((unsigned long long)dealm) - 1;  // lane 15*q holds row q's command
        dealm &= dealm - 1ull;
        const int q = l / 15;
        const uint32_t wq = __builtin_amdgcn_readlane(w0, l);
        const uint4 kk = *reinterpret_cast<const uint4 *>(&ring[4 * g + q][(wq >> 16) & 15u][0]);
        deal_image(img_g + q * TABLE_BYTES, kk.x, kk.y, kk.z, kk.w, c);
        deal_bytes(bimg_g + q * BROW, kk.x, kk.y, kk.z, kk.w, c);
      }
      wave_lds_order();
    };
    bool fast = (A.substeps == 1) && A.out.obs && A.out.legal_action_mask;
#pragma unroll
    for (int k = 0; k < GPW; k++) fast = fast && (left[k] == 4 || left[k] == 0);
    int s_next = 0;  // first slot the general loop still has to process
    if (fast) {
      uint8_t *optr[GPW];
      uint32_t *mptr[GPW];
#pragma unroll
      for (int k = 0; k < GPW; k++) {
        const int g = (wave - FW_EMIT0) + k * NE;
        optr[k] = A.out.obs + (table0 + 4 * g) * BRL_OBS_SIZE + bl.out_off;
        mptr[k] = reinterpret_cast<uint32_t *>(A.out.legal_action_mask + (table0 + 4 * g) * BRL_NUM_ACTIONS) + c.lane;
      }
      const int64_t ostep = A.n * BRL_OBS_SIZE, mstep = A.n * BRL_NUM_ACTIONS;
      const bool olane = gl.r < 4;
      // software-pipelined: the command words of slot s+1 (when already posted) are read while slot s is
      // processed, so a slot costs ONE LDS round trip (the image chunks) instead of two
      uint32_t pw0[GPW];
      uint64_t pla[GPW], plb[GPW];
      bool have = false;
      for (; s_next < total; s_next++) {
        wait_slot(s_next);
        if (dead) break;
        const uint32_t(*cs)[CMD_WORDS] = cmd[s_next & (FL_CR - 1)];
        if (!have) {
#pragma unroll
          for (int k = 0; k < GPW; k++) {
            const int g = (wave - FW_EMIT0) + k * NE;
            if (left[k] == 0) continue;
            pw0[k] = cs[4 * g + rr][0];
            pla[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
            plb[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
          }
        }
        uint32_t w0c[GPW];
        uint64_t lac[GPW], lbc[GPW];
#pragma unroll
        for (int k = 0; k < GPW; k++) { w0c[k] = pw0[k]; lac[k] = pla[k]; lbc[k] = plb[k]; }
        have = (s_next + 1 < avail) && (s_next + 1 < total);
        if (have) {
          const uint32_t(*cn)[CMD_WORDS] = cmd[(s_next + 1) & (FL_CR - 1)];
#pragma unroll
          for (int k = 0; k < GPW; k++) {
            const int g = (wave - FW_EMIT0) + k * NE;
            if (left[k] == 0) continue;
            pw0[k] = cn[4 * g + rr][0];
            pla[k] = *reinterpret_cast<const uint64_t *>(&cn[4 * g + ml.qa][2]);
            plb[k] = *reinterpret_cast<const uint64_t *>(&cn[4 * g + ml.qb][2]);
          }
        }
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          if (left[k] == 0) continue;
          const int g = (wave - FW_EMIT0) + k * NE;
          const uint32_t w0 = w0c[k];
          apply(g, w0, 4);
          uint4 q0, q1;
          byte_chunk_load(bimg + 4 * g * BROW, (w0 >> 10) & 3u, bl, q0, q1);
          if (olane && !(A.debug & 8)) byte_chunk_store(q0, q1, (w0 >> 10) & 3u, (w0 >> 12) & 15u, optr[k], bl);
          if (ml.active && !(A.debug & 16)) *mptr[k] = mask_dword(lac[k], lbc[k], ml);
          optr[k] += ostep;
          mptr[k] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(mptr[k]) + mstep);
        }
        fl_order();
        if (c.lane == 0) fl_st(&f_done[wave], s_next + 1);  // behind this wave's cmd / ring reads of the slot
        FL_STAMP(2 + s_next);
      }
      row0 = table0 + (int64_t)s_next * A.n;  // substeps == 1: macro-step index == slot index
    }
    for (int s = s_next; s <= total && !dead; s++) {
      wait_slot(s);
      if (dead) break;
      const bool fin = (s == total);  // the post-rollout state: emitted as last_obs / last_mask
      const bool emit = ((s < total) && (sub == 0)) || (fin && (A.last_obs || A.last_mask));
      uint8_t *obs_base = fin ? A.last_obs : A.out.obs;
      uint8_t *mask_base = fin ? A.last_mask : A.out.legal_action_mask;
      const int64_t rowb = fin ? table0 : row0;
      const uint32_t(*cs)[CMD_WORDS] = cmd[s & (FL_CR - 1)];
#pragma unroll
      for (int k = 0; k < GPW; k++) {
        const int g = (wave - FW_EMIT0) + k * NE;
        if (left[k] <= 0) continue;
        const uint32_t w0 = cs[4 * g + rr][0];
        apply(g, w0, left[k]);
        if (!emit) continue;
        uint4 q0, q1;
        byte_chunk_load(bimg + 4 * g * BROW, (w0 >> 10) & 3u, bl, q0, q1);
        const uint64_t la = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
        const uint64_t lb = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
        if (gl.r < left[k] && obs_base)
          byte_chunk_store(q0, q1, (w0 >> 10) & 3u, (w0 >> 12) & 15u,
                           obs_base + (rowb + 4 * g) * BRL_OBS_SIZE + bl.out_off, bl);
        if (mask_base) {
          uint8_t *mdst = mask_base + (rowb + 4 * g) * BRL_NUM_ACTIONS;
          if (left[k] >= 4) {
            if (ml.active) reinterpret_cast<uint32_t *>(mdst)[c.lane] = mask_dword(la, lb, ml);
          } else {
            for (int q = 0; q < left[k]; q++) {
              uint64_t lq = *reinterpret_cast<const uint64_t *>(&cs[4 * g + q][2]);
              emit_mask_row(lq, mdst + q * BRL_NUM_ACTIONS, c);
            }
          }
        }
      }
      fl_order();
      if (c.lane == 0) fl_st(&f_done[wave], s + 1);
      FL_STAMP(2 + s);
      if (++sub == A.substeps) {
        sub = 0;
        row0 += A.n;
      }
    }
    if (dead && c.lane == 0) fl_st(&f_done[wave], total + 1);
  }
  FL_STAMP(46);
  __syncthreads();
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    if (tb < A.n) A.state[table0 * 16 + i] = img64[i];
  }
  FL_STAMP(47);
}
