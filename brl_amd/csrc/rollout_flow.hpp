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
// Waiting waves sleep (s_sleep FL_NAP = 64 * FL_NAP clocks per poll) so that their polling does not eat the
// CU's instruction issue slots — the waves on the critical chain need them; the two producers the followers
// actually wait for (prep: commands, apply: byte images) ping them awake right after publishing (s_wakeup ends
// every s_sleep of the workgroup; the flag must be visible first, hence the lgkmcnt wait).
#define FL_NAP 8
__device__ __forceinline__ void fl_wake() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory"); }

// ---- chain-friendly auction state of the logic wave (fast mode) -----------------------------------
// d: [8:0] dealer + turn (seat = low 2 bits) | [14:9] rem = 35 - lb1 | [16:15] last bidder seat |
//    [19:17] e = doubling state: dblst | own << 2 (dblst 0 none / 1 X / 2 XX or "no bid yet"; own = the
//    player to act is on the last bidder's side); X / XX is legal iff e is 0 or 5 |
//    [22:20] pass count, +1 once a bid exists  => the auction is over iff bit 22 is set.
constexpr uint32_t FD_REM = 9, FD_LBSEAT = 15, FD_E = 17, FD_PASS = 20, FD_TERM = 22;
constexpr uint32_t FL_TAG_VALID = 0x80000u;  // ring entry word FR_TAG = sc_bits | (valid | board & 0x7FFFF) << 12
// ring entry (64 B): the board's four packed hand words (precomputed per LUT row, k_lut_hands), its DDS values,
// its LUT row and fresh scalars | tag
constexpr int FR_WORDS = 16, FR_HANDS = 0, FR_VALUES = 8, FR_IDX = 12, FR_TAG = 13;

__device__ __forceinline__ uint32_t fast_from_legacy(uint32_t sc, uint32_t sch) {
  const uint32_t lb1 = bits(sc, SC_LB1, 6), st = bits(sc, SC_DEALER, 2) + bits(sch, SCH_TURN, 9);
  const uint32_t has = lb1 != 0u, x = bits(sc, SC_X, 1), xx = bits(sc, SC_XX, 1);
  const uint32_t own = ((bits(sc, SC_LBSEAT, 2) ^ st) & 1u) ^ 1u;
  const uint32_t e = has ? ((x + xx) | (own << 2)) : 2u;
  return (st & 0x1FFu) | ((35u - lb1) << FD_REM) | (bits(sc, SC_LBSEAT, 2) << FD_LBSEAT) | (e << FD_E) |
         ((bits(sc, SC_PASS, 3) + has) << FD_PASS);
}

// back to the packed scalars (sc, sch) of the table state — valid for a live auction with substeps == 1, where
// _step_count == _turn; TERM / ILLEGAL and the board constants ride in the static word
__device__ __forceinline__ void fast_to_legacy(uint32_t d, uint32_t stw, uint32_t &sc, uint32_t &sch) {
  const uint32_t lb1 = 35u - __builtin_amdgcn_ubfe(d, FD_REM, 6), has = lb1 != 0u;
  const uint32_t dbl = __builtin_amdgcn_ubfe(d, FD_E, 2);
  const uint32_t x = has & (uint32_t)(dbl >= 1u), xx = has & (uint32_t)(dbl == 2u);
  const uint32_t pass = __builtin_amdgcn_ubfe(d, FD_PASS, 3) - has;
  const uint32_t turn = ((d & 0x1FFu) - (stw & 3u)) & 0x1FFu;
  sc = (stw & 0x0A000FFFu) | (lb1 << SC_LB1) | (__builtin_amdgcn_ubfe(d, FD_LBSEAT, 2) << SC_LBSEAT) | (x << SC_X) |
       (xx << SC_XX) | (pass << SC_PASS);
  sch = turn | (turn << SCH_STEP);
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

// the same for BOTH images of a table with one set of ballots: packed image (history zero, the four hand
// words) and byte image (history zero; lane L < 52 writes the 4 hand bytes [4j, 4j+4) of seat L / 13's tail,
// j = L % 13; lanes 52..63 clear the 4 x 3 history dwords of the tails)
struct DealLane {
  uint32_t seat;  // whose hand word this lane expands (lanes < 52)
  uint32_t sh;    // bit offset of its 4 cards in that word
  uint32_t off;   // byte offset of the dword it writes within the table's byte image
};
__device__ __forceinline__ DealLane make_deal_lane() {
  DealLane d;
  const uint32_t lane = threadIdx.x & 63u;
  if (lane < 52u) {
    d.seat = lane / 13u;
    d.sh = 4u * (lane - 13u * d.seat);
    d.off = (uint32_t)BTAIL + 64u * d.seat + 12u + d.sh;
  } else {
    const uint32_t q = lane - 52u;
    d.seat = 4u;
    d.sh = 0u;
    d.off = (uint32_t)BTAIL + 64u * (q / 3u) + 4u * (q % 3u);
  }
  return d;
}
__device__ __forceinline__ void deal_both(uint8_t *img, uint8_t *brow, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3,
                                          const LaneConst &c, const DealLane &dl) {
  const uint32_t ksel = (c.dsuit == 0) ? q0 : ((c.dsuit == 1) ? q1 : ((c.dsuit == 2) ? q2 : q3));
  const uint32_t owner = (ksel >> c.dshift) & 3u;
  const bool card = c.lane < 52;
  const uint64_t h0 = __ballot(card && owner == 0u), h1 = __ballot(card && owner == 1u);
  const uint64_t h2 = __ballot(card && owner == 2u), h3 = __ballot(card && owner == 3u);
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  const uint64_t hv = (c.lane == 7) ? h0 : ((c.lane == 8) ? h1 : ((c.lane == 9) ? h2 : h3));
  if (c.lane < 11) img64[c.lane] = (c.lane < 7) ? 0ull : (hv << 4);
  if (c.lane < BTAIL / 16) *reinterpret_cast<uint4 *>(brow + 16 * c.lane) = make_uint4(0u, 0u, 0u, 0u);
  const uint64_t hs = (dl.seat == 0u) ? h0 : ((dl.seat == 1u) ? h1 : ((dl.seat == 2u) ? h2 : h3));
  const uint32_t nib = (dl.seat < 4u) ? ((uint32_t)(hs >> dl.sh) & 0xFu) : 0u;
  *reinterpret_cast<uint32_t *>(brow + dl.off) = __umul24(nib, 0x204081u) & 0x01010101u;
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
// (hardware wave w runs on SIMD w % 4: the four sequential chains — logic, prep, apply, scorer — get one SIMD
// each, ahead of the emit waves they share it with: s_setprio below)
constexpr int FW_LOGIC = 0, FW_PREP = 1, FW_APPLY = 2, FW_SCORER = 3, FW_LOADER = 4, FW_EMIT0 = 5;

template <int TPB, int NW>
__global__ __launch_bounds__(NW * 64) void k_rollout_flow(RolloutArgs A) {
  static_assert(TPB <= 32 && NW >= 6 && NW <= 16, "logic + loader + scorer + apply + prep + >=1 emit wave");
  static_assert(TPB % 4 == 0, "emit waves write 4 consecutive tables per instruction");
  constexpr int NE = NW - FW_EMIT0;
  constexpr int B = WS_BATCH;
  __shared__ __attribute__((aligned(16))) uint8_t img[TPB * TABLE_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t bimg[2 * TPB * BROW];  // byte images: bimg[s & 1] holds state s
  __shared__ __attribute__((aligned(16))) uint32_t cmd[FL_CR][TPB][CMD_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t ring[TPB][WS_RING][FR_WORDS];
  __shared__ __attribute__((aligned(8))) uint2 spost[FL_CR][TPB];  // fast mode: (d, static word) of state s
  __shared__ uint32_t udraw[FL_DR][TPB];
  __shared__ int f_state;    // states posted by the logic wave (fast mode)
  __shared__ int f_posted;   // command slots posted (prep wave; the logic wave in legacy mode)
  __shared__ int f_applied;  // states the apply wave has put into the byte images
  __shared__ int f_built[16]; // per emit wave: its groups' byte images are built
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
      for (int k = 0; k < WS_RING; k++) ring[tl][k][FR_TAG] = 0u;
    }
  }
  if (tid == 0) {
    f_state = 0;
    f_posted = 0;
    f_applied = 0;
    f_mode = 0;
    f_abort = 0;
    f_draws = min(total + 1, 4 - (int)(A.draw_base & 3u));
  }
  if (tid < 16) { f_done[tid] = 0; f_built[tid] = 0; }
  __syncthreads();  // images and the first draws are in LDS; the ring follows (entry tags)
  FL_STAMP(1);

  auto aborted = [&]() { return fl_ld(&f_abort) != 0; };

  if (wave == FW_LOADER) {
    // ------------------------------------------------------------------ loader wave
    uint32_t nb = 0, nb0 = 0, pbase = 0, pcount = 0, pidx[3] = {0, 0, 0}, pscb[3] = {0, 0, 0};
    brl_u32x4 pha[3], phb[3], pv[3];  // (native vectors: HIP's uint4 struct arrays are not promoted to registers here)
    if (valid) {
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
            brl_u32x4 *dv = reinterpret_cast<brl_u32x4 *>(dst);
            dv[0] = pha[k];
            dv[1] = phb[k];
            dv[2] = pv[k];
            e[FR_IDX] = pidx[k];
            fl_order();
            __hip_atomic_store(&e[FR_TAG], pscb[k] | ((FL_TAG_VALID | (b & 0x7FFFFu)) << 12), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        pcount = 0;
        progress = true;
      }
      // 3. boards whose slots every reader (scorer + apply waves) is done with
      {
        const int R = fl_min_done(f_done, FW_APPLY, 2, c.lane);
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
                pha[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k]];
                phb[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.hands)[2 * (size_t)pidx[k] + 1];
                pv[k] = reinterpret_cast<const brl_u32x4 *>(A.lut.values)[pidx[k]];
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
        __builtin_amdgcn_s_sleep(FL_NAP);
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
      const uint64_t v = __hip_atomic_load(reinterpret_cast<const uint64_t *>(&ring[tls][rslot][FR_IDX]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
      return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    uint2 nxt = ring_peek();
    auto ring_take = [&](bool deal) {  // make sure `nxt` is board bctr + 1 for every dealing lane
      int spins = 0;
      while (__any(deal && (nxt.y >> 12) != (FL_TAG_VALID | ((bctr + 1u) & 0x7FFFFu)))) {
        if (spins) __builtin_amdgcn_s_sleep(FL_NAP);
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
          __builtin_amdgcn_s_sleep(FL_NAP);
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
            __builtin_amdgcn_s_sleep(FL_NAP);
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
            space_upto = fl_min_done(f_done, FW_APPLY, NW - FW_APPLY, c.lane) + FL_CR;
            if (s < space_upto) break;
            __builtin_amdgcn_s_sleep(FL_NAP);
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
    __builtin_amdgcn_s_setprio(2);
    bool dead = false;
    int mode = 0;
    {
      int spins = 0;
      while ((mode = fl_ld(&f_mode)) == 0) {
        __builtin_amdgcn_s_sleep(FL_NAP);
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
            __builtin_amdgcn_s_sleep(FL_NAP);
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
            space_upto = fl_min_done(f_done, FW_APPLY, NW - FW_APPLY, c.lane) + FL_CR;
            if (s < space_upto) break;
            __builtin_amdgcn_s_sleep(FL_NAP);
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
        fl_wake();
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
    __builtin_amdgcn_s_setprio(1);
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
          __builtin_amdgcn_s_sleep(FL_NAP);
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
          const uint4 vv = *reinterpret_cast<const uint4 *>(&ring[tls][(w.x >> 16) & 15u][FR_VALUES]);
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
  } else if (wave == FW_APPLY) {
    // ------------------------------------------------------------------ apply wave
    __builtin_amdgcn_s_setprio(2);
    // Keeps the images up to date for everybody: the packed images (img: what goes back to HBM) and TWO byte
    // images per table (bimg[s & 1] holds state s), so that the emit waves only copy.  Lane = table for the
    // calls (one byte per call and copy); a re-deal is wave-cooperative.  bimg[s & 1] was last written for
    // state s-2, so it needs the calls of sub-steps s-2 (carried over from the previous command) and s-1.
    bool dead = false;
    uint32_t w0p = 0;  // command word 0 of slot s-1
    const int dg = c.lane >> 4, dj = c.lane & 15;  // deal pass: 16 lanes per table, lane dj <-> rank dj (4 cards)
#ifdef BRL_TIMING
    unsigned long long dp_t[4] = {0, 0, 0, 0};
    int dp_n = 0;
#endif
    auto deal_pass = [&](uint64_t mask, uint32_t wsrc, uint8_t *bdst, bool packed) {
      while (mask) {
#ifdef BRL_TIMING
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        dp_n++;
#endif
        int q[4];
        uint32_t wq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          q[k] = mask ? (int)__ffsll((unsigned long long)mask) - 1 : -1;
          mask &= mask - (mask ? 1ull : 0ull);
          wq[k] = __builtin_amdgcn_readlane(wsrc, q[k] < 0 ? 0 : q[k]);
        }
        const int tq = (dg == 0) ? q[0] : ((dg == 1) ? q[1] : ((dg == 2) ? q[2] : q[3]));
        const uint32_t wt = (dg == 0) ? wq[0] : ((dg == 1) ? wq[1] : ((dg == 2) ? wq[2] : wq[3]));
#ifdef BRL_TIMING
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        unsigned long long tc = tb;
#endif
        if (tq >= 0) {
          const uint32_t *e = &ring[tq][(wt >> 16) & 15u][FR_HANDS];
          const uint4 ha = *reinterpret_cast<const uint4 *>(e), hb = *reinterpret_cast<const uint4 *>(e + 4);
#ifdef BRL_TIMING
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          tc = __builtin_amdgcn_s_memtime();
#endif
          const uint64_t h0 = (uint64_t)ha.x | ((uint64_t)ha.y << 32), h1 = (uint64_t)ha.z | ((uint64_t)ha.w << 32);
          const uint64_t h2 = (uint64_t)hb.x | ((uint64_t)hb.y << 32), h3 = (uint64_t)hb.z | ((uint64_t)hb.w << 32);
          uint8_t *brow = bdst + tq * BROW;
          // no history: 26 x 16 B by 16 lanes in two writes (the second one clamped)
          *reinterpret_cast<uint4 *>(brow + 16 * dj) = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4 *>(brow + 16 * min(16 + dj, BTAIL / 16 - 1)) = make_uint4(0u, 0u, 0u, 0u);
          // the four observer tails: lane dj < 13 expands rank dj's 4 cards of every hand, lanes 13..15 clear
          // the 3 history dwords
          const uint32_t sh = 4u * (uint32_t)dj + 4u;
          const uint32_t off = (dj < 13) ? 12u + 4u * (uint32_t)dj : 4u * (uint32_t)(dj - 13);
          const uint32_t keep = (dj < 13) ? 0x01010101u : 0u;
          uint32_t *t0 = reinterpret_cast<uint32_t *>(brow + BTAIL + off);
          t0[0] = __umul24((uint32_t)(h0 >> sh) & 0xFu, 0x204081u) & keep;
          t0[16] = __umul24((uint32_t)(h1 >> sh) & 0xFu, 0x204081u) & keep;
          t0[32] = __umul24((uint32_t)(h2 >> sh) & 0xFu, 0x204081u) & keep;
          t0[48] = __umul24((uint32_t)(h3 >> sh) & 0xFu, 0x204081u) & keep;
          if (packed && dj < 11) {
            const uint64_t hv = (dj == 7) ? h0 : ((dj == 8) ? h1 : ((dj == 9) ? h2 : h3));
            reinterpret_cast<uint64_t *>(img + tq * TABLE_BYTES)[dj] = (dj < 7) ? 0ull : hv;
          }
        }
#ifdef BRL_TIMING
        {
          const unsigned long long td = __builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long te = __builtin_amdgcn_s_memtime();
          dp_t[0] += tb - ta; dp_t[1] += tc - tb; dp_t[2] += td - tc; dp_t[3] += te - td;
        }
#endif
      }
    };
    int posted = 0, emit_read = 0;
    {  // the emit waves build both byte images from the packed ones first
      int spins = 0;
      while (fl_min_done(f_built, FW_EMIT0, NE, c.lane) < 1) {
        __builtin_amdgcn_s_sleep(FL_NAP);
        if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
      }
    }
#ifdef BRL_TIMING
    unsigned long long ap_t[4] = {0, 0, 0, 0};
    int ap_ndeal = 0;
#endif
    for (int s = 0; s <= total && !dead; s++) {
#ifdef BRL_TIMING
      const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
      if (s >= posted) {
        int spins = 0;
        while ((posted = fl_ld(&f_posted)) <= s) {
          __builtin_amdgcn_s_sleep(FL_NAP);
          if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
        }
        if (aborted()) dead = true;
        if (dead) break;
        fl_order();
      }
      if (s - 1 > emit_read) {  // every emit wave must have read state s-2 out of bimg[s & 1]
        int spins = 0;
        while ((emit_read = fl_min_done(f_done, FW_EMIT0, NE, c.lane)) < s - 1) {
          __builtin_amdgcn_s_sleep(FL_NAP);
          if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
        }
        if (dead) break;
      }
#ifdef BRL_TIMING
      const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
#endif
      const uint32_t w0 = cmd[s & (FL_CR - 1)][tls][0];
      uint8_t *bcur = bimg + (s & 1) * (TPB * BROW);
      const uint8_t *bold = bimg + ((s & 1) ^ 1) * (TPB * BROW);
      const bool deal_now = valid && (w0 & 0x200u), deal_prev = valid && (w0p & 0x200u);
      // re-deals, up to 4 tables per pass (16 lanes each): boards dealt at this slot (packed + byte image) and,
      // carried over, those dealt one slot ago (this copy of the byte image still holds the old board)
      deal_pass((A.debug & 32) ? 0ull : __ballot(deal_now), w0, bcur, true);
#ifdef BRL_TIMING
      const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
      ap_ndeal += __popcll(__ballot(deal_now));
#endif
      deal_pass((A.debug & 64) ? 0ull : __ballot(deal_prev), w0p, bcur, false);
#ifdef BRL_TIMING
      const unsigned long long tp3 = __builtin_amdgcn_s_memtime();
#endif
      wave_lds_order();
      if (tl < TPB) {
        uint8_t *brow = bcur + tl * BROW;
        const uint32_t hp = (deal_prev || deal_now) ? 0u : (w0p & 0x1FFu);  // call of sub-step s-2
        const uint32_t hn = deal_now ? 0u : (w0 & 0x1FFu);                 // call of sub-step s-1
        if (hn) {
          const int hb = (int)hn - 1;
          atomicOr(reinterpret_cast<uint32_t *>(img + tl * TABLE_BYTES) + (hb >> 5), 1u << (hb & 31));
        }
#pragma unroll
        for (int which = 0; which < 2; which++) {
          const uint32_t h = which ? hn : hp;
          if (h) {
            const int hb = (int)h - 1;
            if (hb < BTAIL) {
              brow[hb] = 1;
            } else {  // the last bid's 12 bytes live in the four observer tails
#pragma unroll
              for (int q = 0; q < 4; q++) brow[hb + 64 * q] = 1;
            }
          }
        }
      }
#ifdef BRL_TIMING
      {
        const unsigned long long tp4 = __builtin_amdgcn_s_memtime();
        ap_t[0] += tp1 - tp0; ap_t[1] += tp2 - tp1; ap_t[2] += tp3 - tp2; ap_t[3] += tp4 - tp3;
      }
#endif
      w0p = w0;
      fl_order();
      if (c.lane == 0) {
        fl_st(&f_applied, s + 1);
        fl_st(&f_done[FW_APPLY], s);  // the ring entry of a board dealt at slot s is read once more, at slot s+1
      }
      fl_wake();
      FL_STAMP(2 + s);
    }
#ifdef BRL_TIMING
    if (c.lane == 0 && A.terminated_count) {
      unsigned long long *d = A.terminated_count + ((size_t)blockIdx.x * NW + wave) * 48;
      d[38] = ap_t[0]; d[39] = ap_t[1]; d[40] = ap_t[2]; d[41] = ap_t[3]; d[42] = (unsigned long long)ap_ndeal;
      d[43] = dp_t[0]; d[44] = dp_t[1]; d[45] = dp_t[2]; d[37] = dp_t[3]; d[36] = (unsigned long long)dp_n;
    }
#endif
    if (c.lane == 0) {
      fl_st(&f_done[FW_APPLY], total + 1);
      if (dead) {
        fl_st(&f_abort, 1);
        fl_st(&f_applied, total + 1);
      }
    }
  } else {
    // ------------------------------------------------------------------ emit waves
    // Copy-only: bimg[s & 1] (kept by the apply wave) -> 4 observation rows per store instruction, plus the 4
    // legal-mask rows from the command.  One byte per observation bit with the seats in absolute order, so a
    // lane's 32 output bytes are two 16-byte LDS reads and one rotate per dword (the observer's seat).
    const GroupLane gl = make_group_lane();
    const MaskLane ml = make_mask_lane();
    const ByteLane bl = make_byte_lane();
    constexpr int NG = TPB / 4;
    constexpr int GPW = (NG + NE - 1) / NE;
    const int rr = (gl.r < 4) ? gl.r : 3;
    int sub = 0;
    int64_t row0 = table0;
    int left[GPW];
#pragma unroll
    for (int k = 0; k < GPW; k++) {
      const int g = (wave - FW_EMIT0) + k * NE;
      int64_t rem = (g < NG && !(A.debug & 1)) ? A.n - (table0 + 4 * g) : 0;
      left[k] = (int)max((int64_t)0, min((int64_t)4, rem));
      if (g < NG) {
        bimg_build(img + 4 * g * TABLE_BYTES, bimg + 4 * g * BROW, gl, bl);
        bimg_build(img + 4 * g * TABLE_BYTES, bimg + TPB * BROW + 4 * g * BROW, gl, bl);
      }
    }
    fl_order();  // the apply wave waits for every emit wave's f_built before it touches the byte images
    if (c.lane == 0) fl_st(&f_built[wave], 1);
    if (A.debug & 128) {  // experiment: no emit waves at all
      if (c.lane == 0) fl_st(&f_done[wave], total + 1);
      goto flow_done;
    }
    int avail = 0;  // states known to be in the byte images
    bool dead = false;
    auto wait_slot = [&](int s) {
      if (s < avail) return;
      int spins = 0;
      while ((avail = fl_ld(&f_applied)) <= s) {
        __builtin_amdgcn_s_sleep(FL_NAP);
        if (++spins > FL_SPIN_MAX || aborted()) { dead = true; break; }
      }
      if (aborted()) dead = true;
      fl_order();
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
      for (; s_next < total; s_next++) {
        wait_slot(s_next);
        if (dead) break;
        const uint32_t(*cs)[CMD_WORDS] = cmd[s_next & (FL_CR - 1)];
        const uint8_t *bsrc = bimg + (s_next & 1) * (TPB * BROW);
        uint32_t w0[GPW];
        uint4 q0[GPW], q1[GPW];
        uint64_t la[GPW], lb[GPW];
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          if (left[k] == 0) continue;
          const int g = (wave - FW_EMIT0) + k * NE;
          w0[k] = cs[4 * g + rr][0];
          la[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qa][2]);
          lb[k] = *reinterpret_cast<const uint64_t *>(&cs[4 * g + ml.qb][2]);
          byte_chunk_load(bsrc + 4 * g * BROW, (w0[k] >> 10) & 3u, bl, q0[k], q1[k]);
        }
        fl_order();
        if (c.lane == 0) fl_st(&f_done[wave], s_next + 1);  // behind this wave's LDS reads of the slot
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          if (left[k] == 0) continue;
          if (olane) byte_chunk_store(q0[k], q1[k], (w0[k] >> 10) & 3u, (w0[k] >> 12) & 15u, optr[k], bl);
          if (ml.active) *mptr[k] = mask_dword(la[k], lb[k], ml);
          optr[k] += ostep;
          mptr[k] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(mptr[k]) + mstep);
        }
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
      const uint8_t *bsrc = bimg + (s & 1) * (TPB * BROW);
      if (emit) {
#pragma unroll
        for (int k = 0; k < GPW; k++) {
          const int g = (wave - FW_EMIT0) + k * NE;
          if (left[k] <= 0) continue;
          const uint32_t w0 = cs[4 * g + rr][0];
          uint4 q0, q1;
          byte_chunk_load(bsrc + 4 * g * BROW, (w0 >> 10) & 3u, bl, q0, q1);
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
      }
      fl_order();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (general path) the LDS reads above have returned
      if (c.lane == 0) fl_st(&f_done[wave], s + 1);
      FL_STAMP(2 + s);
      if (++sub == A.substeps) {
        sub = 0;
        row0 += A.n;
      }
    }
    if (dead && c.lane == 0) fl_st(&f_done[wave], total + 1);
  }
flow_done:
  FL_STAMP(46);
  __syncthreads();
  for (int i = tid; i < TPB * 16; i += NW * 64) {
    int64_t tb = table0 + i / 16;
    if (tb < A.n) A.state[table0 * 16 + i] = img64[i];
  }
  FL_STAMP(47);
}
