// rollout_common.hpp — helpers shared by the pipelined rollout kernel (rollout_pipe.hpp): the chain-friendly
// auction state of the logic wave (fast_step) and the byte images of the emit waves.
// (The flag-synchronised kernel these were first written for is a measured dead end, kept un-compiled in
// scripts/micro/exp_flow/.)
#pragma once

// ---- chain-friendly auction state of the logic wave (fast mode) -----------------------------------
// d: [8:0] dealer + turn (seat = low 2 bits) | [14:9] rem = 35 - lb1 | [16:15] last bidder seat |
//    [19:17] e = doubling state: dblst | own << 2 (dblst 0 none / 1 X / 2 XX or "no bid yet"; own = the
//    player to act is on the last bidder's side); X / XX is legal iff e is 0 or 5 |
//    [22:20] pass count, +1 once a bid exists  => the auction is over iff bit 22 is set.
constexpr uint32_t FD_REM = 9, FD_LBSEAT = 15, FD_E = 17, FD_PASS = 20, FD_TERM = 22;

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

// re-deal lanes: lane L < 52 writes the 4 hand bytes [4j, 4j+4) of seat L / 13's tail, j = L % 13
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
// a freshly dealt board in one table's byte image from the board's four packed hand words (LDS ring entry)
__device__ __forceinline__ void deal_bytes_hands(uint8_t *brow, const uint32_t *hands, const LaneConst &c,
                                                 const DealLane &dl) {
  if (c.lane < BTAIL / 16) *reinterpret_cast<uint4 *>(brow + 16 * c.lane) = make_uint4(0u, 0u, 0u, 0u);
  else if (c.lane < BTAIL / 16 + 4) *reinterpret_cast<uint4 *>(brow + BTAIL + 64 * (c.lane - BTAIL / 16)) = make_uint4(0u, 0u, 0u, 0u);
  wave_lds_order();  // (the 16 zero bytes at the head of each tail cover its first hand dword, rewritten below)
  if (c.lane < 52) {
    const uint64_t hs = *reinterpret_cast<const uint64_t *>(hands + 2 * dl.seat);
    *reinterpret_cast<uint32_t *>(brow + dl.off) = __umul24((uint32_t)(hs >> (dl.sh + 4u)) & 0xFu, 0x204081u) & 0x01010101u;
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
