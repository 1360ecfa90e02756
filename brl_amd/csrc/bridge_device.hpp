// bridge_device.hpp — device-side building blocks of the bridge-bidding hot path (gfx950).
//
// Execution model (DESIGN.md "Kernels"): a 64-lane wavefront owns K consecutive tables
// (K = 1, 2, 4 or 8).  Two kinds of work alternate:
//   * table LOGIC — integer/branch work on one table's bit-packed scalars.  Lane l runs
//     the logic of table (l % K), so the K tables advance in parallel in one instruction
//     stream and every lane holds the scalars of "its" table in registers.
//   * row EMISSION — the whole wave writes one table's 480-byte observation row with a
//     single coalesced 8-B-per-lane store (60 lanes x 8 B), reading the table's packed
//     auction history / hand image from LDS.
// The 128-byte packed table is mirrored 1:1 in LDS while a wave works on it.
//
// Semantics restate pgx==1.4.0's bridge_bidding as evidenced by the reference call sites;
// every function cites them (paths under /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brl {

// ---- packed table layout (16 x uint64 = 128 B) -------------------------------------
// words 0..6   auction history, ABSOLUTE seats, one nibble per (event): bit index equals
//              the observation bit index (wb5/utils.py:28-46) with relative seat replaced
//              by absolute seat: 4+s opening pass by seat s; 8+12b+s bid b by s;
//              8+12b+4+s doubled by s; 8+12b+8+s redoubled by s.  (bits 0..3 unused)
// words 7..10  hand_obs[seat] << 4 : the 52 own-card bits of each seat in observation
//              order rank*4+suit (wb5/utils.py:18-26), pre-shifted to byte 53's high nibble
// word 11      scalars (SC_* below)      word 12  first denominations | tricks[16..19]
// word 13      tricks nibbles 0..15       word 14  lut_idx | board_ctr<<32
// word 15      rewards, 4 x int16 by player id
constexpr int W_HIST = 0, W_HAND = 7, W_SC = 11, W_FD = 12, W_TR = 13, W_CTR = 14, W_REW = 15;
constexpr int TABLE_BYTES = 128;

// scalars, low 32 bits
constexpr int SC_DEALER = 0;   // 2  seat of the dealer
constexpr int SC_VULNS = 2;    // 1
constexpr int SC_VULEW = 3;    // 1
constexpr int SC_SHUF = 4;     // 8  player id at seat s = (>> (4+2s)) & 3   (_shuffled_players)
constexpr int SC_LB1 = 12;     // 6  _last_bid + 1 (0 = no bid yet)
constexpr int SC_LBSEAT = 18;  // 2  seat of _last_bidder (valid when LB1 != 0)
constexpr int SC_X = 20;       // 1  _call_x
constexpr int SC_XX = 21;      // 1  _call_xx
constexpr int SC_PASS = 22;    // 3  _pass_num
constexpr int SC_TERM = 25;    // 1  terminated
constexpr int SC_MASKALL = 26; // 1  legal_action_mask forced all-True (pgx Env.step at terminal)
constexpr int SC_ILLEGAL = 27; // 1  an illegal action was taken
// scalars, high 32 bits
constexpr int SCH_TURN = 0;    // 9  _turn
constexpr int SCH_STEP = 9;    // 10 _step_count

constexpr uint64_t ALL_ACTIONS = (1ull << 38) - 1;

struct Tbl {
  uint32_t sc, sch;   // scalars
  uint32_t fd, t2;    // first denominations (3 bits each: 0 none, seat+1; NS at 3d, EW at 15+3d) ; tricks 16..19
  uint32_t t0, t1;    // tricks nibbles 0..7 / 8..15 ; nibble index = seat*5 + (4 - strain)
  uint32_t lut, bctr; // LUT row (0xFFFFFFFF = explicit deal), boards dealt by this slot
  uint32_t r01, r23;  // rewards int16 x4 by player id
};

__device__ __forceinline__ uint32_t bits(uint32_t x, int s, int w) { return (x >> s) & ((1u << w) - 1u); }

__device__ __forceinline__ int cur_seat(const Tbl &t) { return (int)((bits(t.sc, SC_DEALER, 2) + bits(t.sch, SCH_TURN, 9)) & 3u); }
__device__ __forceinline__ int player_at(const Tbl &t, int seat) { return (int)((t.sc >> (SC_SHUF + 2 * seat)) & 3u); }
__device__ __forceinline__ int cur_player(const Tbl &t) { return player_at(t, cur_seat(t)); }
__device__ __forceinline__ int seat_of_player(const Tbl &t, int p) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < 4; k++) s = (player_at(t, k) == p) ? k : s;
  return s;
}

__device__ __forceinline__ int reward_of(const Tbl &t, int p) {
  // one 64-bit shift instead of a select between two fields: a select of two plain field
  // loads is folded into a dynamically indexed load, which drags the whole Tbl out of registers
  uint64_t w = (uint64_t)t.r01 | ((uint64_t)t.r23 << 32);
  return (int)(int16_t)(w >> (p * 16));
}
__device__ __forceinline__ void set_rewards(Tbl &t, int a0, int a1, int a2, int a3) {
  t.r01 = ((uint32_t)a0 & 0xFFFFu) | ((uint32_t)a1 << 16);
  t.r23 = ((uint32_t)a2 & 0xFFFFu) | ((uint32_t)a3 << 16);
}

// vulnerability nibble of the observation for an observer at `seat` (wb5/utils.py:15-16)
__device__ __forceinline__ uint32_t vul_nibble(const Tbl &t, int seat) {
  uint32_t ns = bits(t.sc, SC_VULNS, 1), ew = bits(t.sc, SC_VULEW, 1);
  uint32_t we = (seat & 1) ? ew : ns, they = (seat & 1) ? ns : ew;
  return (we ? 2u : 1u) | (they ? 8u : 4u);
}

__device__ __forceinline__ uint32_t vul_nibble_sc(uint32_t sc, int seat) {
  uint32_t ns = bits(sc, SC_VULNS, 1), ew = bits(sc, SC_VULEW, 1);
  uint32_t we = (seat & 1) ? ew : ns, they = (seat & 1) ? ns : ew;
  return (we ? 2u : 1u) | (they ? 8u : 4u);
}

// A2: legal_action_mask of the player to act, derived from the scalars:
// Pass always; bids strictly above the last bid; X iff the last bid is the opponents' and
// undoubled; XX iff own side's bid is doubled and not redoubled; all-True at a terminal.
__device__ __forceinline__ uint64_t legal_mask(const Tbl &t) {
  uint32_t lb1 = bits(t.sc, SC_LB1, 6);
  uint64_t m = (ALL_ACTIONS >> (3 + lb1)) << (3 + lb1);
  m |= 1ull;
  uint32_t own = ((bits(t.sc, SC_LBSEAT, 2) ^ (uint32_t)cur_seat(t)) & 1u) ^ 1u;
  uint32_t x = bits(t.sc, SC_X, 1), xx = bits(t.sc, SC_XX, 1);
  uint32_t has = lb1 != 0;
  uint32_t can_x = has & (own ^ 1u) & (x ^ 1u) & (xx ^ 1u);
  uint32_t can_xx = has & own & x & (xx ^ 1u);
  m |= (uint64_t)((can_x << 1) | (can_xx << 2));
  return bits(t.sc, SC_MASKALL, 1) ? ALL_ACTIONS : m;
}

// A4: duplicate score of the declaring side (13 down X non-vul = 3500,
// src/duplicate.py:36; 13 down XX vul = 7600 = reward_scale, ppo.py:174).
__device__ __forceinline__ int contract_score(int den, int level, int vul, int x, int xx, int trick) {
  int u = level + 6 - trick;  // > 0: down by u
  int down_und = (vul ? 100 : 50) * u;
  int pen_v = 300 * u - 100;
  int pen_nv = (u <= 3) ? (200 * u - 100) : (300 * u - 400);
  int pen = vul ? pen_v : pen_nv;
  pen = xx ? 2 * pen : pen;
  int down = (x | xx) ? pen : down_und;
  int per = (den <= 1) ? 20 : 30;
  int points = (per * level + (den == 4 ? 10 : 0)) * (xx ? 4 : (x ? 2 : 1));
  int sc = points + ((points >= 100) ? (vul ? 500 : 300) : 50);
  sc += (level == 6) ? (vul ? 750 : 500) : 0;
  sc += (level == 7) ? (vul ? 1500 : 1000) : 0;
  sc += xx ? 100 : (x ? 50 : 0);
  int ov = xx ? (vul ? 400 : 200) : (x ? (vul ? 200 : 100) : per);
  sc -= u * ov;  // overtricks = -u
  return (u > 0) ? -down : sc;
}

__device__ __forceinline__ int trick_nibble(const Tbl &t, int seat, int den) {
  int i = seat * 5 + (4 - den);
  uint64_t lo = (uint64_t)t.t0 | ((uint64_t)t.t1 << 32);  // (see reward_of for why not a select)
  uint32_t a = (uint32_t)(lo >> ((i & 15) * 4));
  uint32_t b = t.t2 >> ((i & 3) * 4);
  return (int)(((i < 16) ? a : b) & 15u);
}

// terminal reward by player id (workspace/test_bridge_with_openspiel.py:118-123);
// declarer = first of the declaring side to name the strain
__device__ __forceinline__ void terminal_reward(Tbl &t) {
  uint32_t lb1 = bits(t.sc, SC_LB1, 6);
  int b = (int)lb1 - 1;
  int level = (b * 13) >> 6;  // b / 5 for 0 <= b <= 34
  int den = b - level * 5;
  level += 1;
  int side = (int)bits(t.sc, SC_LBSEAT, 2) & 1;
  int decl = (int)bits(t.fd, side * 15 + den * 3, 3) - 1;
  int vul = (int)(side ? bits(t.sc, SC_VULEW, 1) : bits(t.sc, SC_VULNS, 1));
  int trick = trick_nibble(t, decl & 3, den);
  int s = contract_score(den, level, vul, (int)bits(t.sc, SC_X, 1), (int)bits(t.sc, SC_XX, 1), trick);
  s = (lb1 == 0) ? 0 : s;  // pass-out: all zero (src/evaluation.py:465-467)
  int r[4];
#pragma unroll
  for (int p = 0; p < 4; p++) r[p] = 0;
#pragma unroll
  for (int seat = 0; seat < 4; seat++) {
    int p = player_at(t, seat);
    int v = ((seat & 1) == side) ? s : -s;
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = (p == q) ? v : r[q];
  }
  set_rewards(t, r[0], r[1], r[2], r[3]);
}

// A2, the auction itself: one LEGAL-or-not call by `seat` on a live table.  Touches only the
// scalar words (last bid/bidder, X/XX, pass count, turn, step count, terminated) and returns
// the history bit to OR into the table's image (-1: none).  Termination: four passes with no
// bid (pass-out) or three passes after the last bid.
__device__ __forceinline__ int auction_step(Tbl &t, int action, int seat) {
  uint32_t sc = t.sc;
  uint32_t lb1 = bits(sc, SC_LB1, 6);
  uint32_t pass = bits(sc, SC_PASS, 3);
  const bool is_pass = action == 0, is_x = action == 1, is_xx = action == 2, is_bid = action >= 3;
  const int b = action - 3;
  const int hb_bid = 8 + 12 * b + seat;
  const int hb_dbl = 8 + 12 * ((int)lb1 - 1) + (is_x ? 4 : 8) + seat;
  const int hb_pass = 4 + seat;
  const int hb = is_bid ? hb_bid : (is_pass ? ((lb1 == 0) ? hb_pass : -1) : ((lb1 != 0) ? hb_dbl : -1));
  pass = is_pass ? pass + 1u : 0u;
  sc |= (is_x ? (1u << SC_X) : 0u) | (is_xx ? (1u << SC_XX) : 0u);
  const uint32_t bid_clear = (63u << SC_LB1) | (3u << SC_LBSEAT) | (1u << SC_X) | (1u << SC_XX);
  const uint32_t bid_set = ((uint32_t)(b + 1) << SC_LB1) | ((uint32_t)seat << SC_LBSEAT);
  sc = is_bid ? ((sc & ~bid_clear) | bid_set) : sc;
  lb1 = is_bid ? (uint32_t)(b + 1) : lb1;
  sc = (sc & ~(7u << SC_PASS)) | (pass << SC_PASS);
  const bool term = pass == ((lb1 != 0) ? 3u : 4u);
  sc |= term ? ((1u << SC_TERM) | (1u << SC_MASKALL)) : 0u;  // all-True mask at a terminal (pgx Env.step)
  t.sc = sc;
  t.sch += (1u << SCH_STEP) + (term ? 0u : (1u << SCH_TURN));  // _step_count+1 ; next seat unless over
  return hb;
}

// first player of each side to name each strain (decides the declarer)
__device__ __forceinline__ void note_first_denomination(uint32_t &fd, int seat, int action) {
  int b = action - 3;
  int level0 = (b * 13) >> 6;
  int den = b - level0 * 5;
  int slot = (seat & 1) * 15 + den * 3;
  bool set = (action >= 3) && (bits(fd, slot & 31, 3) == 0);
  fd |= set ? ((uint32_t)(seat + 1) << (slot & 31)) : 0u;
}

// A2: env.step on one table (pgx core.Env.step + bridge _step).
// Returns the history bit to OR into the LDS image (-1: none).
__device__ __forceinline__ int table_step(Tbl &t, int action) {
  if (bits(t.sc, SC_TERM, 1)) {  // finished table stepped again: zero rewards, no-op (G9)
    t.r01 = 0;
    t.r23 = 0;
    return -1;
  }
  uint32_t illegal = (uint32_t)((legal_mask(t) >> action) & 1ull) ^ 1u;
  int seat = cur_seat(t);
  note_first_denomination(t.fd, seat, action);
  int hist_bit = auction_step(t, action, seat);
  if (bits(t.sc, SC_TERM, 1)) {
    terminal_reward(t);
  } else {
    t.r01 = 0;
    t.r23 = 0;
  }
  if (illegal) {  // [RECALL] pgx: offender -1, every other player +1*(4-1); game over
    int p = player_at(t, seat);
    set_rewards(t, p == 0 ? -1 : 3, p == 1 ? -1 : 3, p == 2 ? -1 : 3, p == 3 ? -1 : 3);
    t.sc |= (1u << SC_TERM) | (1u << SC_ILLEGAL) | (1u << SC_MASKALL);
  }
  return hist_bit;
}

// A5 pre-step half of auto_reset (src/utils.py:34-43)
__device__ __forceinline__ void auto_reset_clear(Tbl &t) {
  if (bits(t.sc, SC_TERM, 1)) {
    t.sc &= ~(1u << SC_TERM);
    t.sch &= ~(1023u << SCH_STEP);
    t.r01 = 0;
    t.r23 = 0;
  }
}

// ---- counter-based RNG: Philox4x32-10 ------------------------------------------------
constexpr uint32_t STREAM_RESET = 0x42524C52u;   // 'BRLR'
constexpr uint32_t STREAM_ACTION = 0x42524C41u;  // 'BRLA'

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
  uint32_t k0, k1;  // seed
};

// A1: parameters of board number `board_ctr` of env `env_id` (uniform LUT row, dealer,
// vulnerabilities, one of the 8 team-preserving seatings — SURVEY §8a A1 / App. B), and the
// fresh scalars.  terminated / illegal / rewards are carried by the caller (src/utils.py:50-54).
// (LUT row, fresh scalar bits) of board number `board_ctr` of env `env_id`
__device__ __forceinline__ void board_params(const Rng &g, uint64_t env_id, uint32_t board_ctr, uint32_t lut_len,
                                             uint32_t &idx, uint32_t &sc_bits) {
  uint32_t r[4];
  philox4x32_10((uint32_t)env_id, board_ctr, STREAM_RESET, (uint32_t)(env_id >> 32), g.k0, g.k1, r);
  idx = __umulhi(r[0], lut_len);
  uint32_t dealer = r[1] & 3u, vns = (r[1] >> 2) & 1u, vew = (r[1] >> 3) & 1u, arr = (r[1] >> 4) & 7u;
  uint32_t a0 = arr & 1u, b0 = 2u + ((arr >> 1) & 1u), a1 = 1u - a0, b1 = 5u - b0;
  uint32_t shuf_a = a0 | (b0 << 2) | (a1 << 4) | (b1 << 6);  // NS = team {0,1}
  uint32_t shuf_b = b0 | (a0 << 2) | (b1 << 4) | (a1 << 6);  // NS = team {2,3}
  uint32_t shuf = (arr & 4u) ? shuf_a : shuf_b;
  sc_bits = dealer | (vns << SC_VULNS) | (vew << SC_VULEW) | (shuf << SC_SHUF);
}

__device__ __forceinline__ void apply_fresh(Tbl &t, uint32_t idx, uint32_t sc_bits, uint32_t board_ctr,
                                            uint32_t keep_bits) {
  t.sc = sc_bits | keep_bits;
  t.sch = 0;
  t.fd = 0;
  t.lut = idx;
  t.bctr = board_ctr;
}

__device__ __forceinline__ void fresh_scalars(Tbl &t, const Rng &g, uint64_t env_id, uint32_t board_ctr,
                                              uint32_t lut_len, uint32_t keep_bits) {
  uint32_t idx, sc_bits;
  board_params(g, env_id, board_ctr, lut_len, idx, sc_bits);
  apply_fresh(t, idx, sc_bits, board_ctr, keep_bits);
}

__device__ __forceinline__ void pack_tricks(Tbl &t, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
  v0 &= 0xFFFFFu; v1 &= 0xFFFFFu; v2 &= 0xFFFFFu; v3 &= 0xFFFFFu;
  t.t0 = v0 | (v1 << 20);
  t.t1 = (v1 >> 12) | (v2 << 8) | (v3 << 28);
  t.t2 = v3 >> 4;
}

// ---- per-lane constants ---------------------------------------------------------------
struct LaneConst {
  int lane;
  int hist_idx;    // LDS byte of the history this lane turns into 8 obs bytes (lanes >= 54 clamp)
  int hand_off;    // byte offset into the observer's hand word (lanes >= 53)
  uint32_t hist_keep;  // which bits of the rotated history byte survive
  uint32_t hand_keep;  // which bits of the hand byte survive
  int dsuit, dshift;   // hand decode: LUT key word and digit shift of obs bit `lane`
};

__device__ __forceinline__ LaneConst make_lane_const() {
  LaneConst c;
  c.lane = (int)(threadIdx.x & 63u);
  c.hist_idx = c.lane < 54 ? c.lane : 53;
  c.hand_off = c.lane > 53 ? c.lane - 53 : 0;
  c.hist_keep = c.lane < 53 ? 0xFFu : (c.lane == 53 ? 0x0Fu : 0u);
  c.hand_keep = c.lane < 53 ? 0u : 0xFFu;
  int os_rank = c.lane >> 2, os_suit = c.lane & 3;       // obs bit = rank*4 + suit (C,D,H,S x 2..A)
  c.dsuit = 3 - os_suit;                                    // pgx suit order S,H,D,C
  int rank = (os_rank + 1) % 13;                            // pgx rank order A,2,..,K
  c.dshift = 2 * (12 - rank);
  return c;
}

// A3: one table's 480-byte observation row from its LDS image (wb5/utils.py:15-52).
// Wave-cooperative: lane l produces obs bytes [8l, 8l+8) and the wave issues ONE 8-byte-per-
// lane store = 480 contiguous bytes.  `seat` (observer seat) and `vulnib` are wave-uniform.
__device__ __forceinline__ void emit_obs_row(const uint8_t *img, int seat, uint32_t vulnib, uint8_t *dst_row,
                                             const LaneConst &c) {
  uint32_t a = img[c.hist_idx];
  uint32_t h = img[W_HAND * 8 + seat * 8 + c.hand_off];
  uint32_t m1 = (0xFu >> seat) * 0x11u;
  // relative seat = (caller seat - observer seat) mod 4: rotate every nibble right by `seat`
  uint32_t rot = ((a >> seat) & m1) | ((a << (4 - seat)) & (m1 ^ 0xFFu));
  uint32_t byte = (rot & c.hist_keep) | (h & c.hand_keep);
  byte |= (c.lane == 0) ? vulnib : 0u;
  uint32_t lo = __umul24(byte & 0xFu, 0x204081u) & 0x01010101u;  // 4 bits -> 4 bytes of 0/1
  uint32_t hi = __umul24(byte >> 4, 0x204081u) & 0x01010101u;
  if (c.lane < 60) *reinterpret_cast<uint2 *>(dst_row + c.lane * 8) = make_uint2(lo, hi);
}

// The same row as the network's input: 480 values 0.0 / 1.0 in float (fmt 0), bf16 (1) or fp16 (2) — what
// `obs.astype(float32)` (src/roll_out.py:75) or a low-precision cast would produce from the bytes, written by the launch that
// produces the observation (no separate cast launch per forward).  Lane l: values [8l, 8l+8).
__device__ __forceinline__ void emit_obs_row_cast(const uint8_t *img, int seat, uint32_t vulnib, void *dst_row, int fmt,
                                                  const LaneConst &c) {
  uint32_t a = img[c.hist_idx];
  uint32_t h = img[W_HAND * 8 + seat * 8 + c.hand_off];
  uint32_t m1 = (0xFu >> seat) * 0x11u;
  uint32_t rot = ((a >> seat) & m1) | ((a << (4 - seat)) & (m1 ^ 0xFFu));
  uint32_t byte = (rot & c.hist_keep) | (h & c.hand_keep);
  byte |= (c.lane == 0) ? vulnib : 0u;
  if (c.lane >= 60) return;
  if (fmt == 0) {
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = ((byte >> i) & 1u) ? 0x3F800000u : 0u;
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<float *>(dst_row) + c.lane * 8);
    dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
    dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
  } else {
    const uint32_t one = (fmt == 1) ? 0x3F80u : 0x3C00u;
    uint32_t d[4];
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = (((byte >> (2 * i)) & 1u) ? one : 0u) | (((byte >> (2 * i + 1)) & 1u) ? (one << 16) : 0u);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(dst_row) + c.lane * 8) = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// ---- 4 rows per wave-instruction ------------------------------------------------------------
// 15 lanes per row, each lane turns ONE dword of the packed image (8 nibbles = 32 observation
// bits) into 32 output bytes (2 x 16-B stores): a wave writes the 4 x 480-B rows of 4 consecutive
// tables with two store instructions, and the nibble rotation costs the same for 8 nibbles as for 2.
struct GroupLane {
  int r;               // row (table) within the group of 4; 4 = idle lane (lanes 60..63)
  int ch;              // 32-bit chunk of the packed row, 0..14
  uint32_t keep_hist;  // chunk 0..12: all history; 13: packed bytes 52, 53(low nibble); 14: none
  uint32_t keep_hand;
  uint32_t keep_vul;   // chunk 0 carries the vulnerability nibble
  int hist_off;        // byte offset of this lane's history dword within the group's 4 images
  int hand_off;        // byte offset of the row's hand words
  uint32_t out_off;    // byte offset of this lane's 32 output bytes within the group's 4 rows
};

__device__ __forceinline__ GroupLane make_group_lane() {
  GroupLane g;
  int lane = (int)(threadIdx.x & 63u);
  g.r = lane / 15;
  g.ch = lane - g.r * 15;
  g.keep_hist = (g.ch < 13) ? 0xFFFFFFFFu : ((g.ch == 13) ? 0x00000FFFu : 0u);
  g.keep_hand = (g.ch < 13) ? 0u : 0xFFFFFFFFu;
  g.keep_vul = (g.ch == 0) ? 0xFu : 0u;
  int rr = (g.r < 4) ? g.r : 0;
  g.hist_off = rr * TABLE_BYTES + 4 * ((g.ch < 13) ? g.ch : 13);
  g.hand_off = rr * TABLE_BYTES + W_HAND * 8;
  g.out_off = (uint32_t)(rr * 480 + g.ch * 32);
  return g;
}

// The 4 x 38 mask bytes of a group are 152 contiguous bytes = 38 dwords: lane l < 38 writes dword
// l, whose 4 bytes belong to row qa (the first `split` of them) and row qa+1 (the rest).
struct MaskLane {
  int qa, qb;       // rows of the group this lane's bytes come from
  int sh;           // first action index within row qa
  int split;        // how many of the 4 bytes belong to row qa (1..4)
  uint32_t keep_a;  // nibble mask of the row-qa bytes
  bool active;      // lane < 38
};

__device__ __forceinline__ MaskLane make_mask_lane() {
  MaskLane m;
  int lane = (int)(threadIdx.x & 63u);
  int b0 = 4 * lane;
  m.active = lane < 38;
  m.qa = m.active ? b0 / 38 : 0;
  m.sh = m.active ? b0 - m.qa * 38 : 0;
  m.split = (38 - m.sh < 4) ? 38 - m.sh : 4;
  m.qb = (m.qa + 1 < 4) ? m.qa + 1 : 3;
  m.keep_a = (1u << m.split) - 1u;
  return m;
}

// legal_a / legal_b: the 64-bit legal masks of rows qa / qb (bits above 37 are ignored)
__device__ __forceinline__ uint32_t mask_dword(uint64_t legal_a, uint64_t legal_b, const MaskLane &m) {
  uint32_t sa = (uint32_t)(legal_a >> m.sh);
  uint32_t sb = (uint32_t)legal_b << m.split;
  uint32_t nib = (sa & m.keep_a) | (sb & (0xFu & ~m.keep_a));
  return __umul24(nib, 0x204081u) & 0x01010101u;
}

// this lane's 32 observation bytes of its row (seat / vulnib: the row's observer, per lane).
// Split into an LDS-load half and a compute+store half so that the loads of several groups can
// be in flight together (the emit path is LDS-latency bound, not issue bound).
__device__ __forceinline__ void obs_chunk_load(const uint8_t *img_group, int seat, const GroupLane &g, uint32_t &a,
                                               uint64_t &H) {
  a = *reinterpret_cast<const uint32_t *>(img_group + g.hist_off);
  H = *reinterpret_cast<const uint64_t *>(img_group + g.hand_off + seat * 8);
}

__device__ __forceinline__ void obs_chunk_store(uint32_t a, uint64_t H, int seat, uint32_t vulnib, uint8_t *dst_group,
                                                const GroupLane &g) {
  uint32_t m1 = (0xFu >> seat) * 0x11111111u;
  uint32_t rot = ((a >> seat) & m1) | ((a << (4 - seat)) & ~m1);
  uint32_t hv = (g.ch == 13) ? (uint32_t)(H << 8) : (uint32_t)(H >> 24);
  uint32_t word = (rot & g.keep_hist) | (hv & g.keep_hand) | (vulnib & g.keep_vul);
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = __umul24((word >> (4 * i)) & 0xFu, 0x204081u) & 0x01010101u;
  uint4 *dst = reinterpret_cast<uint4 *>(dst_group + g.out_off);
  dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
  dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
}

__device__ __forceinline__ void emit_obs_chunk(const uint8_t *img_group, int seat, uint32_t vulnib, uint8_t *dst_group,
                                               const GroupLane &g) {
  uint32_t a;
  uint64_t H;
  obs_chunk_load(img_group, seat, g, a, H);
  obs_chunk_store(a, H, seat, vulnib, dst_group, g);
}

__device__ __forceinline__ void emit_mask_row(uint64_t legal, uint8_t *dst_row, const LaneConst &c) {
  if (c.lane < 38) dst_row[c.lane] = (uint8_t)((legal >> c.lane) & 1ull);
}

// Deal one table's cards into its LDS image: zero the history, build the four hand words
// from the LUT key (wb5/vis_pgx.py:13-24 packing).  key words are wave-uniform.
__device__ __forceinline__ void deal_image(uint8_t *img, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3,
                                           const LaneConst &c) {
  uint32_t ksel = (c.dsuit == 0) ? q0 : ((c.dsuit == 1) ? q1 : ((c.dsuit == 2) ? q2 : q3));
  uint32_t owner = (ksel >> c.dshift) & 3u;
  bool card = c.lane < 52;
  uint64_t h0 = __ballot(card && owner == 0u);
  uint64_t h1 = __ballot(card && owner == 1u);
  uint64_t h2 = __ballot(card && owner == 2u);
  uint64_t h3 = __ballot(card && owner == 3u);
  uint64_t *img64 = reinterpret_cast<uint64_t *>(img);
  uint64_t hv = (c.lane == 7) ? h0 : ((c.lane == 8) ? h1 : ((c.lane == 9) ? h2 : h3));
  if (c.lane < 7) img64[c.lane] = 0ull;
  else if (c.lane < 11) img64[c.lane] = hv << 4;
}

// The same from the four packed hand words of the board's LUT row (precomputed once per LUT upload,
// k_lut_hands): no ballots, a re-deal is one 8-byte copy per lane.  `hands` points at 4 x uint64 in LDS.
__device__ __forceinline__ void deal_hands(uint8_t *img, const uint32_t *hands, const LaneConst &c) {
  if (c.lane < 11) {
    const uint64_t hv = *reinterpret_cast<const uint64_t *>(hands + 2 * ((c.lane > 7) ? c.lane - 7 : 0));
    reinterpret_cast<uint64_t *>(img)[c.lane] = (c.lane < 7) ? 0ull : hv;
  }
}

// LDS operations of ONE wave are performed in issue order, so a lane may read what another lane
// of the same wave wrote earlier without waiting; only the compiler must keep the order.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void wave_lds_fence() {
  // same-wave LDS hand-off between lanes: LDS ops of one wave execute in order; this only
  // stops the compiler from moving accesses across it.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void load_scalars(Tbl &t, const uint8_t *img) {
  const uint2 *p = reinterpret_cast<const uint2 *>(img);
  uint2 a = p[W_SC], b = p[W_FD], c = p[W_TR], d = p[W_CTR], e = p[W_REW];
  t.sc = a.x; t.sch = a.y; t.fd = b.x; t.t2 = b.y; t.t0 = c.x; t.t1 = c.y;
  t.lut = d.x; t.bctr = d.y; t.r01 = e.x; t.r23 = e.y;
}
__device__ __forceinline__ void store_scalars(const Tbl &t, uint8_t *img) {
  uint2 *p = reinterpret_cast<uint2 *>(img);
  p[W_SC] = make_uint2(t.sc, t.sch);
  p[W_FD] = make_uint2(t.fd, t.t2);
  p[W_TR] = make_uint2(t.t0, t.t1);
  p[W_CTR] = make_uint2(t.lut, t.bctr);
  p[W_REW] = make_uint2(t.r01, t.r23);
}

// uniform-random legal action: k-th legal action in ascending order, k = mulhi(draw, n).
__device__ __forceinline__ int random_legal_action(const Tbl &t, uint64_t legal, uint32_t draw, int &n_legal) {
  int n = __popcll(legal);
  n_legal = n;
  int k = (int)__umulhi(draw, (uint32_t)n);
  // structure of a live mask: bit 0, at most one of bits 1/2, then a contiguous run of bids
  uint32_t dbl = (uint32_t)(legal >> 1) & 3u;
  int first_bid = 3 + (int)bits(t.sc, SC_LB1, 6);
  int a_dbl = (dbl == 1u) ? 1 : 2;
  int a = (k == 0) ? 0 : (dbl ? ((k == 1) ? a_dbl : first_bid + k - 2) : first_bid + k - 1);
  return bits(t.sc, SC_MASKALL, 1) ? k : a;
}

// ---- lean, straight-line transition for the fused rollout's LOGIC wave ------------------------
// One uniform-random legal call on a LIVE table (never all-True mask, never illegal): same result as
// legal_mask() + random_legal_action() + auto_reset_clear() + auction_step(), with no data-dependent
// branches and no 64-bit popcount — this is the per-table dependency chain of the T-step scan, so
// every instruction here is paid 32 times in sequence.
struct LeanStep {
  uint64_t legal;   // legal_action_mask of the state BEFORE the call
  int seat;         // seat that acts
  int action;       // the call
  int n_legal;      // number of legal calls (for log_prob)
  uint32_t hb1;     // history bit + 1 (0: none)
  uint32_t term;    // auction over
};

__device__ __forceinline__ LeanStep lean_random_step(uint32_t &sc, uint32_t &sch, uint32_t u) {
  LeanStep r;
  const uint32_t lb1 = bits(sc, SC_LB1, 6);
  const uint32_t seat = (bits(sc, SC_DEALER, 2) + bits(sch, SCH_TURN, 9)) & 3u;
  const uint32_t own = ((bits(sc, SC_LBSEAT, 2) ^ seat) & 1u) ^ 1u;
  const uint32_t x = bits(sc, SC_X, 1), xx = bits(sc, SC_XX, 1), has = lb1 != 0;
  const uint32_t can_x = has & (own ^ 1u) & (x ^ 1u) & (xx ^ 1u);
  const uint32_t can_xx = has & own & x & (xx ^ 1u);
  const uint32_t dbl = can_x | can_xx;
  const uint64_t bids = (ALL_ACTIONS >> (3 + lb1)) << (3 + lb1);
  r.legal = bids | (uint64_t)(1u | (can_x << 1) | (can_xx << 2));
  const uint32_t n = 36u - lb1 + dbl;  // pass + (35 - lb1) bids + at most one of X / XX
  const uint32_t k = __umulhi(u, n);   // k-th legal call in ascending order
  const uint32_t a_bid = 2u + lb1 + k - dbl;
  const uint32_t a_dbl = can_x ? 1u : 2u;
  uint32_t a = (dbl & (k == 1u)) ? a_dbl : a_bid;
  a = (k == 0u) ? 0u : a;
  uint32_t nn = n;
  if (bits(sc, SC_MASKALL, 1)) {  // only a caller-supplied finished table can get here (all-True mask)
    r.legal = ALL_ACTIONS;
    nn = 38u;
    a = __umulhi(u, 38u);
  }
  r.action = (int)a;
  r.n_legal = (int)nn;
  r.seat = (int)seat;
  // A5 pre-step half of auto_reset (src/utils.py:34-43)
  const uint32_t was_term = bits(sc, SC_TERM, 1);
  sc &= ~(1u << SC_TERM);
  sch = was_term ? (sch & ~(1023u << SCH_STEP)) : sch;
  // the call
  const bool is_pass = a == 0u, is_bid = a >= 3u, is_x = a == 1u;
  const uint32_t b = a - 3u;
  const uint32_t hb_bid = 9u + 12u * b + seat;
  const uint32_t hb_dbl = 9u + 12u * (lb1 - 1u) + (is_x ? 4u : 8u) + seat;  // lb1 > 0 when X / XX is legal
  const uint32_t hb_pass = (lb1 == 0u) ? 5u + seat : 0u;
  r.hb1 = is_bid ? hb_bid : (is_pass ? hb_pass : hb_dbl);
  const uint32_t pass = is_pass ? bits(sc, SC_PASS, 3) + 1u : 0u;
  const uint32_t set_dbl = is_bid ? 0u : ((a == 1u ? (1u << SC_X) : 0u) | (a == 2u ? (1u << SC_XX) : 0u));
  const uint32_t bid_clear = (63u << SC_LB1) | (3u << SC_LBSEAT) | (1u << SC_X) | (1u << SC_XX);
  const uint32_t bid_set = ((b + 1u) << SC_LB1) | (seat << SC_LBSEAT);
  uint32_t nsc = is_bid ? ((sc & ~bid_clear) | bid_set) : (sc | set_dbl);
  const uint32_t nlb1 = is_bid ? b + 1u : lb1;
  const uint32_t term = pass == ((nlb1 != 0u) ? 3u : 4u);
  nsc = (nsc & ~(7u << SC_PASS)) | (pass << SC_PASS) | (term ? ((1u << SC_TERM) | (1u << SC_MASKALL)) : 0u);
  sc = nsc;
  sch += (1u << SCH_STEP) + (term ? 0u : (1u << SCH_TURN));
  r.term = term;
  return r;
}

// ---- the same transition, split for k_rollout_fs -----------------------------------------------------------
// Its logic wave keeps only what feeds the per-table dependency chain (lean_pick + lean_apply: the call of a uniform draw
// and the scalars after it); what the follower waves need besides — legal mask, n_legal, history bit — is recomputed
// off the chain, slot-parallel, by a helper wave (lean_legal, lean_hb1).  Together they equal lean_random_step.
__device__ __forceinline__ uint32_t lean_seat(uint32_t sc, uint32_t sch) {
  return (bits(sc, SC_DEALER, 2) + bits(sch, SCH_TURN, 9)) & 3u;
}

__device__ __forceinline__ void lean_doubles(uint32_t sc, uint32_t seat, uint32_t lb1, uint32_t &can_x, uint32_t &can_xx) {
  const uint32_t own = ((bits(sc, SC_LBSEAT, 2) ^ seat) & 1u) ^ 1u;
  const uint32_t x = bits(sc, SC_X, 1), xx = bits(sc, SC_XX, 1), has = lb1 != 0;
  can_x = has & (own ^ 1u) & (x ^ 1u) & (xx ^ 1u);
  can_xx = has & own & x & (xx ^ 1u);
}

// the uniform-random legal call of the player to act: k-th legal action in ascending order, k = mulhi(draw, n_legal)
__device__ __forceinline__ uint32_t lean_pick(uint32_t sc, uint32_t seat, uint32_t lb1, uint32_t u) {
  uint32_t can_x, can_xx;
  lean_doubles(sc, seat, lb1, can_x, can_xx);
  const uint32_t dbl = can_x | can_xx;
  const uint32_t n = 36u - lb1 + dbl;  // pass + (35 - lb1) bids + at most one of X / XX
  const uint32_t k = __umulhi(u, n);
  const uint32_t a_bid = 2u + lb1 + k - dbl;
  const uint32_t a_dbl = can_x ? 1u : 2u;
  uint32_t a = (dbl & (k == 1u)) ? a_dbl : a_bid;
  a = (k == 0u) ? 0u : a;
  if (bits(sc, SC_MASKALL, 1)) a = __umulhi(u, 38u);  // a caller-supplied finished table: all-True mask
  return a;
}

// A5 pre-step half of auto_reset (src/utils.py:34-43) + the call `a` by `seat`; returns "auction over"
__device__ __forceinline__ uint32_t lean_apply(uint32_t &sc, uint32_t &sch, uint32_t a, uint32_t seat, uint32_t lb1) {
  const uint32_t was_term = bits(sc, SC_TERM, 1);
  sc &= ~(1u << SC_TERM);
  sch = was_term ? (sch & ~(1023u << SCH_STEP)) : sch;
  const bool is_pass = a == 0u, is_bid = a >= 3u;
  const uint32_t b = a - 3u;
  const uint32_t pass = is_pass ? bits(sc, SC_PASS, 3) + 1u : 0u;
  const uint32_t set_dbl = is_bid ? 0u : ((a == 1u ? (1u << SC_X) : 0u) | (a == 2u ? (1u << SC_XX) : 0u));
  const uint32_t bid_clear = (63u << SC_LB1) | (3u << SC_LBSEAT) | (1u << SC_X) | (1u << SC_XX);
  const uint32_t bid_set = ((b + 1u) << SC_LB1) | (seat << SC_LBSEAT);
  uint32_t nsc = is_bid ? ((sc & ~bid_clear) | bid_set) : (sc | set_dbl);
  const uint32_t nlb1 = is_bid ? b + 1u : lb1;
  const uint32_t term = pass == ((nlb1 != 0u) ? 3u : 4u);
  nsc = (nsc & ~(7u << SC_PASS)) | (pass << SC_PASS) | (term ? ((1u << SC_TERM) | (1u << SC_MASKALL)) : 0u);
  sc = nsc;
  sch += (1u << SCH_STEP) + (term ? 0u : (1u << SCH_TURN));
  return term;
}

// legal_action_mask and its population count for the player to act at `seat`
__device__ __forceinline__ uint64_t lean_legal(uint32_t sc, uint32_t seat, uint32_t &n_legal) {
  const uint32_t lb1 = bits(sc, SC_LB1, 6);
  uint32_t can_x, can_xx;
  lean_doubles(sc, seat, lb1, can_x, can_xx);
  const uint64_t bids = (ALL_ACTIONS >> (3 + lb1)) << (3 + lb1);
  const bool all = bits(sc, SC_MASKALL, 1) != 0u;
  n_legal = all ? 38u : 36u - lb1 + (can_x | can_xx);
  return all ? ALL_ACTIONS : (bids | (uint64_t)(1u | (can_x << 1) | (can_xx << 2)));
}

// history bit + 1 (0: none) of call `a` by `seat` when the last bid + 1 was lb1
__device__ __forceinline__ uint32_t lean_hb1(uint32_t lb1, uint32_t seat, uint32_t a) {
  const bool is_pass = a == 0u, is_bid = a >= 3u, is_x = a == 1u;
  const uint32_t b = a - 3u;
  const uint32_t hb_bid = 9u + 12u * b + seat;
  const uint32_t hb_dbl = 9u + 12u * (lb1 - 1u) + (is_x ? 4u : 8u) + seat;  // lb1 > 0 when X / XX is legal
  const uint32_t hb_pass = (lb1 == 0u) ? 5u + seat : 0u;
  return is_bid ? hb_bid : (is_pass ? hb_pass : hb_dbl);
}

}  // namespace brl
