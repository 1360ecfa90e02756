// mlp_gemm_x3.hpp — the fp32 products of the PPO minibatch step (src/update.py:74-242; src/models.py:23-33: float32 hk.Linear)
// on the bf16 matrix pipe at fp32-grade error ("bf16x3", opt-in: config key gemm_precision).
//
//   * a float32 x is EXACTLY hi + mid + lo with three bf16 pieces (truncation splits: hi = top 16 bits of x, mid = top 16 bits
//     of x - hi, lo = x - hi - mid, which has at most 8 significant bits left); of the nine cross-products of two such operands
//     the six with weight >= 2^-16 (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi) carry everything above 2^-24 relative — the
//     three dropped ones are below the rounding of a single fp32 product.  Products are exact in the fp32 accumulator
//     (8 x 8 significant bits); v_mfma_f32_32x32x16_bf16 runs at 16 x the rate of the f32-input MFMA, so six of them cost
//     6/16 of the exact path's matrix time.
//   * accumulation: three accumulators per output block by magnitude class (hi.hi | hi.mid + mid.hi | the three small ones),
//     summed small -> large ONCE after the K loop: the large class takes K/16 roundings instead of the exact path's K.
//     Measured (scripts/micro/gemm_x3_test.hip, profiles/r06): max |err| against float64 0.3-0.45 x the exact fp32 chain's; with
//     ONE accumulator 0.65-1.46 x; nine products are no better than six; three (16-bit operands) are 17-90 x worse.
//   * operands stay fp32 in HBM (nothing upstream changes) and are staged RAW, exactly as mlp_gemm.hpp stages them: global ->
//     registers in full 128-byte lines (fragment-shaped loads straight to registers run 2.5-3 x slower: scripts/micro/l2_panel_bw.hip)
//     -> LDS (ds_write_b128; "KC" operand = summation index contiguous: 128-byte rows, 16-byte pieces XOR-swizzled by (row >> 1) & 7;
//     "MC" = output index contiguous: 256-byte K rows with bit 4 of the column XOR-ed with bit 2 of the K row).  A wave reads ITS
//     fragments' 8 floats per lane and 16-deep step (two conflict-free ds_read_b128, or eight ds_read_b32 down a column), splits them
//     in registers (44 VALU) and multiplies.  Earlier forms, measured and dropped (profiles/r06/r06_experiments.txt): split BEFORE the
//     LDS (three bf16 plane tiles: 24 KB of plane stores + 48 KB of fragment reads per 32-deep chunk keep the CU's one LDS pipe busy
//     ~490 cycles against 384 of MFMA: 1028-1185 cycles per chunk); no LDS at all with the K dimension divided among the waves
//     (fragment-shaped loads: 19.9 us for the loads alone).
//   * a lone wave per SIMD issues a vector instruction every ~4-6.5 cycles, two waves together one every 2: the workgroup is 512
//     threads = TWO K-groups of four waves: group g works on chunks g, g + 2, ... with its own LDS stages and accumulators (two
//     waves per SIMD, each the other's cover: one's split beside the other's MFMAs), the groups' sums are added through LDS in a
//     fixed order (deterministic).  64 x 64 output tile per workgroup, wave (wm, wn) of a group owns 32 x 32; the product is formed
//     transposed (the instruction's A operand is this kernel's B fragment): a lane ends with 4 x 4 consecutive output columns of
//     one row — 16-byte stores and gate loads; epilogues as in mlp_gemm.hpp.
#pragma once

#include "../../brl_amd/csrc/mlp_gemm.hpp"

namespace mgx {

using mg::Args;
using mg::BoolTag;
using mg::f32x4;
using mg::IntTag;
using mg::row16_sum;

constexpr int BK = 32, THREADS = 512, GT = 256;
constexpr int OPER = 64 * BK * 4;             // bytes of one raw operand tile (64 rows x 32 k, or 32 k x 64 columns, fp32)
constexpr int OFF_B = OPER, STAGE = 2 * OPER;
constexpr int LDS_BYTES = 4 * STAGE;          // two stages per K-group = 64 KB (the epilogue's exchange re-uses them)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef MGX_NPROD
#define MGX_NPROD 6    // cross-products per K step: 6 (default), 9 (all), 3 (hi.hi + hi.mid + mid.hi: a 16-bit-mantissa product, experiments)
#endif
#ifndef MGX_EXP
#define MGX_EXP 0      // timing experiments (wrong results): 1 = no split arithmetic, 2 = no MFMA, 4 = no LDS stores
#endif

// two floats -> the packed (low half = first) bf16 pair of each plane; exact: x == hi + mid + lo in fp32
// (msk = 0xFFFF0000 held in an SGPR by the caller: as a literal it makes each v_and an 8-byte instruction)
__device__ __forceinline__ void split2(float x0, float x1, unsigned msk, unsigned &hi, unsigned &mid, unsigned &lo) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float r0 = x0 - __uint_as_float(u0 & msk), r1 = x1 - __uint_as_float(u1 & msk);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  mid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float l0 = r0 - __uint_as_float(v0 & msk), l1 = r1 - __uint_as_float(v1 & msk);
  lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_tile(const Args &G, unsigned char *lds, int bid) {
  constexpr int NP = 4;                                               // 16-byte fp32 pieces per thread and chunk: 2 of A, 2 of B
  const int tid = (int)threadIdx.x, t = tid & (GT - 1), lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int tiles_m = (G.M + 63) / 64, tiles_n = (G.N + 63) / 64;
  if (bid >= tiles_m * tiles_n) return;
  int tm, tn;
  mg::tile_of(bid, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * 64, n0 = tn * 64;
  const int nchunks = (G.K + BK - 1) / BK, kfull = G.K / BK;
  const int ni = (nchunks - grp + 1) / 2;     // this group's chunks: global chunk 2 i + grp, i = 0 .. ni - 1
  const int NI = (nchunks + 1) / 2;           // phases (= barriers) both groups run
  unsigned char *gl = lds + grp * 2 * STAGE;  // this group's two stages

  // ---- staging (mlp_gemm.hpp's): a chunk of an operand tile is 512 pieces of 16 bytes, two per thread of the group.
  //   KC: piece q = row (q >> 3) of the tile, 16-byte piece (q & 7) of its 128 bytes of K -> LDS row * 128 + ((p ^ swz(row)) << 4)
  //   MC: piece q = K row q >> 4, piece q & 15 of its 64 columns                           -> LDS k * 256 + ((p ^ (k & 4)) << 4)
  // go = byte offset of the piece from the operand's chunk base, gs = the part of it that selects k (K tail: re-aim at k = 0)
  uint32_t go[NP], gs[NP];
  int kk[NP];          // k index of the piece's first element within the chunk
  int lw[NP];          // LDS byte offset (inside a stage) of the piece
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const bool isB = j >= 2;
    const int q = t + 256 * (j & 1);
    const bool kc = isB ? B_KC : A_KC;
    const int x0 = isB ? n0 : m0, X = isB ? G.N : G.M;
    const int64_t ld = isB ? G.ldb : G.lda;
    const int base = isB ? OFF_B : 0;
    if (kc) {
      const int row = q >> 3, p = q & 7;
      const int x = (x0 + row < X) ? x0 + row : X - 1;
      kk[j] = 4 * p;
      go[j] = (uint32_t)(((int64_t)x * ld + 4 * p) * 4);
      gs[j] = (uint32_t)(4 * p * 4);
      lw[j] = base + row * 128 + ((p ^ ((row >> 1) & 7)) << 4);
    } else {
      const int kr = q >> 4, p = q & 15;
      const int col = (x0 + 4 * p < X) ? x0 + 4 * p : 0;
      kk[j] = kr;
      go[j] = (uint32_t)(((int64_t)kr * ld + col) * 4);
      gs[j] = (uint32_t)(((int64_t)kr * ld) * 4);
      lw[j] = base + kr * 256 + ((p ^ (kr & 4)) << 4);
    }
  }
  // a group walks K two chunks at a time
  const uint32_t stepa = (uint32_t)((A_KC ? (int64_t)BK : (int64_t)BK * G.lda) * 4), stepb = (uint32_t)((B_KC ? (int64_t)BK : (int64_t)BK * G.ldb) * 4);
  const __amdgpu_buffer_rsrc_t srda = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.B), (short)0, 0x7FFFFFFF, 0x00020000);
  uint32_t soa = (uint32_t)grp * stepa, sob = (uint32_t)grp * stepb;
  unsigned msk;
  asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(msk));   // (opaque to the compiler: stays an SGPR operand)
  f32x4 rg[2][NP];      // two of the group's chunks in flight: its chunk i lives in set i & 1
  // piece j of the group's chunk i (global chunk c = 2 i + grp) -> set S; a partial last chunk re-aims pieces beyond K at k = 0
  auto gload = [&](auto set_tag, int j, int c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    const bool isB = j >= 2;
    const uint32_t off = go[j] - ((c < kfull || kk[j] < G.K - c * BK) ? 0u : gs[j]);
    rg[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isB ? srdb : srda, (int)off, (int)(isB ? sob : soa), 0));
  };
  auto gload_full = [&](auto set_tag, int j) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    const bool isB = j >= 2;
    rg[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isB ? srdb : srda, (int)go[j], (int)(isB ? sob : soa), 0));
  };
  auto gadvance = [&]() __attribute__((always_inline)) { soa += 2 * stepa; sob += 2 * stepb; };
  // piece j of global chunk c, held in set S -> stage st (raw)
  auto stage_piece = [&](auto set_tag, auto full_tag, int j, int c, unsigned char *st) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    f32x4 v = rg[S][j];
    if (!FULL && c >= kfull && kk[j] >= G.K - c * BK) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // (KC pieces are 4 k wide and K % 4 == 0: whole)
    if (!(MGX_EXP & 4)) *reinterpret_cast<f32x4 *>(st + lw[j]) = v;
  };

  // ---- fragments.  v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, h = lane >> 5) holds row r, k elements 8 h + j of the 16-deep
  // step: k = 16 s + 8 h + j of the chunk.
  //   KC: pieces 4 s + 2 h + e (e = 0, 1) of the row: two ds_read_b128, conflict-free under the staging swizzle
  //   MC: eight ds_read_b32 down column r of K rows 16 s + 8 h + j (rows with bit 2 set hold the column XOR 16: j >= 4)
  const int wm = w >> 1, wn = w & 1, r32 = lane & 31, hh = lane >> 5;
  int fa[2], fb[2];    // KC: byte offsets of the step-0 pieces e = 0, 1 (step 1: piece index + 4 -> offset ^ 64); MC: bases for j < 4 / j >= 4
  {
    const int ra = wm * 32 + r32, rb = wn * 32 + r32;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      fa[e] = A_KC ? ra * 128 + (((2 * hh + e) ^ ((ra >> 1) & 7)) << 4) : 8 * hh * 256 + ((ra ^ (16 * e)) << 2);
      fb[e] = OFF_B + (B_KC ? rb * 128 + (((2 * hh + e) ^ ((rb >> 1) & 7)) << 4) : 8 * hh * 256 + ((rb ^ (16 * e)) << 2));
    }
  }
  f32x4 raw[4];      // the 16-deep step being fetched: [2 operand + e]: A's elements 0..3 / 4..7, B's
  // read r (0..3: operand r >> 1, half e = r & 1) of step s of stage st
  auto read_raw = [&](const unsigned char *st, int s, int r) __attribute__((always_inline)) {
    const bool isB = r >= 2;
    const int e = r & 1;
    const bool kc = isB ? B_KC : A_KC;
    if (kc) {
      raw[r] = *reinterpret_cast<const f32x4 *>(st + ((isB ? fb[e] : fa[e]) ^ (s << 6)));
    } else {
      const unsigned char *p = st + (isB ? fb[e] : fa[e]) + (16 * s + 4 * e) * 256;
      raw[r] = f32x4{*reinterpret_cast<const float *>(p), *reinterpret_cast<const float *>(p + 256), *reinterpret_cast<const float *>(p + 512),
                     *reinterpret_cast<const float *>(p + 768)};
    }
  };
  // split u (0..7: operand u >> 2, half e = (u >> 1) & 1, pair u & 1) of the fetched step -> fragment set f ([0..2] = A hi / mid / lo,
  // [3..5] = B): register 2 e + pair of each plane
  unsigned fr[2][6][4];
  auto split_raw = [&](auto fs_tag, int u) __attribute__((always_inline)) {
    constexpr int FS = decltype(fs_tag)::value;
    const int op = u >> 2, e = (u >> 1) & 1, pr = u & 1;
    const f32x4 v = raw[2 * op + e];
    unsigned h, m, l;
    if (MGX_EXP & 1) h = m = l = __float_as_uint(pr ? v.z : v.x) ^ __float_as_uint(pr ? v.w : v.y);
    else split2(pr ? v.z : v.x, pr ? v.w : v.y, msk, h, m, l);
    fr[FS][3 * op + 0][2 * e + pr] = h;
    fr[FS][3 * op + 1][2 * e + pr] = m;
    fr[FS][3 * op + 2][2 * e + pr] = l;
  };
  auto frag = [&](auto fs_tag, int u) __attribute__((always_inline)) -> bf16x8 {
    constexpr int FS = decltype(fs_tag)::value;
    return __builtin_bit_cast(bf16x8, u32x4{fr[FS][u][0], fr[FS][u][1], fr[FS][u][2], fr[FS][u][3]});
  };

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
  // MFMA i (0 .. NPROD - 1) of a step's products on fragment set FS, small classes first
  auto mf = [&](auto fs_tag, int i) __attribute__((always_inline)) {
    const int k = i + (9 - MGX_NPROD);     // position in the nine-product order: (B plane . A plane)
    const int pb = (k == 0 || k == 1 || k == 3) ? 2 : (k == 2 || k == 5 || k == 6) ? 1 : 0;
    const int pa = (k == 0 || k == 2 || k == 4) ? 2 : (k == 1 || k == 5 || k == 7) ? 1 : 0;
    const int cls = k < 6 ? 2 : k < 8 ? 1 : 0;
    if (!(MGX_EXP & 2)) acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(fs_tag, 3 + pb), frag(fs_tag, pa), acc[cls], 0, 0, 0);
    else acc[cls][i] += __uint_as_float(fr[decltype(fs_tag)::value][3 + pb][i & 3] ^ fr[decltype(fs_tag)::value][pa][i & 3]);
  };

  MG_STAMP(0);
  // ---- prologue: the group's chunk 0 -> its stage 0, its chunks 1 and 2 in flight
  if (ni > 0) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<0>{}, j, grp);
    gadvance();
  }
  if (ni > 1) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<1>{}, j, 2 + grp);
    gadvance();
  }
  if (ni > 0) {
#pragma unroll
    for (int j = 0; j < NP; j++) stage_piece(IntTag<0>{}, BoolTag<false>{}, j, grp, gl);
  }
  if (ni > 2) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<0>{}, j, 4 + grp);
    gadvance();
  }
  __syncthreads();
  if (ni > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) read_raw(gl, 0, r);
#pragma unroll
    for (int u = 0; u < 8; u++) split_raw(IntTag<0>{}, u);
#pragma unroll
    for (int r = 0; r < 4; r++) read_raw(gl, 1, r);
  }
  MG_STAMP(1);
  // ---- the K loop.  Phase i of a group = the 2 NPROD MFMAs of its chunk i, each followed by a slot (sched_barrier pins the order).
  // With NPROD = 6 (H = 6 MFMAs per 16-deep step):
  //   slots 0 .. 5      step 1's floats (fetched during the previous phase) split into fragment set 1: 8 splits of 11 VALU (2 1 2 1 1 1)
  //   slots 0 .. 3      one piece of the group's chunk i + 1 -> the other stage (ds_write_b128); slots 2 .. 5: the requests of chunk i + 3
  //   slot 5            every LDS operation of the wave done, barrier (all eight waves: both groups run NI phases)
  //   slot 6            step 0 of chunk i + 1 fetched from the other stage
  //   slots 7 .. 10     ... and split into fragment set 0 (2 per slot)
  //   slot 11           step 1 of chunk i + 1 fetched (split during the next phase's first half)
  auto phase = [&](auto full_tag, auto par_tag, int i) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;      // i & 1: the stage read; chunk i + 1 is staged from register set PAR ^ 1,
    constexpr int NS = PAR ^ 1;                        // which then takes the request of chunk i + 3
    constexpr int H = MGX_NPROD, NM = 2 * H;
    unsigned char *sn = gl + NS * STAGE;
    const bool real = FULL || i < ni, nxt = FULL || i + 1 < ni, nxt2 = FULL || i + 3 < ni;
    const int c1 = 2 * (i + 1) + grp, c3 = 2 * (i + 3) + grp;
    auto slot = [&](int s) __attribute__((always_inline)) {
      if (s < H && real) {            // the 8 splits of step 1 over the first H slots
        const int lo = (8 * s) / H, hi = (8 * (s + 1)) / H;
#pragma unroll
        for (int u = lo; u < hi; u++) split_raw(IntTag<1>{}, u);
      }
      if (s < NP && nxt) stage_piece(IntTag<NS>{}, full_tag, s, c1, sn);
      if (s >= 2 && s < 2 + NP && nxt2) {
        if (FULL) gload_full(IntTag<NS>{}, s - 2); else gload(IntTag<NS>{}, s - 2, c3);
        if (s == 1 + NP) gadvance();
      }
      if (s == H - 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (s == H && nxt) {
#pragma unroll
        for (int r = 0; r < 4; r++) read_raw(sn, 0, r);
      }
      if (s > H && s < NM - 1 && nxt) {     // the 8 splits of the next chunk's step 0 over slots H + 1 .. NM - 2
        const int n = NM - 2 - H, k = s - H - 1;
        const int lo = (8 * k) / n, hi = (8 * (k + 1)) / n;
#pragma unroll
        for (int u = lo; u < hi; u++) split_raw(IntTag<0>{}, u);
      }
      if (s == NM - 1 && nxt) {
#pragma unroll
        for (int r = 0; r < 4; r++) read_raw(sn, 1, r);
      }
    };
#define MGX_STEP(s)                                                                                  \
    if ((s) < NM) {                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (real) { if ((s) < H) mf(IntTag<0>{}, (s)); else mf(IntTag<1>{}, (s) - H); }                 \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      slot(s);                                                                                       \
    }
    MGX_STEP(0) MGX_STEP(1) MGX_STEP(2) MGX_STEP(3) MGX_STEP(4) MGX_STEP(5) MGX_STEP(6) MGX_STEP(7) MGX_STEP(8)
    MGX_STEP(9) MGX_STEP(10) MGX_STEP(11) MGX_STEP(12) MGX_STEP(13) MGX_STEP(14) MGX_STEP(15) MGX_STEP(16) MGX_STEP(17)
    __builtin_amdgcn_sched_barrier(0);
#undef MGX_STEP
  };
  {
    int i = 0;
    // FULL phase i: the group's chunks i + 1 and i + 3 exist and are whole: 2 (i + 3) + grp < kfull
    for (; 2 * (i + 4) + grp < kfull; i += 2) {
      phase(BoolTag<true>{}, IntTag<0>{}, i);
      phase(BoolTag<true>{}, IntTag<1>{}, i + 1);
    }
    for (; i < NI; i++) {
      if (i & 1) phase(BoolTag<false>{}, IntTag<1>{}, i);
      else phase(BoolTag<false>{}, IntTag<0>{}, i);
    }
  }
  MG_STAMP(2);

  // the epilogue's operands are requested here (in the loop they would cost 16 registers of a 256-register budget: two waves per
  // SIMD); they land behind the exchange of the groups' sums
  f32x4 ebias[4], egate[4];
  const int em = m0 + 32 * wm + r32, emc = em < G.M ? em : G.M - 1;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int n = n0 + 32 * wn + 8 * q + 4 * hh, nc = n < G.N ? n : 0;
    if (EPI == mg::EPI_BIAS_ACT) ebias[q] = *reinterpret_cast<const f32x4 *>(G.bias + nc);
    if (EPI == mg::EPI_GATE_COLSUM) egate[q] = *reinterpret_cast<const f32x4 *>(G.gate + (int64_t)emc * G.ldg + nc);
  }
  // ---- the groups' sums: classes small -> large per group, then group 0 + group 1 through LDS (fixed order)
  f32x16 v = acc[2];
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = (v[e] + acc[1][e]) + acc[0][e];
  __syncthreads();        // every wave is done with the stages
  float *xch = reinterpret_cast<float *>(lds);
  if (grp == 1) {
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4 *>(xch + (q * GT + t) * 4) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 o = *reinterpret_cast<const f32x4 *>(xch + (q * GT + t) * 4);
#pragma unroll
      for (int i = 0; i < 4; i++) v[4 * q + i] += o[i];
    }
  }
  // ---- epilogue (group 0): lane holds row em, columns n0 + 32 wn + 8 q + 4 hh + (0..3), q = register >> 2
  const bool relu = G.act == 0;
  float sq = 0.0f;
  float *red = xch + 4 * GT * 4;     // behind the exchange image
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int n = n0 + 32 * wn + 8 * q + 4 * hh;
    const bool ok = grp == 0 && em < G.M && n < G.N;
    f32x4 o = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    if (EPI == mg::EPI_BIAS_ACT) {
      const f32x4 bb = ebias[q];
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = fmaxf(o[i] + bb[i], 0.0f);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = tanhf(o[i] + bb[i]);
      }
    }
    if (EPI == mg::EPI_GATE_COLSUM) {
      const f32x4 g = egate[q];
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = g[i] > 0.0f ? o[i] : 0.0f;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = o[i] * (1.0f - g[i] * g[i]);
      }
      if (G.colsum != nullptr) {
        // column sums of the tile's 64 rows, fixed order: the 16 lanes of a DPP row, the half's two rows, wave wm = 0 + wave wm = 1
        f32x4 cs;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float c = row16_sum(ok ? o[i] : 0.0f);
          c += __shfl_xor(c, 16, 64);
          cs[i] = c;
        }
        if (grp == 0 && r32 == 0) *reinterpret_cast<f32x4 *>(red + wm * 64 + 32 * wn + 8 * q + 4 * hh) = cs;
      }
    }
    if (EPI == mg::EPI_SQSUM) {
#pragma unroll
      for (int i = 0; i < 4; i++) sq += ok ? o[i] * o[i] : 0.0f;
    }
    if (ok) *reinterpret_cast<f32x4 *>(G.C + (int64_t)em * G.ldc + n) = o;
  }
  if (EPI == mg::EPI_GATE_COLSUM && G.colsum != nullptr) {
    __syncthreads();
    if (tid < 64 && n0 + tid < G.N) G.colsum[(int64_t)tm * G.N + n0 + tid] = red[tid] + red[64 + tid];
  }
  if (EPI == mg::EPI_SQSUM && G.sqsum != nullptr) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
    if (grp == 0 && lane == 0) red[w] = sq;
    __syncthreads();
    if (tid == 0) G.sqsum[tm * tiles_n + tn] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  MG_STAMP(3);
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(THREADS) void k_gemm_x3(Args G) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  gemm_tile<A_KC, B_KC, EPI>(G, lds, (int)blockIdx.x);
}

}  // namespace mgx
