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
//   * operands stay fp32 in HBM (nothing upstream changes): a K chunk is staged global -> registers (buffer loads, scalar K
//     offset) -> split in registers (11 VALU per two elements) -> three bf16 plane tiles in LDS.  "KC" operand (summation index
//     contiguous): 64-byte plane rows, 16-byte pieces XOR-swizzled by (row >> 2) & 3, fragments by one conflict-free ds_read_b128;
//     the K order inside a 32-deep chunk is permuted identically for both operands (piece P holds k = 4P..4P+3 and
//     16+4P..16+4P+3).  "MC" operand (output index contiguous): plane rows [k][x] of 128 bytes, the 64-byte halves swapped on k
//     rows 2, 3 mod 4; fragments by two ds_read_b64_tr_b16 (k rows 4P.. and 16+4P..: the same K order), conflict-free.
//   * what bounds it is NOT the matrix pipe: per 32-deep chunk a 64 x 64 tile needs 88 VALU instructions per thread for the split
//     (~5 cycles each beside MFMAs: ~450 cycles per SIMD), ~490 cycles of the CU's LDS pipe (24 KB of plane stores at ~80 B/clk,
//     48 KB of fragment reads at 256 B/clk) and 384 cycles of MFMA.  One wave per SIMD cannot overlap its own stalls (1185 cycles
//     per chunk measured), so the workgroup is 512 threads = TWO K-groups of four waves: group g works on chunks g, g + 2, ...
//     with its own LDS stages and accumulators (two waves per SIMD, each the other's cover), the groups' sums are added through
//     LDS in a fixed order (deterministic).  64 x 64 output tile per workgroup, wave (wm, wn) of a group owns 32 x 32; the
//     product is formed transposed (the instruction's A operand is this kernel's B fragment): a lane ends with 4 x 4 consecutive
//     output columns of one row — 16-byte stores and gate loads; epilogues as in mlp_gemm.hpp.
#pragma once

#include "../../brl_amd/csrc/mlp_gemm.hpp"

namespace mgx {

using mg::Args;
using mg::BoolTag;
using mg::f32x4;
using mg::IntTag;
using mg::row16_sum;

constexpr int BK = 32, THREADS = 512, GT = 256;
constexpr int PLANE = 64 * BK * 2;            // bytes of one bf16 plane tile (64 rows x 32 k, or 32 k x 64 columns)
constexpr int OFF_B = 3 * PLANE, STAGE = 6 * PLANE;
constexpr int LDS_BYTES = 4 * STAGE;          // two stages per K-group = 96 KB (the epilogue's exchange re-uses them)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef MGX_NPROD
#define MGX_NPROD 6    // cross-products per K step: 6 (default), 9 (all), 3 (hi.hi + hi.mid + mid.hi: a 16-bit-mantissa product, experiments)
#endif
#ifndef MGX_EXP
#define MGX_EXP 0      // timing experiments (wrong results): 1 = no split arithmetic, 2 = no MFMA, 4 = no plane stores
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

  // ---- staging.  KC operand, 64 rows x 32 k = 512 loads of 4 k: item q = t + 256 jj -> row = q >> 3, P = (q >> 1) & 3,
  //   half = q & 1: k = 16 half + 4 P (8 lanes load one 128-byte line; 16 consecutive lanes store two whole 64-byte plane rows)
  //   MC operand, 32 k rows x 64 columns = 512 loads of 4 columns: item q -> k row q >> 4, columns 4 (q & 15) ..
  // go = byte offset of the piece from the operand's chunk base, gs = the part of it that selects k (K tail: re-aim at k = 0)
  uint32_t go[NP], gs[NP];
  int kk[NP];          // k index of the piece's first element within the chunk
  int lw[NP];          // LDS byte offset (inside a stage, plane 0) the piece's split goes to
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const bool isB = j >= 2;
    const int q = t + 256 * (j & 1);
    const bool kc = isB ? B_KC : A_KC;
    const int x0 = isB ? n0 : m0, X = isB ? G.N : G.M;
    const int64_t ld = isB ? G.ldb : G.lda;
    const int base = isB ? OFF_B : 0;
    if (kc) {
      const int row = q >> 3, P = (q >> 1) & 3, half = q & 1;
      const int x = (x0 + row < X) ? x0 + row : X - 1;
      kk[j] = 16 * half + 4 * P;
      go[j] = (uint32_t)(((int64_t)x * ld + kk[j]) * 4);
      gs[j] = (uint32_t)(kk[j] * 4);
      lw[j] = base + row * 64 + ((P ^ ((row >> 2) & 3)) << 4) + 8 * half;
    } else {
      const int kr = q >> 4, p = q & 15;
      const int col = (x0 + 4 * p < X) ? x0 + 4 * p : 0;
      kk[j] = kr;
      go[j] = (uint32_t)(((int64_t)kr * ld + col) * 4);
      gs[j] = (uint32_t)(((int64_t)kr * ld) * 4);
      lw[j] = base + kr * 128 + ((p ^ (((kr >> 1) & 1) << 3)) << 3);
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
  unsigned sh[3][2];   // the piece being split: [plane][half]
  // half `half` (elements 0, 1 / 2, 3) of piece j of global chunk c, held in set S: split; behind the second half the three
  // plane stores (8 bytes each) into stage st
  auto stage_half = [&](auto set_tag, auto full_tag, int j, int half, int c, unsigned char *st) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    f32x4 v = rg[S][j];
    if (!FULL && c >= kfull && kk[j] >= G.K - c * BK) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // (KC pieces are 4 k wide and K % 4 == 0: whole)
    if (MGX_EXP & 1) sh[0][half] = sh[1][half] = sh[2][half] = __float_as_uint(half ? v.z : v.x) ^ __float_as_uint(half ? v.w : v.y);
    else split2(half ? v.z : v.x, half ? v.w : v.y, msk, sh[0][half], sh[1][half], sh[2][half]);
    if (half && !(MGX_EXP & 4)) {
      unsigned char *p = st + lw[j];
      *reinterpret_cast<u32x2 *>(p) = u32x2{sh[0][0], sh[0][1]};
      *reinterpret_cast<u32x2 *>(p + PLANE) = u32x2{sh[1][0], sh[1][1]};
      *reinterpret_cast<u32x2 *>(p + 2 * PLANE) = u32x2{sh[2][0], sh[2][1]};
    }
  };

  // ---- fragments.  v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, h = lane >> 5) holds row r, k elements 8 h + j of the 16-deep
  // step; step s of the chunk takes piece P = 2 s + h: k = 4 P + (0..3), 16 + 4 P + (0..3).
  //   KC: one ds_read_b128 at row * 64 + ((P ^ swz(row)) << 4); step 1 = step 0 ^ 32
  //   MC: two ds_read_b64_tr_b16 (lane 4 q + p of a 16-lane group supplies k row q, columns 4 p ..; lane i receives column i):
  //       k rows 4 P + q and 16 + 4 P + q -> lane base + 1024 s + 2048 e
  const int wm = w >> 1, wn = w & 1, r32 = lane & 31, hh = lane >> 5;
  int fa, fb;
  {
    const int qq = (lane >> 2) & 3, xg = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int ra = wm * 32 + r32, rb = wn * 32 + r32;
    fa = A_KC ? ra * 64 + ((hh ^ ((ra >> 2) & 3)) << 4) : (4 * hh + qq) * 128 + ((((wm * 32 + xg) >> 2) ^ (((qq >> 1) & 1) << 3)) << 3);
    fb = OFF_B + (B_KC ? rb * 64 + ((hh ^ ((rb >> 2) & 3)) << 4) : (4 * hh + qq) * 128 + ((((wn * 32 + xg) >> 2) ^ (((qq >> 1) & 1) << 3)) << 3));
  }
  // fragment u of step s: u = 0..2 = A hi / mid / lo, 3..5 = B hi / mid / lo
  auto read_frag = [&](const unsigned char *st, int s, int u) __attribute__((always_inline)) -> bf16x8 {
    const bool isB = u >= 3;
    const bool kc = isB ? B_KC : A_KC;
    const int plane = (isB ? u - 3 : u) * PLANE;
    if (kc) {
      const int off = (isB ? fb : fa) ^ (s << 5);
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(st + off + plane));
    }
    const unsigned char *p = st + (isB ? fb : fa) + plane + 1024 * s;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 2048));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, s16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
  };

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
#define MGX_MF(c, x, y) if (!(MGX_EXP & 2)) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[c], 0, 0, 0)
  // MFMA i (0 .. NPROD - 1) of a step's products, small classes first
  auto mf = [&](const bf16x8 (&f)[6], int i) __attribute__((always_inline)) {
    const int k = i + (9 - MGX_NPROD);     // position in the nine-product order
    switch (k) {
      case 0: MGX_MF(2, f[5], f[2]); break;
      case 1: MGX_MF(2, f[5], f[1]); break;
      case 2: MGX_MF(2, f[4], f[2]); break;
      case 3: MGX_MF(2, f[5], f[0]); break;
      case 4: MGX_MF(2, f[3], f[2]); break;
      case 5: MGX_MF(2, f[4], f[1]); break;
      case 6: MGX_MF(1, f[4], f[0]); break;
      case 7: MGX_MF(1, f[3], f[1]); break;
      default: MGX_MF(0, f[3], f[0]); break;
    }
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
    for (int j = 0; j < NP; j++) {
      stage_half(IntTag<0>{}, BoolTag<false>{}, j, 0, grp, gl);
      stage_half(IntTag<0>{}, BoolTag<false>{}, j, 1, grp, gl);
    }
  }
  if (ni > 2) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<0>{}, j, 4 + grp);
    gadvance();
  }
  __syncthreads();
  bf16x8 f0[6], f1[6];
  if (ni > 0) {
#pragma unroll
    for (int u = 0; u < 6; u++) f0[u] = read_frag(gl, 0, u);
  }
  MG_STAMP(1);
  // ---- the K loop.  Phase i of a group = the 2 NPROD MFMAs of its chunk i, each followed by a slot (sched_barrier pins the order):
  //   slots 0 .. 5          the step-1 fragments of this stage
  //   slots 0 .. 7          half a piece of the group's chunk i + 1 split in registers (11 VALU); behind a piece's second half its
  //                         three plane stores into the other stage
  //   slots 8, 9            the requests of the group's chunk i + 3 into the registers just stored
  //   slot 2 NPROD - 4      every LDS operation of the wave done, barrier (all eight waves: both groups run NI phases)
  //   the last three slots  the step-0 fragments of chunk i + 1 (two each)
  auto phase = [&](auto full_tag, auto par_tag, int i) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;      // i & 1: the stage read; chunk i + 1 is staged from register set PAR ^ 1,
    constexpr int NS = PAR ^ 1;                        // which then takes the request of chunk i + 3
    constexpr int NM = 2 * MGX_NPROD, BAR = NM - 4;
    unsigned char *st = gl + PAR * STAGE, *sn = gl + NS * STAGE;
    const bool real = FULL || i < ni, nxt = FULL || i + 1 < ni, nxt2 = FULL || i + 3 < ni;
    const int c1 = 2 * (i + 1) + grp, c3 = 2 * (i + 3) + grp;
    auto slot = [&](int s) __attribute__((always_inline)) {
      if (s < 6 && real) f1[(s & 1) * 3 + (s >> 1)] = read_frag(st, 1, (s & 1) * 3 + (s >> 1));
      if (s < 2 * NP && nxt) stage_half(IntTag<NS>{}, full_tag, s >> 1, s & 1, c1, sn);
      if (s >= 2 * NP && s < 2 * NP + 2 && nxt2) {
#pragma unroll
        for (int j = (s - 2 * NP) * 2; j < (s - 2 * NP) * 2 + 2; j++) {
          if (FULL) gload_full(IntTag<NS>{}, j); else gload(IntTag<NS>{}, j, c3);
        }
        if (s == 2 * NP + 1) gadvance();
      }
      if (s == BAR) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (s > BAR && nxt) {
        const int q = 2 * (s - BAR - 1);
#pragma unroll
        for (int u = q; u < q + 2; u++) f0[(u & 1) * 3 + (u >> 1)] = read_frag(sn, 0, (u & 1) * 3 + (u >> 1));
      }
    };
#define MGX_STEP(s)                                                                                  \
    if ((s) < NM) {                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if (real) { if ((s) < MGX_NPROD) mf(f0, (s)); else mf(f1, (s) - MGX_NPROD); }                   \
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
#undef MGX_MF
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
