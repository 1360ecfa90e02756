// mlp_gemm_x3w.hpp — "bf16x3" products (see mlp_gemm_x3.hpp for the arithmetic: three exact bf16 pieces per fp32 operand, six
// v_mfma_f32_32x32x16_bf16 products per K step, three accumulators by magnitude class), second structure: NO LDS in the K loop.
//
// What bounded the LDS-staged form (mlp_gemm_x3.hpp: 1028-1185 cycles per 32-deep chunk against 384 of MFMA) was the staging
// itself: 24 KB of plane stores + 48 KB of fragment reads per chunk through the CU's one LDS pipe (~490 cycles), the split's
// 88 vector instructions per thread (~450 cycles per SIMD) and a barrier per chunk that keeps the four waves in step, so that
// nobody's stall is covered.  Here the K dimension is divided among the four WAVES instead: a 256-thread workgroup owns a
// 64 x 64 output tile, wave w owns the 32-deep K steps w, w + 4, w + 8, ... of the WHOLE tile (2 x 2 blocks of 32 x 32, three
// accumulators each: 192 accumulator registers of the 512 a lone wave per SIMD may hold).  A wave loads ITS K slice of both
// operands straight into registers in fragment order (lane (r, h) of a block: 16 bytes of row r per load — the K order inside a
// step is permuted identically for both operands), splits it in registers and multiplies: every operand element is still loaded
// once and split once per workgroup, nothing goes through LDS, there is no barrier in the loop, and the four waves drift apart
// as their loads land.  Raw data is four half-steps deep in registers (requests are three half-steps = ~2500 cycles old when
// their split starts).  After the loop the four waves' partial tiles are added through LDS in a fixed order (wave 0 + 1 + 2 + 3:
// deterministic) and wave w runs the epilogue of block w.
//
// Layouts: "KC" operands only in this file (the summation index contiguous: the forward pass, dz as dh's A operand).
#pragma once

#include "../../brl_amd/csrc/mlp_gemm.hpp"
#include "mlp_gemm_x3.hpp"

namespace mgw {

using mg::Args;
using mg::f32x4;
using mg::row16_sum;
using mgx::bf16x8;
using mgx::f32x16;
using mgx::split2;
using mgx::u32x4;

constexpr int THREADS = 256;
#ifndef MGW_EXP
#define MGW_EXP 0      // timing experiments (wrong results): 1 = no split arithmetic, 2 = no MFMA, 8 = no loads in the loop
#endif
#ifndef MGW_NACC
#define MGW_NACC 2     // accumulators per block: 2 = hi.hi | the five smaller products; 3 = hi.hi | hi.mid + mid.hi | the three small ones
#endif
constexpr int LDS_BYTES = 4 * 4 * 4 * 64 * 16;     // the epilogue's exchange: [source wave][block][register group][lane] float4 = 64 KB

template <int V>
struct IntC { static constexpr int value = V; };

template <int EPI>
__device__ __forceinline__ void gemm_tile_nt(const Args &G, unsigned char *lds, int bid) {
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (G.M + 63) / 64, tiles_n = (G.N + 63) / 64;
  if (bid >= tiles_m * tiles_n) return;
  int tm, tn;
  mg::tile_of(bid, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * 64, n0 = tn * 64;
  const int r32 = lane & 31, hh = lane >> 5;
  const int nsteps = (G.K + 31) / 32;
  const int nw = (nsteps - w + 3) / 4;          // this wave's steps: j = w + 4 n, n = 0 .. nw - 1
  const int Q = 2 * nw;                         // ... as half-steps q = 2 n + s

  // block-operand bo: 0, 1 = rows m0 .. + 31, m0 + 32 .. of A; 2, 3 = rows n0 .., n0 + 32 .. of B.  Lane (r, h) loads, for load i
  // (0..3) of a step, bytes 32 i + 16 h .. + 15 of its row's 128-byte line: floats k = 8 i + 4 h + (0..3); half-step s = loads 2 s,
  // 2 s + 1 = one 16-deep MFMA step whose 8 elements per lane are k = 16 s + 4 h + (0..3), 16 s + 8 + 4 h + (0..3).
  uint32_t vo[4];
#pragma unroll
  for (int bo = 0; bo < 4; bo++) {
    const bool isB = bo >= 2;
    const int x0 = (isB ? n0 : m0) + 32 * (bo & 1), X = isB ? G.N : G.M;
    const int x = (x0 + r32 < X) ? x0 + r32 : X - 1;
    vo[bo] = (uint32_t)(((int64_t)x * (isB ? G.ldb : G.lda) + 4 * hh) * 4);
  }
  const __amdgpu_buffer_rsrc_t srda = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.B), (short)0, 0x7FFFFFFF, 0x00020000);
  unsigned msk;
  asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(msk));

  f32x4 R[4][8];            // raw half-steps: [ring slot][2 bo + ii]
  unsigned F[2][4][3][4];   // fragments: [set][bo][plane][register]: register pp = elements 2 pp, 2 pp + 1
  f32x16 acc[2][2][MGW_NACC];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int c = 0; c < MGW_NACC; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[a][b][c][e] = 0.0f;

  // load u (0..7: bo = u >> 1, ii = u & 1) of half-step q into ring slot RS; a piece beyond K (the last step of a K that is not a
  // multiple of 32) is re-aimed at the step's first piece (inside the matrix) and zeroed by the split.  The scalar offset stays
  // wave-uniform: a per-lane one makes hipcc wrap every load in a waterfall loop
  auto load1 = [&](auto rs_tag, auto s_tag, int q, int u) __attribute__((always_inline)) {
    constexpr int RS = decltype(rs_tag)::value, S = decltype(s_tag)::value;
    const int bo = u >> 1, ii = u & 1;
    const int j = w + 4 * (q >> 1);
    const int kp = 16 * S + 8 * ii + 4 * hh;                 // first k of the piece within the step
    const bool in = 32 * j + kp < G.K;
    const uint32_t off = in ? vo[bo] + (uint32_t)(32 * (2 * S + ii)) : vo[bo] - (uint32_t)(16 * hh);
    R[RS][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bo >= 2 ? srdb : srda, (int)off, 128 * j, 0));
  };
  auto load1_full = [&](auto rs_tag, auto s_tag, int q, int u) __attribute__((always_inline)) {
    constexpr int RS = decltype(rs_tag)::value, S = decltype(s_tag)::value;
    const int bo = u >> 1, ii = u & 1;
    const int j = w + 4 * (q >> 1);
    R[RS][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bo >= 2 ? srdb : srda, (int)(vo[bo] + (uint32_t)(32 * (2 * S + ii))), 128 * j, 0));
  };
  // split t (0..15: bo = t >> 2, ii = (t >> 1) & 1, pair t & 1) of the half-step held in ring slot RS -> fragment set FS
  auto split1 = [&](auto rs_tag, auto fs_tag, auto s_tag, auto full_tag, int q, int t) __attribute__((always_inline)) {
    constexpr int RS = decltype(rs_tag)::value, FS = decltype(fs_tag)::value, S = decltype(s_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    const int bo = t >> 2, ii = (t >> 1) & 1, pr = t & 1;
    f32x4 v = R[RS][2 * bo + ii];
    if (!FULL) {
      const int j = w + 4 * (q >> 1);
      if (32 * j + 16 * S + 8 * ii + 4 * hh >= G.K) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    if (MGW_EXP & 1) F[FS][bo][0][2 * ii + pr] = F[FS][bo][1][2 * ii + pr] = F[FS][bo][2][2 * ii + pr] = __float_as_uint(pr ? v.z : v.x) ^ __float_as_uint(pr ? v.w : v.y);
    else split2(pr ? v.z : v.x, pr ? v.w : v.y, msk, F[FS][bo][0][2 * ii + pr], F[FS][bo][1][2 * ii + pr], F[FS][bo][2][2 * ii + pr]);
  };
  auto frag = [&](auto fs_tag, int bo, int plane) __attribute__((always_inline)) -> bf16x8 {
    constexpr int FS = decltype(fs_tag)::value;
    return __builtin_bit_cast(bf16x8, u32x4{F[FS][bo][plane][0], F[FS][bo][plane][1], F[FS][bo][plane][2], F[FS][bo][plane][3]});
  };
  // MFMA t (0..23) of a half-step on fragment set FS: block (bi, bj) = t / 6, product t % 6, small classes first.  The product is
  // formed transposed (first operand = the B rows): a lane ends with 4 x 4 consecutive output columns of one row.
  auto mfma1 = [&](auto fs_tag, int t) __attribute__((always_inline)) {
    const int blk = t / 6, p = t % 6, bi = blk >> 1, bj = blk & 1;
    const int pa = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;    // A plane: lo.hi(A hi) hi.lo mid.mid mid.hi hi.mid hi.hi
    const int pb = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;    //   listed as (B plane . A plane)
    const int cls = MGW_NACC == 3 ? (p < 3 ? 2 : p < 5 ? 1 : 0) : (p < 5 ? 1 : 0);
    if (MGW_EXP & 2) { acc[bi][bj][cls][t & 15] += __uint_as_float(F[decltype(fs_tag)::value][bi][pa][t & 3] ^ F[decltype(fs_tag)::value][2 + bj][pb][t & 3]); return; }
    acc[bi][bj][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(fs_tag, 2 + bj, pb), frag(fs_tag, bi, pa), acc[bi][bj][cls], 0, 0, 0);
  };

  // ---- prologue: half-steps 0 .. 3 requested, half-step 0 split
  if (Q > 0) {
#pragma unroll
    for (int u = 0; u < 8; u++) load1(IntC<0>{}, IntC<0>{}, 0, u);
#pragma unroll
    for (int u = 0; u < 8; u++) load1(IntC<1>{}, IntC<1>{}, 1, u);
    if (Q > 2) {
#pragma unroll
      for (int u = 0; u < 8; u++) load1(IntC<2>{}, IntC<0>{}, 2, u);
#pragma unroll
      for (int u = 0; u < 8; u++) load1(IntC<3>{}, IntC<1>{}, 3, u);
    }
#pragma unroll
    for (int t = 0; t < 16; t++) split1(IntC<0>{}, IntC<0>{}, IntC<0>{}, mg::BoolTag<false>{}, 0, t);
  }
  // ---- the K loop.  Half-step q (Q4 = q mod 4 is a literal: ring slot and fragment set q & 1 are register names): 24 MFMAs on
  // fragment set q & 1, each followed by a slot (sched_barrier pins the order):
  //   slots t with t mod 3 != 2 (16 of them)   one split (11 VALU) of half-step q + 1: ring slot (q + 1) mod 4 -> the other fragment set
  //   slots t with t mod 3 == 2 (8 of them)    one request of half-step q + 4 into ring slot q mod 4 (split during half-step q - 1)
  auto half = [&](auto q4_tag, auto full_tag, int q) __attribute__((always_inline)) {
    constexpr int Q4 = decltype(q4_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    using FS = IntC<Q4 & 1>;
    using FN = IntC<(Q4 & 1) ^ 1>;
    using RN = IntC<(Q4 + 1) % 4>;
    using RL = IntC<Q4>;
    using SN = IntC<(Q4 + 1) & 1>;     // s of half-step q + 1
    using SL = IntC<Q4 & 1>;           // s of half-step q + 4
    const bool real = FULL || q < Q, nxt = FULL || q + 1 < Q, nxt4 = FULL || q + 4 < Q;
#define MGW_STEP(t)                                                                                              \
    {                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      if (real) mfma1(FS{}, (t));                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      if ((t) % 3 != 2) { if (nxt) split1(RN{}, FN{}, SN{}, full_tag, q + 1, (t) - (t) / 3); }                    \
      else if (nxt4 && !(MGW_EXP & 8)) { if (FULL) load1_full(RL{}, SL{}, q + 4, (t) / 3); else load1(RL{}, SL{}, q + 4, (t) / 3); } \
    }
    MGW_STEP(0) MGW_STEP(1) MGW_STEP(2) MGW_STEP(3) MGW_STEP(4) MGW_STEP(5) MGW_STEP(6) MGW_STEP(7)
    MGW_STEP(8) MGW_STEP(9) MGW_STEP(10) MGW_STEP(11) MGW_STEP(12) MGW_STEP(13) MGW_STEP(14) MGW_STEP(15)
    MGW_STEP(16) MGW_STEP(17) MGW_STEP(18) MGW_STEP(19) MGW_STEP(20) MGW_STEP(21) MGW_STEP(22) MGW_STEP(23)
    __builtin_amdgcn_sched_barrier(0);
#undef MGW_STEP
  };
  {
    int q = 0;
    // FULL group of four: half-steps up to q + 3 + 4 exist and lie in whole steps (the step of half-step q + 7 is w + 4 ((q + 7) >> 1))
    for (; q + 7 < Q && 32 * (w + 4 * ((q + 7) >> 1) + 1) <= G.K; q += 4) {
      half(IntC<0>{}, mg::BoolTag<true>{}, q);
      half(IntC<1>{}, mg::BoolTag<true>{}, q + 1);
      half(IntC<2>{}, mg::BoolTag<true>{}, q + 2);
      half(IntC<3>{}, mg::BoolTag<true>{}, q + 3);
    }
    for (; q < Q; q += 4) {      // (the last group may run past Q: those half-steps do nothing)
      half(IntC<0>{}, mg::BoolTag<false>{}, q);
      half(IntC<1>{}, mg::BoolTag<false>{}, q + 1);
      half(IntC<2>{}, mg::BoolTag<false>{}, q + 2);
      half(IntC<3>{}, mg::BoolTag<false>{}, q + 3);
    }
  }

  // ---- the four waves' partial tiles through LDS: classes summed small -> large per wave, then wave 0 + 1 + 2 + 3 per block
  f32x4 *xch = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
    const int bi = blk >> 1, bj = blk & 1;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; i++)
        o[i] = MGW_NACC == 3 ? (acc[bi][bj][MGW_NACC - 1][4 * g + i] + acc[bi][bj][1][4 * g + i]) + acc[bi][bj][0][4 * g + i]
                             : acc[bi][bj][1][4 * g + i] + acc[bi][bj][0][4 * g + i];
      xch[((w * 4 + blk) * 4 + g) * 64 + lane] = o;
    }
  }
  // the epilogue's operands: wave w finishes block w
  const int wm = w >> 1, wn = w & 1;
  f32x4 ebias[4], egate[4];
  const int em = m0 + 32 * wm + r32, emc = em < G.M ? em : G.M - 1;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int n = n0 + 32 * wn + 8 * g + 4 * hh, nc = n < G.N ? n : 0;
    if (EPI == mg::EPI_BIAS_ACT) ebias[g] = *reinterpret_cast<const f32x4 *>(G.bias + nc);
    if (EPI == mg::EPI_GATE_COLSUM) egate[g] = *reinterpret_cast<const f32x4 *>(G.gate + (int64_t)emc * G.ldg + nc);
  }
  __syncthreads();
  const bool relu = G.act == 0;
  float sq = 0.0f;
  float *red = reinterpret_cast<float *>(lds) + (w * 4 + w) * 4 * 64 * 4;   // (wave w's own image of block w: read below, then free)
  f32x4 outv[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    f32x4 o = xch[((0 * 4 + w) * 4 + g) * 64 + lane];
#pragma unroll
    for (int ws = 1; ws < 4; ws++) {
      const f32x4 p = xch[((ws * 4 + w) * 4 + g) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] += p[i];
    }
    outv[g] = o;
  }
  (void)red;
  float *red2 = reinterpret_cast<float *>(lds);   // re-used after the barrier below (colsum / sqsum only)
  if ((EPI == mg::EPI_GATE_COLSUM && G.colsum != nullptr) || (EPI == mg::EPI_SQSUM && G.sqsum != nullptr)) __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int n = n0 + 32 * wn + 8 * g + 4 * hh;
    const bool ok = em < G.M && n < G.N;
    f32x4 o = outv[g];
    if (EPI == mg::EPI_BIAS_ACT) {
      const f32x4 bb = ebias[g];
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = fmaxf(o[i] + bb[i], 0.0f);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = tanhf(o[i] + bb[i]);
      }
    }
    if (EPI == mg::EPI_GATE_COLSUM) {
      const f32x4 gt = egate[g];
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = gt[i] > 0.0f ? o[i] : 0.0f;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = o[i] * (1.0f - gt[i] * gt[i]);
      }
      if (G.colsum != nullptr) {
        f32x4 cs;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float c = row16_sum(ok ? o[i] : 0.0f);
          c += __shfl_xor(c, 16, 64);
          cs[i] = c;
        }
        if (r32 == 0) *reinterpret_cast<f32x4 *>(red2 + wm * 64 + 32 * wn + 8 * g + 4 * hh) = cs;
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
    if (tid < 64 && n0 + tid < G.N) G.colsum[(int64_t)tm * G.N + n0 + tid] = red2[tid] + red2[64 + tid];
  }
  if (EPI == mg::EPI_SQSUM && G.sqsum != nullptr) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) red2[w] = sq;
    __syncthreads();
    if (tid == 0) G.sqsum[tm * tiles_n + tn] = (red2[0] + red2[1]) + (red2[2] + red2[3]);
  }
}

template <int EPI>
__global__ __launch_bounds__(THREADS) void k_gemm_x3w_nt(Args G) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  gemm_tile_nt<EPI>(G, lds, (int)blockIdx.x);
}

}  // namespace mgw
