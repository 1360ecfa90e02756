// mlp_gemm_x3.hpp — fp32 products on the bf16 matrix pipe at fp32-grade error ("bf16x3"): the forward layers of LARGE batches (the
// policy rollout's 8192-row forwards, the evaluators': src/roll_out.py:49-108, src/evaluation.py) — opt-in (config["inference_gemm"]).
//
//   * a float32 x is EXACTLY hi + mid + lo with three bf16 pieces (truncation splits: hi = top 16 bits of x, mid = top 16 bits of
//     x - hi, lo = x - hi - mid: at most 8 significant bits are left); of the nine cross-products the six with weight >= 2^-16
//     (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi) carry everything above 2^-24 relative, each exact in the fp32 accumulator;
//     v_mfma_f32_32x32x16_bf16 runs at 16 x the f32-input MFMA's rate.  Two accumulators per output block (hi.hi | the five smaller
//     products), added once after the K loop: the large class takes K / 16 roundings where the exact kernel's fma chain takes K —
//     measured max |err| against float64 0.07 - 0.44 x the exact kernel's (scripts/micro/gemm_x3_test.hip, profiles/r06).
//   * the register split costs 5.5 vector instructions per operand element, a vector instruction ~4.2 cycles of a SIMD's issue, and
//     beside a bf16 MFMA (32 cycles, 8 of them holding the vector issue) at most ~5 of them hide.  A 64 x 64 tile splits
//     (64 + 64) x 32 elements per 12 MFMAs of a wave — 7.3 per MFMA, never hidden: five structures on that tile ran 21 - 28 us per
//     1024^3 product against the exact kernel's 19 (profiles/r06/r06_experiments.txt section 1).  The ratio is set by the TILE:
//     128 x 128 splits (128 + 128) x 32 elements per 24 MFMAs of each of eight waves = 3.7 per MFMA, and halves the LDS bytes and
//     the operand stream per MFMA as well: 8192 x 1024 x 1024 in 88 - 90 us against the exact kernel's 125 (and the library's 129).
//   * outputs of fewer than 256 such tiles (the PPO minibatch step's 1024 x 1024 products: 64) need K divided among SPLITK workgroups
//     per tile; the partial tiles then cross memory once more (write-through slabs, a ticket per tile, the last arriver adds them in
//     slice order: deterministic, nobody spins) and that costs what the tile saved: 26.8 us at 1024^3.  Kept (and tested) for
//     completeness; the step's products stay on the exact kernel.
#pragma once

#include "mlp_gemm.hpp"

namespace mgs {

using mg::BoolTag;
using mg::f32x4;
using mg::IntTag;
using mg::row16_sum;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32, THREADS = 512;
constexpr int PLANE = 128 * BK * 2;           // bytes of one bf16 plane tile (128 rows x 32 k, or 32 k x 128 columns) = 8 KB
constexpr int OFF_B = 3 * PLANE, STAGE = 6 * PLANE;
constexpr int LDS_BYTES = 2 * STAGE;          // 96 KB
constexpr int SLAB_FLOATS = 128 * 128;

#ifndef MGS_R1
#define MGS_R1 6      // first slot of the step-1 fragment reads (B, then A block 0: three slots each)
#endif
#ifndef MGS_BAR
#define MGS_BAR 20    // the barrier's slot in a phase of 24 (the staging ends before it, the next chunk's first fragments are read behind it)
#endif
#ifndef MGS_EXP
#define MGS_EXP 0      // timing experiments (wrong results): 1 = no split arithmetic, 2 = no MFMA
#endif

struct Args {
  mg::Args g;
  float *slabs;        // [tiles][splitk][128 * 128] partial tiles (splitk > 1)
  unsigned *tickets;   // [tiles] arrival counters, zero before the first launch: the last arriver of a tile puts its counter back to zero
  int splitk;
};

template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_tile(const Args &X, unsigned char *lds, int bid, int nblk) {
  const mg::Args &G = X.g;
  constexpr int NP = 4;                                               // 16-byte fp32 pieces per thread and chunk: 2 of A, 2 of B
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (G.M + 127) / 128, tiles_n = (G.N + 127) / 128, SK = X.splitk;
  if (bid >= tiles_m * tiles_n * SK) return;
  // workgroup -> (tile, K slice): consecutive logical ids (= one XCD: blocks b and b + 8 share one) are the SK slices of a tile, then
  // a 2 x 4 block of tiles (1 MB of A rows + 2 MB of B rows of a 1024^3 product in the XCD's 4 MB L2).  Speed only.
  const int L = (nblk % 8 == 0) ? (bid % 8) * (nblk / 8) + bid / 8 : bid;
  const int tile = L / SK, ks = L - tile * SK;
  int tm, tn;
  if (tiles_m % 2 == 0 && tiles_n % 4 == 0) {
    const int blk = tile >> 3, i = tile & 7;
    tm = 2 * (blk / (tiles_n / 4)) + (i >> 2);
    tn = 4 * (blk % (tiles_n / 4)) + (i & 3);
  } else {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  }
  const int m0 = tm * 128, n0 = tn * 128;
  const int nchunks = G.K / BK;                                 // (K % 32 == 0: the host checks)
  const int cps = (nchunks + SK - 1) / SK;                      // chunks per slice
  const int c0 = ks * cps, c1 = (c0 + cps < nchunks) ? c0 + cps : nchunks;
  const int ni = c1 > c0 ? c1 - c0 : 0;                         // this workgroup's chunks: c0 + i

  // ---- staging.  KC operand, 128 rows x 32 k = 1024 loads of 4 k: item q = tid + 512 jj -> row = q >> 3, P = (q >> 1) & 3,
  //   half = q & 1: k = 16 half + 4 P.   MC operand, 32 k rows x 128 columns: item q -> k row q >> 5, columns 4 (q & 31) ..
  uint32_t go[NP];
  int lw[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const bool isB = j >= 2;
    const int q = tid + 512 * (j & 1);
    const int x0 = isB ? n0 : m0, Xn = isB ? G.N : G.M;
    const int64_t ld = isB ? G.ldb : G.lda;
    const int base = isB ? OFF_B : 0;
    if (isB ? B_KC : A_KC) {
      const int row = q >> 3, P = (q >> 1) & 3, half = q & 1;
      const int x = (x0 + row < Xn) ? x0 + row : Xn - 1;
      go[j] = (uint32_t)(((int64_t)x * ld + 16 * half + 4 * P) * 4);
      lw[j] = base + row * 64 + ((P ^ ((row >> 2) & 3)) << 4) + 8 * half;
    } else {
      const int kr = q >> 5, p = q & 31;
      const int col = (x0 + 4 * p < Xn) ? x0 + 4 * p : 0;
      go[j] = (uint32_t)(((int64_t)kr * ld + col) * 4);
      lw[j] = base + kr * 256 + ((p ^ ((kr & 3) << 3)) << 3);
    }
  }
  const uint32_t stepa = (uint32_t)((A_KC ? (int64_t)BK : (int64_t)BK * G.lda) * 4), stepb = (uint32_t)((B_KC ? (int64_t)BK : (int64_t)BK * G.ldb) * 4);
  const __amdgpu_buffer_rsrc_t srda = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.B), (short)0, 0x7FFFFFFF, 0x00020000);
  uint32_t soa = (uint32_t)c0 * stepa, sob = (uint32_t)c0 * stepb;
  unsigned msk;
  asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(msk));
#ifndef MGS_SETS
#define MGS_SETS 1     // register sets of the operand stream: 1 = one chunk in flight (a request made in phase i lands before phase i + 1 splits
#endif                 // it), 2 = two (chunk c lives in set c & 1: requested two phases before it is split) — measured equal (r06ap_sets.txt)
  f32x4 rg[MGS_SETS][NP];
  // Every phase is branch-free: the requests of the chunks BEHIND the slice's last one (two or three per slice) read that last chunk again — the
  // scalar offset stops advancing —, are split and stored like any other and never multiplied.
  int nreq = 0;         // chunks requested so far
  auto gload = [&](auto set_tag, int j) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    const bool isB = j >= 2;
    rg[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isB ? srdb : srda, (int)go[j], (int)(isB ? sob : soa), 0));
  };
  auto gadvance = [&]() __attribute__((always_inline)) {
    nreq++;
    if (nreq < ni) { soa += stepa; sob += stepb; }
  };
  // the split of one half piece in three parts, so that no MFMA gap carries more than ~5 vector instructions:
  //   part 0: hi pair, r = x - hi        part 1: mid pair, l = r - mid        part 2: lo pair; behind the second half the plane stores
  unsigned sh[3][2];   // [plane][half]
  float sr0, sr1;      // the residuals between the parts
  auto stage_part = [&](auto set_tag, int j, int half, int part, unsigned char *st) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    if (part == 0) {
      const f32x4 v = rg[S][j];
      const float x0 = half ? v.z : v.x, x1 = half ? v.w : v.y;
      const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
      sh[0][half] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
      if (MGS_EXP & 1) { sr0 = x0; sr1 = x1; }
      else { sr0 = x0 - __uint_as_float(u0 & msk); sr1 = x1 - __uint_as_float(u1 & msk); }
    } else if (part == 1) {
      const unsigned v0 = __float_as_uint(sr0), v1 = __float_as_uint(sr1);
      sh[1][half] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
      if (!(MGS_EXP & 1)) { sr0 = sr0 - __uint_as_float(v0 & msk); sr1 = sr1 - __uint_as_float(v1 & msk); }
    } else {
      sh[2][half] = __builtin_amdgcn_perm(__float_as_uint(sr1), __float_as_uint(sr0), 0x07060302u);
      if (half) {
        unsigned char *p = st + lw[j];
        *reinterpret_cast<u32x2 *>(p) = u32x2{sh[0][0], sh[0][1]};
        *reinterpret_cast<u32x2 *>(p + PLANE) = u32x2{sh[1][0], sh[1][1]};
        *reinterpret_cast<u32x2 *>(p + 2 * PLANE) = u32x2{sh[2][0], sh[2][1]};
      }
    }
  };

  // ---- fragments (lane (r = lane & 31, h = lane >> 5); step s takes piece P = 2 s + h: k = 4 P + (0..3), 16 + 4 P + (0..3))
  const int wm = w >> 2, wn = w & 3, r32 = lane & 31, hh = lane >> 5;
  int fa[2], fb;        // byte offsets inside a stage: A blocks 0 / 1 (rows / columns 64 wm + 32 blk ..), the B block (32 wn ..)
  {
    const int qq = (lane >> 2) & 3, xg = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int ra = wm * 64 + 32 * b + r32;
      fa[b] = A_KC ? ra * 64 + ((hh ^ ((ra >> 2) & 3)) << 4) : (4 * hh + qq) * 256 + ((((wm * 64 + 32 * b + xg) >> 2) ^ (qq << 3)) << 3);
    }
    const int rb = wn * 32 + r32;
    fb = OFF_B + (B_KC ? rb * 64 + ((hh ^ ((rb >> 2) & 3)) << 4) : (4 * hh + qq) * 256 + ((((wn * 32 + xg) >> 2) ^ (qq << 3)) << 3));
  }
  // fragment u of step s: u = 0..2 = A block 0 hi / mid / lo, 3..5 = A block 1, 6..8 = B
  auto read_frag = [&](const unsigned char *st, int s, int u) __attribute__((always_inline)) -> bf16x8 {
    const bool isB = u >= 6;
    const bool kc = isB ? B_KC : A_KC;
    const int base = isB ? fb : fa[u / 3], plane = (u % 3) * PLANE;
    if (kc) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(st + (base ^ (s << 5)) + plane));
    const unsigned char *p = st + base + plane + 2048 * s;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4096));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, s16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
  };

  f32x16 acc[2][2];     // [A block][class: 0 = hi.hi, 1 = the five smaller products]
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[b][c][e] = 0.0f;
  // MFMA t (0..11) of a step: block t / 6, product t % 6 in the order lo.hi hi.lo mid.mid mid.hi hi.mid hi.hi (B plane . A plane);
  // the product is formed transposed (first operand = the B rows): a lane ends with 4 x 4 consecutive output columns of one row
  auto mf = [&](const bf16x8 (&f)[9], int t) __attribute__((always_inline)) {
    const int b = t / 6, p = t % 6;
    const int pa = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;
    const int pb = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;
    const int cls = p < 5 ? 1 : 0;
    if (!(MGS_EXP & 2)) acc[b][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[6 + pb], f[3 * b + pa], acc[b][cls], 0, 0, 0);
  };

  MG_STAMP(0);
  // ---- prologue: chunk 0 -> stage 0, chunk 1 in flight
  if (ni > 0) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<0>{}, j);
    gadvance();
#pragma unroll
    for (int j = 0; j < NP; j++)
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int part = 0; part < 3; part++) stage_part(IntTag<0>{}, j, h, part, lds);
#pragma unroll
    for (int j = 0; j < NP; j++) gload(IntTag<(MGS_SETS == 2 ? 1 : 0)>{}, j);      // chunk 1
    gadvance();
    if (MGS_SETS == 2) {
#pragma unroll
      for (int j = 0; j < NP; j++) gload(IntTag<0>{}, j);                          // chunk 2
      gadvance();
    }
  }
  __syncthreads();
  bf16x8 f0[9], f1[9];
  if (ni > 0) {
#pragma unroll
    for (int u = 0; u < 9; u++) f0[u] = read_frag(lds, 0, u);
  }
  MG_STAMP(1);
  // ---- the K loop.  Phase i = the 24 MFMAs of chunk i (step 0: slots 0..11 on f0, step 1: 12..23 on f1), each followed by a slot:
  //   slots 6 .. 14     the step-1 fragments of this stage (one read each: B, A block 0, A block 1)
  //   slots 0 .. 19     chunk i + 1 split in registers: half piece h (0..7) as three parts at slots floor(2.5 h) + 0, 1, 2 (5, 5 and 1
  //                     vector instructions; the plane stores behind a piece's second half): <= 6 vector instructions per MFMA gap
  //   slots 4 9 14 19   the request of piece j of chunk i + 2 into the registers just split
  //   slot 20           every LDS operation of the wave done, barrier
  //   slots 21 .. 23    the step-0 fragments of chunk i + 1 (three reads each: B, A block 0, A block 1)
  // (tried: THREE stages, the barrier at slot 0 waiting for stores a phase old — 98.7 us against 88.4 for 8192 x 1024 x 1024, dropped)
  auto phase = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value, NS = PAR ^ 1;
    unsigned char *st = lds + PAR * STAGE, *sn = lds + NS * STAGE;
    constexpr bool real = true, nxt = true, nxt2 = true;
    // staging schedule: half piece h (0..7) starts at slot HB(h) = floor(h (BAR - 2) / 8) and takes three slots; BAR = the barrier's slot
    constexpr int BAR = MGS_BAR;
    auto slot = [&](int s) __attribute__((always_inline)) {
      // (reads placed late: step 0's A-block-0 fragments are dead after slot 5, its other fragments after slot 11 — the two sets never
      //  live whole side by side: 256 registers per wave at two waves per SIMD)
      if (s >= MGS_R1 && s < MGS_R1 + 3 && real) f1[6 + (s - MGS_R1)] = read_frag(st, 1, 6 + (s - MGS_R1));           // B hi / mid / lo
      if (s >= MGS_R1 + 3 && s < MGS_R1 + 6 && real) f1[s - MGS_R1 - 3] = read_frag(st, 1, s - MGS_R1 - 3);         // A block 0
      if (s >= 12 && s < 15 && real) f1[3 + (s - 12)] = read_frag(st, 1, 3 + (s - 12));   // A block 1
      if (s < BAR && nxt) {
#pragma unroll
        for (int h = 0; h < 8; h++) {
          const int b = (h * (BAR - 2)) / 8;
#pragma unroll
          for (int part = 0; part < 3; part++)
            if (s == b + part) stage_part(IntTag<(MGS_SETS == 2 ? NS : 0)>{}, h >> 1, h & 1, part, sn);
        }
      }
#pragma unroll
      for (int j = 0; j < NP; j++) {
        if (s == ((2 * j + 1) * (BAR - 2)) / 8 + 2 && nxt2) {      // behind the last part that reads piece j's registers
          gload(IntTag<(MGS_SETS == 2 ? NS : 0)>{}, j);
          if (j == NP - 1) gadvance();
        }
      }
      if (s == BAR) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (s > BAR && nxt) {      // B first, then A block 0 — what the next phase's first MFMAs multiply —, A block 1 last
        constexpr int order[9] = {6, 7, 8, 0, 1, 2, 3, 4, 5};
        const int n = 23 - BAR, k = s - BAR - 1;
#pragma unroll
        for (int q = (9 * k) / n; q < (9 * (k + 1)) / n; q++) f0[order[q]] = read_frag(sn, 0, order[q]);
      }
    };
#define MGS_STEP(s)                                                                 \
    {                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                            \
      if (real) { if ((s) < 12) mf(f0, (s)); else mf(f1, (s) - 12); }                \
      __builtin_amdgcn_sched_barrier(0);                                            \
      slot(s);                                                                      \
    }
    MGS_STEP(0) MGS_STEP(1) MGS_STEP(2) MGS_STEP(3) MGS_STEP(4) MGS_STEP(5) MGS_STEP(6) MGS_STEP(7)
    MGS_STEP(8) MGS_STEP(9) MGS_STEP(10) MGS_STEP(11) MGS_STEP(12) MGS_STEP(13) MGS_STEP(14) MGS_STEP(15)
    MGS_STEP(16) MGS_STEP(17) MGS_STEP(18) MGS_STEP(19) MGS_STEP(20) MGS_STEP(21) MGS_STEP(22) MGS_STEP(23)
    __builtin_amdgcn_sched_barrier(0);
#undef MGS_STEP
  };
  {
    int i = 0;
    for (; i + 1 < ni; i += 2) {
      phase(IntTag<0>{});
      phase(IntTag<1>{});
    }
    if (i < ni) phase(IntTag<0>{});
  }
  MG_STAMP(2);

  // ---- this slice's partial tile: classes small -> large
  f32x16 v[2];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int e = 0; e < 16; e++) v[b][e] = acc[b][1][e] + acc[b][0][e];
  if (SK > 1) {
    // slab layout: [wave][block][register group][lane] float4 — a wave instruction stores 1 KB contiguous.  The stores are
    // WRITE-THROUGH (sc1) and the reducer's loads sc1: no release fence, no acquire — an agent-scope release writes back the XCD's
    // whole L2, and with thirty-two workgroups per XCD each doing so behind 64 KB of fresh stores the reduction took 16 us (first build)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(X.slabs, (short)0, 0x7FFFFFFF, 0x00020000);
    const uint32_t sbase = (uint32_t)(((int64_t)tile * SK + ks) * SLAB_FLOATS * 4);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int g = 0; g < 4; g++)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[b][4 * g], v[b][4 * g + 1], v[b][4 * g + 2], v[b][4 * g + 3]}), srs,
                                               (int)(sbase + (uint32_t)((((w * 2 + b) * 4 + g) * 64 + lane) * 16)), 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave, then the workgroup's barrier, then the ticket
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(lds);
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&X.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (old == (unsigned)(SK - 1)) ? 1u : 0u;
      if (old == (unsigned)(SK - 1)) __hip_atomic_store(&X.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the last arriver: ready for the next launch)
    }
    __syncthreads();
    if (*flag == 0u) return;
    // the last arriver adds the slices in slice order (its own comes from its slab too: one code path, one order)
    const uint32_t tbase = (uint32_t)((int64_t)tile * SK * SLAB_FLOATS * 4);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint32_t idx = (uint32_t)((((w * 2 + b) * 4 + g) * 64 + lane) * 16);
        f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(tbase + idx), 0, 16));
        for (int q = 1; q < SK; q++) {
          const f32x4 p = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(tbase + idx), q * SLAB_FLOATS * 4, 16));
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        v[b][4 * g] = o.x; v[b][4 * g + 1] = o.y; v[b][4 * g + 2] = o.z; v[b][4 * g + 3] = o.w;
      }
  }
  // ---- epilogue: lane holds, per block b, row m0 + 64 wm + 32 b + r32, columns n0 + 32 wn + 8 g + 4 hh + (0..3)
  const bool relu = G.act == 0;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int n = n0 + 32 * wn + 8 * g + 4 * hh, nc = n < G.N ? n : 0;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == mg::EPI_BIAS_ACT) bias4 = *reinterpret_cast<const f32x4 *>(G.bias + nc);
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int em = m0 + 64 * wm + 32 * b + r32;
      const bool ok = em < G.M && n < G.N;
      f32x4 o = f32x4{v[b][4 * g], v[b][4 * g + 1], v[b][4 * g + 2], v[b][4 * g + 3]};
      if (EPI == mg::EPI_BIAS_ACT) {
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = fmaxf(o[i] + bias4[i], 0.0f);
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = tanhf(o[i] + bias4[i]);
        }
      }
      if (EPI == mg::EPI_GATE_COLSUM) {
        const f32x4 gt = *reinterpret_cast<const f32x4 *>(G.gate + (int64_t)(em < G.M ? em : G.M - 1) * G.ldg + nc);
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = gt[i] > 0.0f ? o[i] : 0.0f;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = o[i] * (1.0f - gt[i] * gt[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) cs[i] += ok ? o[i] : 0.0f;     // block 0 + block 1: the wave's 64 rows = one 64-row tile of the sums
      }
      if (ok) *reinterpret_cast<f32x4 *>(G.C + (int64_t)em * G.ldc + n) = o;
    }
    if (EPI == mg::EPI_GATE_COLSUM && G.colsum != nullptr) {
      // column sums per 64-row tile (mlp_gemm.hpp's layout [ceil(M / 64)][N]): the 16 lanes of a DPP row, the half's two rows — the
      // wave owns its 64 rows x 32 columns whole: no other wave adds to them
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float c = row16_sum(cs[i]);
        c += __shfl_xor(c, 16, 64);
        cs[i] = c;
      }
      if (r32 == 0 && n < G.N && 2 * tm + wm < (G.M + 63) / 64) *reinterpret_cast<f32x4 *>(G.colsum + (int64_t)(2 * tm + wm) * G.N + n) = cs;
    }
  }
  MG_STAMP(3);
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(THREADS) void k_gemm_x3s(Args X) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  gemm_tile<A_KC, B_KC, EPI>(X, lds, (int)blockIdx.x, (int)gridDim.x);
}

// Several products of one layout as ONE launch, one K slice each (the DeepMind step's four weight gradients dz_l^T h_{l-1}: three
// 1024 x 1024 outputs + one 1024 x 480 = 224 tiles of 128 x 128 — one per CU and no partial tiles to add): workgroup b works on tile
// b - first[p] of problem p.
constexpr int GROUP_MAX = 8;
struct GroupArgs {
  int n;
  int first[GROUP_MAX + 1];   // prefix sums of the problems' tile counts
  Args x[GROUP_MAX];
};
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS) void k_gemm_x3s_group(GroupArgs GA) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < GA.n && b >= GA.first[p + 1]) p++;
  gemm_tile<A_KC, B_KC, mg::EPI_NONE>(GA.x[p], lds, b - GA.first[p], GA.first[p + 1] - GA.first[p]);
}

// slices per tile: enough workgroups for the chip (256 CUs), at least two chunks per slice
static inline int pick_splitk(int64_t m, int64_t n, int64_t k) {
  const int64_t tiles = ((m + 127) / 128) * ((n + 127) / 128);
  int sk = 1;
  while (sk < 8 && tiles * sk < 256 && k / 32 / (2 * sk) >= 2) sk *= 2;
  return sk;
}

}  // namespace mgs
