// mlp_gemm_x3p_exp.hpp — EXPERIMENT (round 6, after brl_mlp_gemm_x3_group): the bf16x3 product with its operands ALREADY split into three
// bf16 planes in memory (hi / mid / lo, each the fp32 array's shape) — the producers (a layer's epilogue, the optimizer) would write the
// planes beside the fp32 values, the product then has no vector work at all: DMA -> LDS -> MFMA.  The question it answers: can a
// 1024 x 1024 x 1024 product (256 tiles of 64 x 64, one per CU: the PPO minibatch step's forward and input-gradient products, 19 - 21 us
// on the exact fp32 kernels) run in <= 13.5 us this way?  The in-kernel split could not (7.3 vector instructions per MFMA on that tile).
//   * 64 x 64 tile, 256 threads = 4 waves; every wave multiplies the WHOLE tile over one 16-deep K step of each 64-deep chunk (wave w:
//     k = 16 w .. 16 w + 15): every LDS byte is read by exactly one wave (12 fragments of 1 KB per 24 MFMAs); the four partial tiles are
//     added through LDS behind the loop, in wave order;
//   * staging as csrc/mlp_infer.hpp: global_load_lds 16 B per lane, 128-byte LDS rows, pieces XOR-swizzled on the source address, 3
//     stages of 48 KB (6 planes x 64 rows x 128 B), two chunks in flight across one barrier per chunk.
// Layout NT only (both operands K-contiguous), M, N, K multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace x3p {

#ifndef X3P_EXP
#define X3P_EXP 0   // timing experiments (wrong results): 1 = no DMA in the loop, 2 = no MFMA, 4 = no fragment reads, 8 = 4-byte DMA pieces
#endif
#ifndef X3P_LOADERS
#define X3P_LOADERS 0   // 1: 512 threads, waves 4..7 issue the DMA instructions (one loader wave per SIMD), waves 0..3 multiply
#endif


constexpr int BK = 64, STAGES = 3, THREADS = X3P_LOADERS ? 512 : 256;
constexpr int PLANE = 64 * 128;              // 8 KB
constexpr int STAGE_BYTES = 6 * PLANE;       // 48 KB
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const uint16_t *a[3];   // [M][lda] hi, mid, lo
  int64_t lda;
  const uint16_t *b[3];   // [N][ldb]
  int64_t ldb;
  float *c;               // [M][ldc]
  int64_t ldc;
  int M, N, K;
};

template <bool V>
struct BoolTag { static constexpr bool value = V; };

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_base,
                                   (X3P_EXP & 8) ? 4 : 16, 0, 0);
}


__global__ __launch_bounds__(THREADS) void k_gemm_x3p(Args G) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = G.N / 64, nblk = (int)gridDim.x, bid = (int)blockIdx.x;
  // XCD x (blocks b = x mod 8) owns a 4 x 8 block of tiles when the grid allows
  int tm, tn;
  if (nblk % 8 == 0 && tiles_n % 8 == 0 && (G.M / 64) % 4 == 0) {
    const int L = (bid % 8) * (nblk / 8) + bid / 8;
    const int blk = L >> 5, i = L & 31, bpr = tiles_n / 8;
    tm = 4 * (blk / bpr) + (i >> 3);
    tn = 8 * (blk % bpr) + (i & 7);
  } else {
    tm = bid / tiles_n;
    tn = bid - tm * tiles_n;
  }
  const int m0 = tm * 64, n0 = tn * 64;
  const int nchunks = G.K / BK;
  const bool loader = X3P_LOADERS && w >= 4;
  const int wl = X3P_LOADERS ? (w & 3) : w;      // the wave's share of the DMA instructions / its K step

  // ---- staging: a chunk = 48 DMA instructions of 1 KB (8 rows x 128 B); wave w issues, of every plane, rows 16 w .. 16 w + 15 (2)
  uint32_t off[4];   // [operand][jj]
#pragma unroll
  for (int o = 0; o < 2; o++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int row = 16 * wl + 8 * jj + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      off[2 * o + jj] = (uint32_t)(((int64_t)((o ? n0 : m0) + row) * (o ? G.ldb : G.lda) + 8 * c) * 2);
    }
  int kc = 0;
  auto stage_one = [&](unsigned char *st, int j) __attribute__((always_inline)) {   // j = 0..11: plane j >> 1, row group 2 w + (j & 1)
    const int pl = j >> 1, jj = j & 1;
    const char *base = reinterpret_cast<const char *>(pl < 3 ? G.a[pl] : G.b[pl - 3]);
    uint32_t o = off[2 * (pl >= 3) + jj] + (uint32_t)kc * (BK * 2);
    asm volatile("" : "+v"(o));
    glds16(base + o, st + pl * PLANE + (2 * wl + jj) * 1024);
  };

  // ---- fragments: lane (i, h); wave w reads K pieces 2 w + h of every row
  const int i = lane & 31, h = lane >> 5;
  const int fo = i * 128 + (((2 * wl + h) ^ ((i >> 1) & 7)) << 4);
  // fragment u: 0..5 = A block (u / 3) plane (u % 3); 6..11 = B block ((u - 6) / 3) plane
  auto read_frag = [&](const unsigned char *st, int u) __attribute__((always_inline)) -> bf16x8 {
    const int isb = u >= 6, v = isb ? u - 6 : u, blk = v / 3, pl = v % 3;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(st + (3 * isb + pl) * PLANE + blk * 4096 + fo));
  };
  // the order the MFMAs first need them: p = 0 (B lo . A hi): A hi 0, B lo 0, B lo 1, A hi 1; p = 1 (B hi . A lo); p = 2 (mid . mid)
  constexpr int RORDER[12] = {0, 8, 11, 3, 2, 6, 9, 5, 1, 7, 10, 4};

  f32x16 acc[4][2];   // [block bm * 2 + bn][class: 0 = hi.hi, 1 = the five smaller products]
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[b][c][e] = 0.0f;
  // MFMA t (0..23): product p = t >> 2 in the order lo.hi hi.lo mid.mid mid.hi hi.mid hi.hi (B plane . A plane), block t & 3
  auto mf = [&](const bf16x8 (&f)[12], int t) __attribute__((always_inline)) {
    const int p = t >> 2, blk = t & 3, bm = blk >> 1, bn = blk & 1;
    const int pa = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;
    const int pb = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;
    const int cls = p < 5 ? 1 : 0;
    if (!(X3P_EXP & 2)) acc[blk][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[6 + 3 * bn + pb], f[3 * bm + pa], acc[blk][cls], 0, 0, 0);
  };

  // ---- prologue: chunks 0 and 1 requested; chunk 2 when chunk 0 has landed
  const bool issuer = !X3P_LOADERS || loader;
  for (int c = 0; c < 2 && c < nchunks; c++) {
    if (issuer) {
#pragma unroll
      for (int j = 0; j < 12; j++) stage_one(lds + c * STAGE_BYTES, j);
    }
    kc++;
  }
  if (nchunks >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (nchunks >= 3) {
    if (issuer) {
#pragma unroll
      for (int j = 0; j < 12; j++) stage_one(lds + 2 * STAGE_BYTES, j);
    }
    kc++;
  }
  if (loader) {
    // the loader's phases: the same waits and barriers as the multiplying waves', the DMA instructions of chunk c + 3 behind barrier c
    int stage = 0;
    for (int c = 0; c < nchunks; c++) {
      if (c + 1 < nchunks) {
        if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (c + 3 < nchunks && !(X3P_EXP & 1)) {
#pragma unroll
        for (int j = 0; j < 12; j++) stage_one(lds + stage * STAGE_BYTES, j);
        kc++;
      }
      stage = (stage + 1 == STAGES) ? 0 : stage + 1;
    }
    __syncthreads();
    __syncthreads();
    return;
  }
  bf16x8 f0[12], f1[12];
#pragma unroll
  for (int u = 0; u < 12; u++) f0[u] = read_frag(lds, u);

  // ---- phase c: the 24 MFMAs of chunk c from registers; behind the first: wait for this wave's DMA pieces of chunk c + 1, barrier
  // (chunk c + 1 has landed for everybody, nobody reads chunk c's stage any more); then one DMA instruction of chunk c + 3 (into chunk
  // c's stage) per gap and, later, one fragment read of chunk c + 1 per gap
  auto phase = [&](auto full_tag, const bf16x8 (&fu)[12], bf16x8 (&fn)[12], int c, int stage) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool next = FULL || c + 1 < nchunks;
    const bool dma = ((X3P_EXP & 1) || X3P_LOADERS) ? false : (FULL || c + 3 < nchunks);
    unsigned char *st = lds + stage * STAGE_BYTES;
    const unsigned char *sn = lds + ((stage + 1 == STAGES) ? 0 : stage + 1) * STAGE_BYTES;
    __builtin_amdgcn_sched_barrier(0);
    mf(fu, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (next) {
      if (FULL || c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#define RI(t) ((t) < 11 ? 0 : (t) > 22 ? 11 : (t) - 11)
#define X3P_GAP(t)                                                                  \
    mf(fu, t);                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if ((t) >= 1 && (t) <= 12 && dma) stage_one(st, (t) - 1);                       \
    if ((t) >= 11 && (t) <= 22 && next && !(X3P_EXP & 4)) fn[RORDER[RI(t)]] = read_frag(sn, RORDER[RI(t)]); \
    __builtin_amdgcn_sched_barrier(0);
    X3P_GAP(1) X3P_GAP(2) X3P_GAP(3) X3P_GAP(4) X3P_GAP(5) X3P_GAP(6) X3P_GAP(7) X3P_GAP(8) X3P_GAP(9) X3P_GAP(10) X3P_GAP(11) X3P_GAP(12)
    X3P_GAP(13) X3P_GAP(14) X3P_GAP(15) X3P_GAP(16) X3P_GAP(17) X3P_GAP(18) X3P_GAP(19) X3P_GAP(20) X3P_GAP(21) X3P_GAP(22) X3P_GAP(23)
#undef X3P_GAP
#undef RI
    if (dma) kc++;
  };
  {
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    const int nfull = nchunks - 3;
    int c = 0, stage = 0;
    auto nxt = [&]() { stage = (stage + 1 == STAGES) ? 0 : stage + 1; };
    for (; c + 1 < nfull; c += 2) {
      phase(T{}, f0, f1, c, stage); nxt();
      phase(T{}, f1, f0, c + 1, stage); nxt();
    }
    for (; c + 1 < nchunks; c += 2) {
      phase(F{}, f0, f1, c, stage); nxt();
      phase(F{}, f1, f0, c + 1, stage); nxt();
    }
    if (c < nchunks) phase(F{}, f0, f1, c, stage);
  }

  // ---- the four waves' partial tiles through LDS ([wave][block][register group][lane] float4), wave w finishes block w
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = acc[b][1][4 * g + e] + acc[b][0][4 * g + e];
      *reinterpret_cast<f32x4 *>(lds + (((w * 4 + b) * 4 + g) * 64 + lane) * 16) = v;
    }
  __syncthreads();
  const int bm = w >> 1, bn = w & 1;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    f32x4 o = *reinterpret_cast<const f32x4 *>(lds + (((0 * 4 + w) * 4 + g) * 64 + lane) * 16);
#pragma unroll
    for (int ww = 1; ww < 4; ww++) {
      const f32x4 p = *reinterpret_cast<const f32x4 *>(lds + (((ww * 4 + w) * 4 + g) * 64 + lane) * 16);
      o += p;
    }
    *reinterpret_cast<f32x4 *>(G.c + (int64_t)(m0 + 32 * bm + i) * G.ldc + n0 + 32 * bn + 8 * g + 4 * h) = o;
  }
}

// fp32 [rows][ld] -> three bf16 planes of the same shape (truncation splits: x = hi + mid + lo exactly)
__global__ void k_split_planes(const float *x, uint16_t *hi, uint16_t *mid, uint16_t *lo, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float v = x[idx];
  const unsigned u = __float_as_uint(v);
  const float r = v - __uint_as_float(u & 0xffff0000u);
  const unsigned ur = __float_as_uint(r);
  const float l = r - __uint_as_float(ur & 0xffff0000u);
  hi[idx] = (uint16_t)(u >> 16);
  mid[idx] = (uint16_t)(ur >> 16);
  lo[idx] = (uint16_t)(__float_as_uint(l) >> 16);
}

}  // namespace x3p
