// mlp_linear_x3p.hpp — one layer of the policy MLP for LARGE-batch fp32 inference (the policy rollout's 8192-row forwards, the
// evaluators': src/roll_out.py:49-108, src/evaluation.py; src/models.py:23-33: hk.Linear + relu) as a bf16x3 product (mlp_gemm_x3.hpp: a
// float32 is exactly hi + mid + lo in bf16; six bf16 MFMA products carry everything above 2^-24: fp32-grade, measured below the exact
// fp32 kernel's error) whose operands are ALREADY split into planes in memory:
//     y[M, N] = act(x[M, K] W[N, K]^T + b[N]),   x, W as three bf16 planes each, y as fp32 and / or as three bf16 planes
// mlp_gemm_x3.hpp splits in registers while staging — every element of x once per column tile (8 times at N = 1024), every element of W
// once per row tile (64 times at M = 8192), 3.7 vector instructions per MFMA.  In inference the weights are constant over a rollout's 128
// forwards (split ONCE per refresh: brl_split_planes) and an activation is split by the epilogue that produces it — once —, so the K loop
// here has no vector work at all: DMA -> LDS -> MFMA, the skeleton of mlp_infer.hpp (16-bit inference) with mlp_gemm_x3.hpp's products.
//   * 128 x 128 tile, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 64 x 32 = two 32 x 32 blocks, two accumulators per block (hi.hi |
//     the five smaller products), v_mfma_f32_32x32x16_bf16, the product formed transposed (a lane ends with 4 consecutive columns);
//   * 32-deep K chunks: six plane tiles of 128 rows x 64 B = 48 KB per stage, 3 stages, global_load_lds 16 B per lane (one instruction =
//     16 rows), the 16-byte pieces XOR-swizzled on the SOURCE address ((row >> 2) & 3): every ds_read_b128 of 32 rows is conflict-free;
//     two chunks in flight across ONE raw barrier per chunk (counted vmcnt); the fragments of chunk c + 1 are read while chunk c is
//     multiplied (two register sets); per MFMA gap at most one DMA instruction and one fragment read;
//   * NPX = 1: x is exact in bf16 (the 0/1 observation, written as bf16 by the step kernel: brl_macro_ext.obs_cast) — one plane, three
//     products (W lo, mid, hi times x);
//   * epilogue: + bias, ReLU; fp32 rows straight from the registers (the last hidden layer: the heads' library product reads them) and /
//     or the three planes through LDS (272-byte rows), stored 256 B per row at a time.
// N % 128 == 0, K % 32 == 0 (the callers fall back to brl_mlp_gemm_x3 otherwise); any M (edge rows are clamped on load, masked on store).
// Included by brl_mlp_gemm_x3.hip.
#pragma once

#include "mlp_gemm.hpp"

namespace lx3 {

using mg::f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32, STAGES = 3, THREADS = 512;
constexpr int PLANE = 128 * 64;              // one plane's tile of a chunk: 128 rows x 64 B = 8 KB
constexpr int STAGE_BYTES = 6 * PLANE;       // 48 KB
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
constexpr int C_ROW_BYTES = 128 * 2 + 16;    // an output plane's tile in LDS: 272-byte rows
static_assert(3 * 128 * C_ROW_BYTES <= LDS_BYTES, "the output planes reuse the stages");

struct Args {
  const uint16_t *x;      // planes of x: [NPX][M][ldx] (plane p at x + p * sx), K contiguous
  int64_t ldx, sx;
  const uint16_t *w;      // planes of W: [3][N][ldw] (nn.Linear's own layout), plane p at w + p * sw
  int64_t ldw, sw;
  const float *bias;      // [N]
  float *y;               // [M][ldy] fp32, or NULL
  int64_t ldy;
  uint16_t *yp;           // planes of y: [3][M][ldyp] (plane p at yp + p * syp), or NULL
  int64_t ldyp, syp;
  int M, N, K;
  int relu;
#ifdef LX3_TIMING
  unsigned long long *dbg;   // [workgroups][4]: s_memtime (shader clock) and s_memrealtime (100 MHz) at the start and the end of the K loop
#endif
};
#ifdef LX3_TIMING
#define LX3_STAMP(k) do { if (threadIdx.x == 0 && G.dbg) { G.dbg[(size_t)blockIdx.x * 4 + 2 * (k)] = __builtin_amdgcn_s_memtime(); \
                                                          G.dbg[(size_t)blockIdx.x * 4 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define LX3_STAMP(k) do { } while (0)
#endif

template <bool V>
struct BoolTag { static constexpr bool value = V; };

#ifndef LX3_STORE
#define LX3_STORE 0   // the planes' stores: 0 plain, 1 non-temporal, 2 write-through (sc0 sc1)
#endif
#ifndef LX3_EXP
#define LX3_EXP 0   // timing experiments (wrong results; scripts/micro/lx3_exp.hip): 1 = no DMA in the loop, 2 = no MFMA, 4 = no fragment reads in the loop, 8 = no barrier
#endif

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_base, 16,
                                   0, 0);
}

// x = hi + mid + lo exactly (truncation splits); the three bf16 bit patterns
__device__ __forceinline__ void split3(const float x, unsigned &hi, unsigned &mid, unsigned &lo) {
  const unsigned u = __float_as_uint(x);
  const float r = x - __uint_as_float(u & 0xffff0000u);
  const unsigned ur = __float_as_uint(r);
  const float l = r - __uint_as_float(ur & 0xffff0000u);
  hi = u >> 16;
  mid = ur >> 16;
  lo = __float_as_uint(l) >> 16;
}

template <int NPX>
__global__ __launch_bounds__(THREADS) void k_linear_x3p(Args G) {
  static_assert(NPX == 1 || NPX == 3, "x: one plane (exact in bf16) or three");
  constexpr int NI = NPX + 3;                  // DMA instructions per wave and chunk: one per plane (its 16 rows)
  constexpr int NP = NPX == 3 ? 6 : 3;         // products per block and K step
  constexpr int NMF = 4 * NP;                  // MFMAs per chunk and wave: 2 K steps x 2 blocks x NP
  constexpr int NFR = 2 * NPX + 3;             // fragments per K step: 2 x blocks x their planes, the W block's three
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (G.M + 127) / 128, tiles_n = G.N / 128, nblk = (int)gridDim.x, bid = (int)blockIdx.x;
  // workgroup -> tile: blocks b and b + 8 share an XCD; consecutive logical ids = one XCD = 2 x 4 blocks of tiles.  Speed only.
  const int L = (nblk % 8 == 0) ? (bid % 8) * (nblk / 8) + bid / 8 : bid;
  int tm, tn;
  if (tiles_m % 2 == 0 && tiles_n % 4 == 0) {
    const int blk = L >> 3, i = L & 7;
    tm = 2 * (blk / (tiles_n / 4)) + (i >> 2);
    tn = 4 * (blk % (tiles_n / 4)) + (i & 3);
  } else {
    tm = L / tiles_n;
    tn = L - tm * tiles_n;
  }
  const int m0 = tm * 128, n0 = tn * 128;
  const int nchunks = G.K / BK;

  // ---- staging: instruction j of a chunk (j < NPX: plane j of x, else plane j - NPX of W), this wave's rows 16 w .. 16 w + 15: lane ->
  // row 16 w + (lane >> 2), LDS slot lane & 3 <- the row's logical 16-byte piece (lane & 3) ^ ((row >> 2) & 3)
  uint32_t offx, offw;
  {
    const int row = 16 * w + (lane >> 2), pc = (lane & 3) ^ ((row >> 2) & 3);
    const int mr = (m0 + row < G.M) ? m0 + row : G.M - 1;
    offx = (uint32_t)(((int64_t)mr * G.ldx + 8 * pc) * 2);
    offw = (uint32_t)(((int64_t)(n0 + row) * G.ldw + 8 * pc) * 2);
  }
  int kc = 0;
  auto stage_one = [&](unsigned char *st, int j) __attribute__((always_inline)) {
    const bool isw = j >= NPX;
    const char *base = reinterpret_cast<const char *>(isw ? G.w + (int64_t)(j - NPX) * G.sw : G.x + (int64_t)j * G.sx);
    uint32_t o = (isw ? offw : offx) + (uint32_t)kc * (BK * 2);
    asm volatile("" : "+v"(o));
    glds16(base + o, st + j * PLANE + w * 1024);
  };

  // ---- fragments: lane (r, hh): K step s = the 16-byte piece 2 s + hh of operand row r
  const int wm = w >> 2, wn = w & 3, r32 = lane & 31, hh = lane >> 5;
  const int sw4 = (r32 >> 2) & 3;
  const int fa0 = (64 * wm + r32) * 64 + ((hh ^ sw4) << 4), fb0 = NPX * PLANE + (32 * wn + r32) * 64 + ((hh ^ sw4) << 4);
  // fragment u of a step: u < 2 NPX: x block u / NPX, plane u % NPX; else W plane u - 2 NPX
  auto read_frag = [&](const unsigned char *st, int s, int u) __attribute__((always_inline)) -> bf16x8 {
    const int off = (u < 2 * NPX) ? (u % NPX) * PLANE + (u / NPX) * 2048 + fa0 : (u - 2 * NPX) * PLANE + fb0;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(st + (off ^ (s << 5))));
  };

  f32x16 acc[2][2];     // [x block][class: 0 = hi.hi, 1 = the smaller products]
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[b][c][e] = 0.0f;
  // MFMA t of a chunk: K step t / (2 NP), then product-major, block = t & 1 (consecutive MFMAs alternate between the blocks' accumulators);
  // products in the order small -> large: NPX = 3: lo.hi hi.lo mid.mid mid.hi hi.mid hi.hi (W plane . x plane); NPX = 1: lo mid hi (W) . x
  auto mf = [&](const bf16x8 (&f)[2][NFR], int t) __attribute__((always_inline)) {
    const int s = t / (2 * NP), u = t % (2 * NP), p = u >> 1, b = u & 1;
    int px, pw;
    if (NPX == 3) {
      px = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;
      pw = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;
    } else {
      px = 0;
      pw = 2 - p;
    }
    const int cls = (px == 0 && pw == 0) ? 0 : 1;
    if (!(LX3_EXP & 2)) acc[b][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s][2 * NPX + pw], f[s][b * NPX + px], acc[b][cls], 0, 0, 0);
  };
  // the order in which a chunk's MFMAs first need its fragments
  auto rorder = [&](int q, int &s, int &u) __attribute__((always_inline)) {
    s = q / NFR;
    const int i = q % NFR;
    if (NPX == 3) {
      constexpr int O[9] = {8, 0, 3, 6, 2, 5, 7, 1, 4};     // W lo, x hi 0 / 1; W hi, x lo 0 / 1; W mid, x mid 0 / 1
      u = O[i];
    } else {
      constexpr int O[5] = {4, 0, 1, 3, 2};                // W lo, x 0 / 1; W mid; W hi
      u = O[i];
    }
  };

  // ---- prologue: chunks 0 and 1 requested, chunk 2 when chunk 0 has landed (the first wait then shares the memory system with one chunk)
  for (int c = 0; c < 2 && c < nchunks; c++) {
#pragma unroll
    for (int j = 0; j < NI; j++) stage_one(lds + c * STAGE_BYTES, j);
    kc++;
  }
  if (nchunks >= 2) {
    if (NI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (nchunks >= 3) {
#pragma unroll
    for (int j = 0; j < NI; j++) stage_one(lds + 2 * STAGE_BYTES, j);
    kc++;
  }
  LX3_STAMP(0);
  bf16x8 f0[2][NFR], f1[2][NFR];
#pragma unroll
  for (int q = 0; q < 2 * NFR; q++) {
    int s, u;
    rorder(q, s, u);
    f0[s][u] = read_frag(lds, s, u);
  }

  // ---- phase c: the chunk's MFMAs from registers; behind the first: this wave's DMA pieces of chunk c + 1 have landed, barrier (they
  // have for everybody; nobody reads chunk c's stage any more: those reads were issued in phase c - 1); then one DMA instruction of
  // chunk c + 3 (into chunk c's stage) and one fragment read of chunk c + 1 per gap.  In flight across the barrier: chunk c + 2.
  auto phase = [&](auto full_tag, const bf16x8 (&fu)[2][NFR], bf16x8 (&fn)[2][NFR], int c, int stage) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool next = FULL || c + 1 < nchunks;
    const bool dma = (LX3_EXP & 1) ? false : (FULL || c + 3 < nchunks);
    unsigned char *st = lds + stage * STAGE_BYTES;
    const unsigned char *sn = lds + ((stage + 1 == STAGES) ? 0 : stage + 1) * STAGE_BYTES;
    __builtin_amdgcn_sched_barrier(0);
    mf(fu, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (next) {
      if (FULL || c + 2 < nchunks) {
        if (NI == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if (!(LX3_EXP & 8)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < NMF; t++) {
      mf(fu, t);
      __builtin_amdgcn_sched_barrier(0);
      if (t - 1 < NI && dma) stage_one(st, t - 1);
      if (t - 1 < 2 * NFR && next && !(LX3_EXP & 4)) {
        int s, u;
        rorder(t - 1, s, u);
        fn[s][u] = read_frag(sn, s, u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dma) kc++;
  };
  static_assert(2 * NFR <= NMF - 1, "a fragment read per gap");
  {
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    const int nfull = nchunks - 3;     // phases c < nfull: chunk c + 3 exists
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

  LX3_STAMP(1);
  // ---- epilogue: lane holds, per block b, row m0 + 64 wm + 32 b + r32, columns n0 + 32 wn + 8 g + 4 hh + (0..3)
  __syncthreads();      // every wave is done with the stages (its last fragments are in registers; no DMA is in flight)
  const float floor_v = G.relu ? 0.0f : -__builtin_inff();
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const int row = 64 * wm + 32 * b + r32, em = m0 + row;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int col = 32 * wn + 8 * g + 4 * hh;
      const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(G.bias + n0 + col);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = fmaxf((acc[b][1][4 * g + e] + acc[b][0][4 * g + e]) + bias4[e], floor_v);
      if (G.y != nullptr && em < G.M) *reinterpret_cast<f32x4 *>(G.y + (int64_t)em * G.ldy + n0 + col) = o;
      if (G.yp != nullptr) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; e++) split3(o[e], h[e], m[e], l[e]);
        unsigned char *p = lds + row * C_ROW_BYTES + col * 2;
        *reinterpret_cast<u32x2 *>(p) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2 *>(p + 128 * C_ROW_BYTES) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *reinterpret_cast<u32x2 *>(p + 2 * 128 * C_ROW_BYTES) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
      }
    }
  }
  if (G.yp != nullptr) {
    __syncthreads();
    // 3 planes x 128 rows x 16 pieces of 16 B: a wave instruction stores four whole 256-byte rows
#pragma unroll
    for (int it = 0; it < (3 * 128 * 16) / THREADS; it++) {
      const int idx = it * THREADS + tid, pl = idx >> 11, row = (idx >> 4) & 127, ch = idx & 15;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + (pl * 128 + row) * C_ROW_BYTES + ch * 16);
      if (m0 + row < G.M) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(G.yp + (int64_t)pl * G.syp + (int64_t)(m0 + row) * G.ldyp + n0 + ch * 8);
#if LX3_STORE == 1
        __builtin_nontemporal_store(v, dst);
#elif LX3_STORE == 2
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#else
        *dst = v;
#endif
      }
    }
  }
}

// fp32 [n] -> three bf16 planes (plane p at planes + p * stride): the weights, once per refresh
__global__ __launch_bounds__(256) void k_split_planes(const float *x, uint16_t *planes, int64_t stride, int64_t n4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4 *>(x)[idx];
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; e++) split3(v[e], h[e], m[e], l[e]);
  uint16_t *p = planes + 4 * idx;
  *reinterpret_cast<u32x2 *>(p) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
  *reinterpret_cast<u32x2 *>(p + stride) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
  *reinterpret_cast<u32x2 *>(p + 2 * stride) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
}

}  // namespace lx3
