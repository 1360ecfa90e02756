// mlp_gemm_x3p.hpp — the bf16x3 product (csrc/mlp_gemm_x3.hpp: fp32 = hi + mid + lo in bf16, six bf16 MFMA products, fp32-grade error) with
// its operands ALREADY split into three bf16 planes in memory: the PPO minibatch step's 1024 x 1024 x 1024 products (src/update.py:74-242:
// the forward layers h_l = act(h_{l-1} W_l^T + b_l), the input gradients dz_{l-1} = (dz_l W_l) * act'(h_{l-1})), whose 64 tiles of
// 128 x 128 cannot fill 256 CUs and whose 64 x 64 tiles cannot carry the split (7.3 vector instructions per MFMA: mlp_gemm_x3.hpp).  The
// PRODUCERS split instead — a product's own epilogue writes the planes of its output beside the fp32 values (4 + 6 bytes per element, 11
// vector instructions per 2 elements, once), the optimizer writes the planes of the weights — and the product itself has no vector work:
// DMA -> LDS -> MFMA.
//   * 64 x 64 tile; 512 threads = 8 waves: waves 0..3 multiply — each the WHOLE tile over ONE 16-deep K step of every 64-deep chunk
//     (wave w: k = 16 w .. 16 w + 15), so that every LDS byte is read by exactly one wave: 12 fragment reads per 24 MFMAs — and waves 4..7
//     (one per SIMD, beside a multiplying wave) issue the DMA instructions: a global_load_lds holds its wave's issue for ~60 cycles,
//     twelve of them per chunk are 720 of the chunk's 768 MFMA cycles (measured: 16.4 -> 15.4 us against the multiplying waves issuing
//     them; scripts/micro/x3p_test.hip, profiles/r06/r06_experiments.txt section 8);
//   * staging as csrc/mlp_infer.hpp: 16 bytes per lane straight into LDS, 128-byte LDS rows, the 16-byte pieces XOR-swizzled on the SOURCE
//     address, 3 stages of 48 KB (6 planes x 64 rows x 128 B), two chunks in flight across ONE barrier per chunk;
//     K-contiguous operand (A always; B of layout NT): rows = m / n, ds_read_b128 fragments; the other (B of layout NN, [K][N]): rows = k,
//     two ds_read_b64_tr_b16 per fragment;
//   * behind the loop the four partial tiles are added through LDS in wave order (deterministic) and all eight waves run the epilogue:
//     + bias, activation | * act'(gate) and the column sums per 64-row tile (the bias gradient's partials), fp32 store and, optionally, the
//     three planes of what was stored;
//   * the limit: 48 KB per chunk through the CU's 64 B / clock vector memory path = the chunk's 768 MFMA cycles — both pipes full at this
//     tile; 15.2 - 16.4 us per 1024^3 product against the exact kernels' 19 - 21.
// M, N, K multiples of 64 (the callers fall back to the exact kernels otherwise).  Included by brl_mlp_gemm_x3.hip.
#pragma once

#include "mlp_gemm.hpp"

namespace x3p {

using mg::f32x4;
using mg::row16_sum;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 64, STAGES = 3, THREADS = 512;
constexpr int PLANE = 64 * 128;              // one plane's tile of a chunk: 64 rows x 128 B
constexpr int STAGE_BYTES = 6 * PLANE;       // 48 KB: A hi / mid / lo, B hi / mid / lo
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;

struct Args {
  const uint16_t *a;      // planes of A: [3][M][lda] (plane p at a + p * sa), K contiguous
  int64_t lda, sa;
  const uint16_t *b;      // planes of B: layout NT [3][N][ldb] (K contiguous), layout NN [3][K][ldb] (N contiguous)
  int64_t ldb, sb;
  float *c;               // [M][ldc]
  int64_t ldc;
  uint16_t *cp;           // planes of the stored output [3][M][ldcp] (plane p at cp + p * scp), or NULL
  int64_t ldcp, scp;
  int M, N, K;
  int act;                // 0 = ReLU, 1 = tanh
  const float *bias;      // EPI_BIAS_ACT: [N]
  const float *gate;      // EPI_GATE_COLSUM: [M][ldg] = the layer's forward output
  int64_t ldg;
  float *colsum;          // EPI_GATE_COLSUM: [M / 64][N] column sums of the stored values per 64-row tile, or NULL
};

template <bool V>
struct BoolTag { static constexpr bool value = V; };

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
// four consecutive values -> 8 bytes of each plane
__device__ __forceinline__ void store_planes4(uint16_t *p, const int64_t stride, const f32x4 v) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; e++) split3(v[e], h[e], m[e], l[e]);
  *reinterpret_cast<u32x2 *>(p) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
  *reinterpret_cast<u32x2 *>(p + stride) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
  *reinterpret_cast<u32x2 *>(p + 2 * stride) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
}

template <bool B_KC, int EPI>
__global__ __launch_bounds__(THREADS) void k_gemm_x3p(Args G) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = G.N / 64, nblk = (int)gridDim.x, bid = (int)blockIdx.x;
  // XCD x (workgroups b = x mod 8) owns a 4 x 8 block of tiles where the grid allows (its L2 then holds 256 rows of A, 512 of B).  Speed only.
  int tm, tn;
  if (nblk % 256 == 0 && tiles_n % 8 == 0 && (G.M / 64) % 4 == 0) {
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
  const bool loader = w >= 4;
  const int wl = w & 3;      // a loader's share of the DMA instructions / a multiplying wave's K step

  if (loader) {
    // ---- a chunk = 48 DMA instructions of 1 KB (8 rows x 128 B); loader wl issues, of every plane, rows 16 wl .. 16 wl + 15 (2
    // instructions).  LDS slot (row, ps) holds the row's logical 16-byte piece ps ^ f(row); f = (row >> 1) & 7 where the rows are m / n
    // (K contiguous), ((row >> 1) & 1) << 2 where they are k (layout NN's B): the fragment reads below are then conflict-free.
    uint32_t offa[2], offb[2];
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int row = 16 * wl + 8 * jj + (lane >> 3);
      offa[jj] = (uint32_t)(((int64_t)(m0 + row) * G.lda + 8 * ((lane & 7) ^ ((row >> 1) & 7))) * 2);
      if (B_KC) offb[jj] = (uint32_t)(((int64_t)(n0 + row) * G.ldb + 8 * ((lane & 7) ^ ((row >> 1) & 7))) * 2);
      else offb[jj] = (uint32_t)(((int64_t)row * G.ldb + n0 + 8 * ((lane & 7) ^ (((row >> 1) & 1) << 2))) * 2);
    }
    const uint32_t stepa = BK * 2, stepb = B_KC ? (uint32_t)(BK * 2) : (uint32_t)((int64_t)BK * G.ldb * 2);
    int kc = 0;
    auto stage_chunk = [&](unsigned char *st) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 12; j++) {     // plane j >> 1 (0..2: A, 3..5: B), row group 2 wl + (j & 1)
        const int pl = j >> 1, jj = j & 1;
        const bool isb = pl >= 3;
        const char *base = reinterpret_cast<const char *>(isb ? G.b + (pl - 3) * G.sb : G.a + pl * G.sa);
        uint32_t o = (isb ? offb[jj] + (uint32_t)kc * stepb : offa[jj] + (uint32_t)kc * stepa);
        asm volatile("" : "+v"(o));
        glds16(base + o, st + pl * PLANE + (2 * wl + jj) * 1024);
      }
      kc++;
    };
    // chunks 0 and 1 requested, chunk 2 when chunk 0 has landed; then: the multiplying waves' waits and barriers, the DMA
    // instructions of chunk c + 3 (into chunk c's stage) behind barrier c
    for (int c = 0; c < 2 && c < nchunks; c++) stage_chunk(lds + c * STAGE_BYTES);
    if (nchunks >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunks >= 3) stage_chunk(lds + 2 * STAGE_BYTES);
    int stage = 0;
    for (int c = 0; c < nchunks; c++) {
      if (c + 1 < nchunks) {
        if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (c + 3 < nchunks) stage_chunk(lds + stage * STAGE_BYTES);
      stage = (stage + 1 == STAGES) ? 0 : stage + 1;
    }
  } else {
    // ---- fragments: lane (i, h) of a 32-row block holds k = 16 wl + 8 h + (0..7) of operand row i
    const int i = lane & 31, h = lane >> 5;
    const int fkc = i * 128 + (((2 * wl + h) ^ ((i >> 1) & 7)) << 4);
    // layout NN's B ([k][n] rows): ds_read_b64_tr_b16 r (0, 1): a lane addresses row 16 wl + 8 h + 4 r + qq, the 8 bytes of n = 32 bn +
    // 16 g1 + 4 q .. + 3, and receives k = 16 wl + 8 h + 4 r + (0..3) of n = 32 bn + i
    const int qq = (lane >> 2) & 3, g1 = (lane >> 4) & 1, q = lane & 3;
    const int krow0 = 16 * wl + 8 * h + qq;
    const int fsw = ((krow0 >> 1) & 1) << 2;            // (row + 4 has the same bit 1)
    auto read_frag = [&](const unsigned char *st, int u) __attribute__((always_inline)) -> bf16x8 {   // u: 0..5 = A block u / 3 plane u % 3; 6..11 = B
      const int isb = u >= 6, v = isb ? u - 6 : u, blk = v / 3, pl = v % 3;
      const unsigned char *pb = st + (3 * isb + pl) * PLANE;
      if (!isb || B_KC) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pb + blk * 4096 + fkc));
      const unsigned char *p = pb + krow0 * 128 + ((((4 * blk + 2 * g1 + (q >> 1)) ^ fsw)) << 4) + 8 * (q & 1);
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * 128));
      return __builtin_bit_cast(bf16x8, s16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
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
    // MFMA t (0..23): product p = t >> 2 in the order lo.hi hi.lo mid.mid mid.hi hi.mid hi.hi (B plane . A plane), block t & 3; the
    // product is formed transposed (first operand = the B rows): a lane ends with 4 x 4 consecutive output columns of one row
    auto mf = [&](const bf16x8 (&f)[12], int t) __attribute__((always_inline)) {
      const int p = t >> 2, blk = t & 3, bm = blk >> 1, bn = blk & 1;
      const int pa = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;
      const int pb = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;
      const int cls = p < 5 ? 1 : 0;
      acc[blk][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[6 + 3 * bn + pb], f[3 * bm + pa], acc[blk][cls], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();      // chunk 0 has landed
    asm volatile("" ::: "memory");
    bf16x8 f0[12], f1[12];
#pragma unroll
    for (int u = 0; u < 12; u++) f0[u] = read_frag(lds, u);
    // phase c: the 24 MFMAs of chunk c from registers; behind the first the barrier (chunk c + 1 has landed for everybody, nobody reads
    // chunk c's stage any more: those reads were issued in phase c - 1); in the later gaps one fragment read of chunk c + 1 each
    auto phase = [&](auto full_tag, const bf16x8 (&fu)[12], bf16x8 (&fn)[12], int c, int stage) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      const bool next = FULL || c + 1 < nchunks;
      const unsigned char *sn = lds + ((stage + 1 == STAGES) ? 0 : stage + 1) * STAGE_BYTES;
      __builtin_amdgcn_sched_barrier(0);
      mf(fu, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#define X3P_RI(t) ((t) < 11 ? 0 : (t) > 22 ? 11 : (t) - 11)
#define X3P_GAP(t)                                                                                        \
      mf(fu, t);                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      if ((t) >= 11 && (t) <= 22 && next) fn[RORDER[X3P_RI(t)]] = read_frag(sn, RORDER[X3P_RI(t)]);       \
      __builtin_amdgcn_sched_barrier(0);
      X3P_GAP(1) X3P_GAP(2) X3P_GAP(3) X3P_GAP(4) X3P_GAP(5) X3P_GAP(6) X3P_GAP(7) X3P_GAP(8) X3P_GAP(9) X3P_GAP(10) X3P_GAP(11) X3P_GAP(12)
      X3P_GAP(13) X3P_GAP(14) X3P_GAP(15) X3P_GAP(16) X3P_GAP(17) X3P_GAP(18) X3P_GAP(19) X3P_GAP(20) X3P_GAP(21) X3P_GAP(22) X3P_GAP(23)
#undef X3P_GAP
#undef X3P_RI
    };
    {
      using T = BoolTag<true>;
      using F = BoolTag<false>;
      int c = 0, stage = 0;
      auto nxt = [&]() { stage = (stage + 1 == STAGES) ? 0 : stage + 1; };
      for (; c + 2 < nchunks; c += 2) {
        phase(T{}, f0, f1, c, stage); nxt();
        phase(T{}, f1, f0, c + 1, stage); nxt();
      }
      for (; c + 1 < nchunks; c += 2) {
        phase(F{}, f0, f1, c, stage); nxt();
        phase(F{}, f1, f0, c + 1, stage); nxt();
      }
      if (c < nchunks) phase(F{}, f0, f1, c, stage);
    }
    // ---- this wave's partial tile -> LDS ([wave][block][register group][lane] float4), classes small -> large
    __syncthreads();      // (no DMA is in flight, every fragment is in registers: the stages are free)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[b][1][4 * g + e] + acc[b][0][4 * g + e];
        *reinterpret_cast<f32x4 *>(lds + (((wl * 4 + b) * 4 + g) * 64 + lane) * 16) = v;
      }
  }
  if (loader) __syncthreads();
  __syncthreads();

  // ---- epilogue, all eight waves: wave w finishes block (bm, bn) = w & 3, register groups 2 (w >> 2) and + 1: lane (i, h) holds row
  // 32 bm + i, columns 32 bn + 8 g + 4 h + (0..3)
  const int blk = w & 3, bm = blk >> 1, bn = blk & 1, i = lane & 31, h = lane >> 5;
  const int em = m0 + 32 * bm + i;
  const bool relu = G.act == 0;
  float *red = reinterpret_cast<float *>(lds + 65536);      // [bm][64 columns] column sums of the blocks' 32 rows
#pragma unroll
  for (int gg = 0; gg < 2; gg++) {
    const int g = 2 * (w >> 2) + gg;
    f32x4 o = *reinterpret_cast<const f32x4 *>(lds + (((0 * 4 + blk) * 4 + g) * 64 + lane) * 16);
#pragma unroll
    for (int ww = 1; ww < 4; ww++) o += *reinterpret_cast<const f32x4 *>(lds + (((ww * 4 + blk) * 4 + g) * 64 + lane) * 16);
    const int col = 32 * bn + 8 * g + 4 * h, n = n0 + col;
    if (EPI == mg::EPI_BIAS_ACT) {
      const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(G.bias + n);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = fmaxf(o[e] + bias4[e], 0.0f);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = tanhf(o[e] + bias4[e]);
      }
    }
    if (EPI == mg::EPI_GATE_COLSUM) {
      const f32x4 gt = *reinterpret_cast<const f32x4 *>(G.gate + (int64_t)em * G.ldg + n);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = gt[e] > 0.0f ? o[e] : 0.0f;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = o[e] * (1.0f - gt[e] * gt[e]);
      }
      if (G.colsum != nullptr) {
        f32x4 cs = o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float c = row16_sum(cs[e]);
          c += __shfl_xor(c, 16, 64);
          cs[e] = c;
        }
        if (i == 0) *reinterpret_cast<f32x4 *>(red + bm * 64 + col) = cs;
      }
    }
    *reinterpret_cast<f32x4 *>(G.c + (int64_t)em * G.ldc + n) = o;
    if (G.cp != nullptr) store_planes4(G.cp + (int64_t)em * G.ldcp + n, G.scp, o);
  }
  if (EPI == mg::EPI_GATE_COLSUM && G.colsum != nullptr) {
    __syncthreads();
    if (tid < 64) G.colsum[(int64_t)tm * G.N + n0 + tid] = red[tid] + red[64 + tid];
  }
}

// fp32 [n] -> three bf16 planes (plane p at planes + p * stride): the producers that are not a product of this file (the minibatch
// gather's observations, a library product's output)
__global__ __launch_bounds__(256) void k_split_planes(const float *x, uint16_t *planes, int64_t stride, int64_t n4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  store_planes4(planes + 4 * idx, stride, reinterpret_cast<const f32x4 *>(x)[idx]);
}

}  // namespace x3p
