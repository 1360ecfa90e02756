// mlp_infer.hpp — one layer of the policy MLP for the 16-bit INFERENCE path of the rollout / evaluators:
//     y[M, N] = act(x[M, K] W[N, K]^T + b[N]),   x, W, y bf16 (or fp16), fp32 accumulation, b fp32
// (src/models.py:23-33: hk.Linear + relu, four times per forward; roll_out.py:73-84 calls it once per env.step.)
// Opt-in precision (`inference_dtype`), never the default — the fp32 path stays with the library GEMM.
//
// Why not the library GEMM: at M = 8192 tables, N = K = 1024 hipBLASLt's pick runs 20.6 us = 0.83 PFLOP/s (a third of the
// dense bf16 peak; profiles/r03/r03h_policy_rollout_bf16_graph_kernel_stats.txt), and the rollout issues 512 of them (this
// kernel: 18.0-18.9 us averaged over the four layers, profiles/r03/r03k_policy_rollout_bf16_graph_kernel_stats.txt).  Here:
//   * 256 (M) x 128 (N) output tile per 512-thread workgroup: 8192 x 1024 = 256 tiles = ONE per CU, all resident at once;
//     workgroup -> tile so that an XCD owns 4 row tiles x all 8 column tiles (its L2 holds W once and 2 MB of x);
//   * v_mfma_f32_32x32x16_bf16, 8 waves as 4 (M) x 2 (N), 64 x 64 per wave = 2 x 2 accumulators; the product is formed
//     TRANSPOSED (W rows are the MFMA's A operand): a lane then owns 4 consecutive outputs of one row of y;
//   * operands staged global -> LDS by the DMA path (global_load_lds, 16 B per lane, no VGPR round trip): 3 stages of
//     48 KB (64-deep K chunks), two chunks in flight across ONE raw s_barrier per chunk (counted vmcnt, never 0 in the loop);
//     128-byte LDS rows whose 16-byte pieces are XOR-swizzled on the SOURCE address, so every ds_read_b128 of 32 rows is
//     conflict-free; fragments of chunk c + 1 are read while chunk c is multiplied (two register sets);
//   * per MFMA gap at most one DMA instruction and one or two fragment reads (the two waves of a SIMD run in step);
//   * epilogue: + bias, ReLU, round to 16 bits (v_cvt_pk_bf16_f32), through LDS in two halves behind raw barriers, rows stored
//     256 B at a time, write-through (no dirty L2 line is left for the end of the launch);
//   * optionally (brl_linear_act_heads) the tile is multiplied, while it sits in LDS, with its 128 columns of the policy
//     heads' weights: partial products per column tile, summed by brl_policy_step_ex — and y itself need not be stored;
//   * K need not be a multiple of 64 (the observation is 480 wide): the pieces of the last chunk that lie beyond K are
//     fetched from a 16-byte block of zeros instead.
// Included by brl_infer16.hip.
#pragma once

namespace lin16 {

constexpr int BM = 256, BN = 128, BK = 64, STAGES = 3, THREADS = 512;
constexpr int A_BYTES = BM * BK * 2;               // the x tile of a chunk (32 KB)
constexpr int B_BYTES = BN * BK * 2;               // the W tile (16 KB)
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;     // 48 KB
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;    // 144 KB (+ 512 B of bias) of the CU's 160 KB
constexpr int C_ROW_BYTES = BN * 2 + 16;           // output tile in LDS: 272-byte rows (conflict-free 8-byte writes, 16-byte aligned)
constexpr int HW_OFF = BM * C_ROW_BYTES;           // behind it: the tile's slice of the head weights, 48 rows of the same stride
static_assert(HW_OFF + 48 * C_ROW_BYTES <= LDS_BYTES, "the output tile and the head weights reuse the stages");

typedef short b16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

struct Args {
  const uint16_t *x;   // [M][ldx]
  int64_t ldx;
  const uint16_t *w;   // [N][ldw]  (nn.Linear's own layout)
  int64_t ldw;
  const float *bias;   // [N] or NULL
  uint16_t *y;         // [M][ldy]
  int64_t ldy;
  int M, N, K;         // N % 128 == 0, K % 8 == 0, ldx / ldw / ldy % 8 == 0
  int relu;
  int store_mode;      // y stores: 0 = plain, 1 = non-temporal, 2 = write-through (sc0 sc1)
  // the policy heads' share of this launch (brl_linear_act_heads), or head_part == NULL
  const uint16_t *head_w;   // [n_heads][ld_head_w], same 16-bit type as y
  int64_t ld_head_w;
  int n_heads;              // <= 48
  float *head_part;         // [N / 128][head_part_stride]: part p, row i, head j at p * head_part_stride + i * head_part_ld + j
  int64_t head_part_ld, head_part_stride;
#ifdef LIN16_TIMING
  unsigned long long *dbg;
#endif
};
#ifdef LIN16_TIMING
#define LIN16_STAMP(k) do { if (threadIdx.x == 0 && G.dbg) G.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LIN16_STAMP(k) do { } while (0)
#endif
#ifndef LIN16_SCHED
#define LIN16_SCHED 1   // 0: the six DMA instructions of a phase back to back, reads two per gap behind them (experiment)
#endif
#ifndef LIN16_EXP
#define LIN16_EXP 0   // timing experiments: 1 = no DMA in the K loop, 2 = no MFMA, 4 = no fragment reads
#endif

__device__ __attribute__((aligned(16))) const uint32_t lin16_zero[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <bool V>
struct BoolTag { static constexpr bool value = V; };

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int FMT>
__device__ __forceinline__ f32x4 mma16(const b16x8 a, const b16x8 b, const f32x4 c) {
  if (FMT == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

template <int FMT>   // 1 = bf16, 2 = fp16
__device__ __forceinline__ f32x16 mma(const b16x8 a, const b16x8 b, const f32x16 c) {
  if (FMT == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

template <int FMT>
__device__ __forceinline__ uint32_t pack2(const f32x2 v) {   // two floats -> two 16-bit values, round to nearest even
  if (FMT == 1) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2));
}

// One 256 x 128 tile of y = act(x W^T + b): tile (tm, tn), by all 512 threads of the workgroup.  `lds`: LDS_BYTES, `bias_s`: BN floats.
// The caller puts a workgroup barrier between two tiles (the output tile leaves through the stages' memory).
template <int FMT>
__device__ __forceinline__ void linear_tile(const Args &G, char *lds, float *bias_s, const int tm, const int tn) {
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nchunks = (G.K + BK - 1) / BK;
  const bool ktail = (G.K % BK) != 0;

  if (tid < BN) bias_s[tid] = G.bias ? G.bias[n0 + tid] : 0.0f;   // (read behind the barriers of the K loop)

  // ---- the heads' share (brl_linear_act_heads): part[row][head] = sum over this tile's 128 columns of y[row][col] head_w[head][n0 +
  // col].  The tile's slice of the head weights — 48 rows (n_heads <= 48, the rest zeros) x 256 B — is requested NOW, ahead of
  // every DMA instruction (loads return in order: the counted vmcnt waits below still mean what they say), one or two 16-byte
  // pieces per thread; it goes to LDS behind the K loop.  (Measured: every wave fetching its own operands in the epilogue cost
  // 5 k cycles — 2048 waves after the same 80 KB.)
  const bool heads = G.head_part != nullptr;
  uint4 hwv[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  if (heads) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int hrow = (tid >> 4) + 32 * u;   // 0..63; rows >= 48 are not kept
      if (hrow < G.n_heads) hwv[u] = *reinterpret_cast<const uint4 *>(G.head_w + (int64_t)hrow * G.ld_head_w + n0 + 8 * (tid & 15));
    }
  }

  // ---- staging: a chunk is 48 DMA instructions of 1 KB = 8 rows x 128 B; wave w issues x rows 32 w .. 32 w + 31 (4) and
  // W rows 16 w .. 16 w + 15 (2).  LDS slot (row, p) holds the row's logical 16-byte piece p ^ ((row >> 1) & 7).
  // Source address = a wave-uniform base that walks K + a per-lane 32-bit byte offset fixed for the whole launch.
  uint32_t offa[4], offb[2];
  int pca[4], pcb[2];   // the logical piece each instruction's lane fetches (K-tail test)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = 32 * w + 8 * j + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    const int m = (m0 + row < G.M) ? m0 + row : G.M - 1;
    offa[j] = (uint32_t)(((int64_t)m * G.ldx + 8 * c) * 2);
    pca[j] = c;
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int row = 16 * w + 8 * j + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    offb[j] = (uint32_t)(((int64_t)(n0 + row) * G.ldw + 8 * c) * 2);
    pcb[j] = c;
  }
  const char *const basea = reinterpret_cast<const char *>(G.x), *const baseb = reinterpret_cast<const char *>(G.w);
  int kc = 0;   // chunk the DMA stream is at
  const int dsta = 32 * w * 128, dstb = A_BYTES + 16 * w * 128;
  // (the empty asm keeps each offset a 32-bit value defined right here: the address then folds into the instruction as
  //  scalar base (the kernel argument) + 32-bit lane offset instead of a loop-carried 64-bit pointer per lane)
  auto stage_one = [&](auto full_tag, char *st, int j) {   // DMA instruction j of a chunk (0..3: x rows, 4..5: W rows)
    constexpr bool FULL = decltype(full_tag)::value;       // FULL: the chunk lies inside K
    const uint32_t kb = (uint32_t)kc * (BK * 2);
    const bool isa = j < 4;
    uint32_t o = (isa ? offa[j & 3] : offb[j & 1]) + kb;
    char *dst = st + (isa ? dsta + (j & 3) * 1024 : dstb + (j & 1) * 1024);
    const char *base = isa ? basea : baseb;
    if (FULL) {
      asm volatile("" : "+v"(o));
      glds16(base + o, dst);
    } else {   // .. or it is the last, partial one: pieces beyond K come from the block of zeros
      const int left = G.K - kc * BK;   // > 0
      const int pc = isa ? pca[j & 3] : pcb[j & 1];
      glds16((8 * pc < left) ? (const void *)(base + o) : (const void *)lin16_zero, dst);
    }
  };
  auto stage_any = [&](char *st) {
#pragma unroll
    for (int j = 0; j < 6; j++) stage_one(BoolTag<false>{}, st, j);
  };
  auto advance = [&]() { kc++; };

  // ---- fragments: wave (wm, wn) owns rows 64 wm .., columns 64 wn .. of the tile; lane = (i, h): operand row i of a 32-row
  // block, K pieces 2 ks + h (ks = 0..3: the four 16-deep MFMA steps of a chunk)
  const int wm = w >> 1, wn = w & 1, i = lane & 31, h = lane >> 5;
  const int sw = (i >> 1) & 7;
  const int xa = (wm * 64 + i) * 128, wa = A_BYTES + (wn * 64 + i) * 128;
  int ko[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) ko[ks] = ((2 * ks + h) ^ sw) << 4;
  auto read_x = [&](const char *st, b16x8 (&xf)[2][4], int mb, int ks) {
    xf[mb][ks] = *reinterpret_cast<const b16x8 *>(st + xa + mb * 4096 + ko[ks]);
  };
  auto read_w = [&](const char *st, b16x8 (&wf)[2][4], int nb, int ks) {
    wf[nb][ks] = *reinterpret_cast<const b16x8 *>(st + wa + nb * 4096 + ko[ks]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

#define L16_SB() __builtin_amdgcn_sched_barrier(0)
  // MFMA number t of a phase (t = 0..15): K step t >> 2, blocks (mb, nb) = ((t >> 1) & 1, t & 1)
#define L16_MF(xu, wu, t)                                                                                          \
  if (!(LIN16_EXP & 2)) {                                                                                          \
    acc[((t) >> 1) & 1][(t) & 1] = mma<FMT>(wu[(t) & 1][(t) >> 2], xu[((t) >> 1) & 1][(t) >> 2], acc[((t) >> 1) & 1][(t) & 1]); \
  } else {                                                                                                         \
    acc[((t) >> 1) & 1][(t) & 1][(t) & 15] += (float)wu[(t) & 1][(t) >> 2][0] * (float)xu[((t) >> 1) & 1][(t) >> 2][0];      \
  }                                                                                                                \
  L16_SB()
  // the 16 fragment reads of a chunk in the order the MFMAs want them: K step by K step, (w0, x0, w1, x1)
#define L16_RD(sn, xn, wn_, u)                                                                                     \
  if (!(LIN16_EXP & 4)) {                                                                                          \
    if (((u) & 1) == 0) read_w(sn, wn_, ((u) >> 1) & 1, (u) >> 2);                                                 \
    else read_x(sn, xn, ((u) >> 1) & 1, (u) >> 2);                                                                 \
  }

  // ---- phase c: multiplies chunk c from registers (16 MFMAs) and, between them (order pinned with sched_barrier(0)):
  //   waits for its own DMA pieces of chunk c + 1 and meets the other waves at the barrier — behind it chunk c + 1 has landed
  //   for everybody and nobody reads chunk c's stage any more (those reads were issued in phase c - 1, lgkmcnt(0) in front
  //   of the barrier) —, issues its 6 DMA instructions of chunk c + 3 into that stage, reads the fragments of chunk c + 1
  //   into the other register set.  In flight across the barrier: chunk c + 2 (vmcnt(6)).
  // FULL phases (chunk c + 1 exists, chunk c + 3 exists and lies inside K) are branch-free.
  auto phase = [&](auto full_tag, const b16x8 (&xu)[2][4], const b16x8 (&wu)[2][4], b16x8 (&xn)[2][4],
                   b16x8 (&wn_)[2][4], int c, int stage) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool next = FULL || c + 1 < nchunks;
    const bool dma = (LIN16_EXP & 1) ? false : (FULL || c + 3 < nchunks);
    char *st = lds + stage * STAGE_BYTES;                                    // stage of chunk c (refilled with chunk c + 3)
    const char *sn = lds + ((stage + 1 == STAGES) ? 0 : stage + 1) * STAGE_BYTES;   // stage of chunk c + 1
    L16_SB();
    L16_MF(xu, wu, 0);
    if (next) {
      if (FULL || c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    L16_SB();
    // A DMA instruction holds its wave's issue for ~60 cycles and the two waves of a SIMD run in step: six of them back to back
    // leave the matrix pipe with nothing to do (measured: +2.9 us per launch).  One per MFMA gap, a fragment read beside it.
#if LIN16_SCHED == 0
#define L16_GAP(t)                                                                       \
    L16_MF(xu, wu, t);                                                                   \
    if ((t) == 1 && dma) {                                                               \
      _Pragma("unroll") for (int j = 0; j < 6; j++) stage_one(full_tag, st, j);          \
    }                                                                                    \
    if ((t) >= 6 && (t) <= 13 && next) { L16_RD(sn, xn, wn_, 2 * ((t) - 6)) L16_RD(sn, xn, wn_, 2 * ((t) - 6) + 1) } \
    L16_SB();
#else
#define L16_GAP(t)                                                                       \
    L16_MF(xu, wu, t);                                                                   \
    if ((t) >= 1 && (t) <= 6 && dma) stage_one(full_tag, st, (t) - 1);                   \
    if (next) {                                                                          \
      if ((t) >= 1 && (t) <= 6) { L16_RD(sn, xn, wn_, (t) - 1) }                         \
      else if ((t) == 7) { L16_RD(sn, xn, wn_, 6) L16_RD(sn, xn, wn_, 7) }               \
      else if ((t) >= 8) { L16_RD(sn, xn, wn_, (t)) }                                    \
    }                                                                                    \
    L16_SB();
#endif
    L16_GAP(1) L16_GAP(2) L16_GAP(3) L16_GAP(4) L16_GAP(5) L16_GAP(6) L16_GAP(7) L16_GAP(8)
    L16_GAP(9) L16_GAP(10) L16_GAP(11) L16_GAP(12) L16_GAP(13) L16_GAP(14) L16_GAP(15)
#undef L16_GAP
    if (dma) advance();
  };

  b16x8 x0[2][4], w0[2][4], x1[2][4], w1[2][4];
  LIN16_STAMP(0);
#ifndef LIN16_PROLOGUE
#define LIN16_PROLOGUE 1   // 0: all three stages requested before the first wait (experiment)
#endif
  // prologue: every stage filled.  All 256 workgroups start at once and ask for 3 x 48 KB each: chunk 0 — the only one the first
  // MFMA waits for — would share the memory system with 25 MB of chunks 1 and 2.  So chunk 2 is requested only when chunk 0
  // has arrived (it is needed two phases later).
  for (int c = 0; c < (LIN16_PROLOGUE ? 2 : STAGES) && c < nchunks; c++) {
    stage_any(lds + c * STAGE_BYTES);
    advance();
  }
  {
    if (LIN16_PROLOGUE) {
      if (nchunks >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (nchunks >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (nchunks == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (LIN16_PROLOGUE && nchunks >= 3) {
      stage_any(lds + 2 * STAGE_BYTES);
      advance();
    }
#pragma unroll
    for (int u = 0; u < 16; u++) { L16_RD(lds, x0, w0, u) }
  }
  LIN16_STAMP(1);
  using T = BoolTag<true>;
  using F = BoolTag<false>;
  const int nfull = nchunks - 3 - (ktail ? 1 : 0);   // phases c < nfull are FULL
  {
    int c = 0, stage = 0;
    auto nxt = [&]() { stage = (stage + 1 == STAGES) ? 0 : stage + 1; };
    for (; c + 1 < nfull; c += 2) {
      phase(T{}, x0, w0, x1, w1, c, stage); nxt();
      phase(T{}, x1, w1, x0, w0, c + 1, stage); nxt();
    }
    for (; c + 1 < nchunks; c += 2) {
      phase(F{}, x0, w0, x1, w1, c, stage); nxt();
      phase(F{}, x1, w1, x0, w0, c + 1, stage); nxt();
    }
    if (c < nchunks) phase(F{}, x0, w0, x1, w1, c, stage);
  }
  LIN16_STAMP(2);
#undef L16_MF
#undef L16_RD
#undef L16_SB

  // ---- epilogue.  Accumulator register r of lane (i, h) of block (mb, nb) = y[row 64 wm + 32 mb + i][column 64 wn + 32 nb +
  // 8 (r >> 2) + 4 h + (r & 3)]: four consecutive columns per register group -> one 8-byte LDS write
  __syncthreads();   // every wave is done with the stages (its last fragments are in registers; no DMA is in flight)
  LIN16_STAMP(5);
  if (heads) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int hrow = (tid >> 4) + 32 * u;
      if (hrow < 48) *reinterpret_cast<uint4 *>(lds + HW_OFF + hrow * C_ROW_BYTES + 16 * (tid & 15)) = hwv[u];
    }
  }
  const float floor_v = G.relu ? 0.0f : -__builtin_inff();
  // Two halves (mb = 0: tile rows 64 wm + 0..31, mb = 1: + 32..63): the second half is packed while the first half's stores are
  // on their way.  Raw barriers with lgkmcnt(0) only: __syncthreads() would also wait for those stores (vmcnt(0)).
#pragma unroll
  for (int mb = 0; mb < 2; mb++) {
    const int row = wm * 64 + mb * 32 + i;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int col = wn * 64 + nb * 32 + 8 * q + 4 * h;
        const float4 bv = *reinterpret_cast<const float4 *>(&bias_s[col]);
        f32x2 lo = {acc[mb][nb][4 * q + 0], acc[mb][nb][4 * q + 1]}, hi = {acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]};
        lo = lo + f32x2{bv.x, bv.y};   // (v_pk_add_f32)
        hi = hi + f32x2{bv.z, bv.w};
        lo.x = fmaxf(lo.x, floor_v); lo.y = fmaxf(lo.y, floor_v);
        hi.x = fmaxf(hi.x, floor_v); hi.y = fmaxf(hi.y, floor_v);
        uint2 pk;
        pk.x = pack2<FMT>(lo);
        pk.y = pack2<FMT>(hi);
        *reinterpret_cast<uint2 *>(lds + row * C_ROW_BYTES + col * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (mb == 0) LIN16_STAMP(6);
    if (G.y != nullptr && !((LIN16_EXP & 8) && G.M > 0)) {
#pragma unroll
      for (int it = 0; it < (BM * BN) / (THREADS * 16); it++) {   // this half: 128 rows x 16 pieces of 16 B
        const int idx = it * THREADS + tid, r128 = idx >> 4, ch = idx & 15;
        const int trow = (r128 >> 5) * 64 + mb * 32 + (r128 & 31);
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + trow * C_ROW_BYTES + ch * 16);
        if (m0 + trow < G.M) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4 *dst = reinterpret_cast<u32x4 *>(G.y + (int64_t)(m0 + trow) * G.ldy + n0 + ch * 8);
          const u32x4 d = {v.x, v.y, v.z, v.w};
          if (G.store_mode == 1) __builtin_nontemporal_store(d, dst);
          else if (G.store_mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(d) : "memory");
          else *dst = d;
        }
      }
    }
  }
  LIN16_STAMP(4);
  if (heads) {
    // y as it stands in LDS (rounded to 16 bits — what a separate head product would read back).  Wave w: tile rows 32 w .. 32 w + 31 =
    // two 16-row MFMA tiles x three 16-head blocks x four 32-deep steps of v_mfma_f32_16x16x32, formed transposed (A = head
    // 16 nb + r, B = y row r, K pieces 32 ks + 8 kq): a lane ends up with four consecutive heads of one row = one 16-byte store
    const int r = lane & 15, kq = lane >> 4, rq = lane >> 4;
    float *part = G.head_part + (int64_t)tn * G.head_part_stride;
    b16x8 hb[3][4];
#pragma unroll
    for (int nb = 0; nb < 3; nb++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
        hb[nb][ks] = *reinterpret_cast<const b16x8 *>(lds + HW_OFF + (16 * nb + r) * C_ROW_BYTES + 64 * ks + 16 * kq);
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
      const char *ya = lds + (32 * w + 16 * rb + r) * C_ROW_BYTES + 16 * kq;
      f32x4 hacc[3];
#pragma unroll
      for (int nb = 0; nb < 3; nb++) hacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const b16x8 yv = *reinterpret_cast<const b16x8 *>(ya + 64 * ks);
#pragma unroll
        for (int nb = 0; nb < 3; nb++) hacc[nb] = mma16<FMT>(hb[nb][ks], yv, hacc[nb]);
      }
      // accumulator register q of lane (r, rq): head 16 nb + 4 rq + q of tile row r (heads beyond n_heads: exact zeros)
      const int m = m0 + 32 * w + 16 * rb + r;
#pragma unroll
      for (int nb = 0; nb < 3; nb++) {
        const int n = 16 * nb + 4 * rq;
        if (n < G.n_heads && m < G.M) {   // write-through like y: no dirty line is left for the end of the launch to write back
          f32x4 *dst = reinterpret_cast<f32x4 *>(part + (int64_t)m * G.head_part_ld + n);
          if (G.store_mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(hacc[nb]) : "memory");
          else *dst = hacc[nb];
        }
      }
    }
  }
  LIN16_STAMP(3);
}

// workgroup -> logical id: blocks b, b + 8, .. share an XCD; give them CONSECUTIVE logical ids (bijective for any grid size)
__device__ __forceinline__ int xcd_logical_id(int b, int nblk) {
  const int q = nblk / 8, r = nblk % 8, xcd = b % 8;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

template <int FMT>
__global__ __launch_bounds__(THREADS) void k_linear16(Args G) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ float bias_s[BN];
  // the column tiles walk fastest: an XCD's 32 workgroups = 4 row tiles x all 8 column tiles at N = 1024 (its L2 holds W once)
  const int tiles_n = G.N / BN;
  const int L = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);
  const int tm = L / tiles_n;
  linear_tile<FMT>(G, lds, bias_s, tm, L - tm * tiles_n);
}

}  // namespace lin16
