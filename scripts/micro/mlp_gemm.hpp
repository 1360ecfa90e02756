// mlp_gemm.hpp — the fp32 MFMA GEMMs of one PPO minibatch step (src/update.py:74-242: forward, dW = dz^T h_prev, dh = dz W),
// hand-written for gfx950.  EXPERIMENTAL: built and checked by scripts/micro/gemm_f32_test.hip only — not part of
// libbrl_hip.so (86 TFLOP/s per launch against the library GEMM's 107 on the step's 1024^3 products:
// profiles/r03/r03_experiments.txt).
//
// Why not the library GEMM: at minibatch 1024 every product of the step is ~1024^3.  hipBLASLt's heuristic picks 128 x 128
// tiles = 64 workgroups on a 256-CU chip (20 us = 107 TFLOP/s for 2.1 GFLOP, profiles/r02w), and leaves ReLU backward,
// the bias column sums and their launch boundaries to separate kernels.  Here:
//   * 64 x 64 output tile per 256-thread workgroup -> 256 workgroups for a 1024 x 1024 result = one per CU; the pair of
//     backward products of one layer (dW_l and dh_{l-1}, both fed by dz_l) is ONE launch of 512 workgroups = two per CU,
//     i.e. two waves per SIMD that cover each other's stalls;
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per instruction and SIMD = the chip's 157 TFLOP/s fp32 peak): each
//     of the 4 waves owns a 32 x 32 quadrant, 16 MFMAs per 32-deep K chunk;
//   * operands staged global -> LDS by the DMA path (global_load_lds_dwordx4, no VGPR round trip), 4 stages of 16 KB, three
//     chunks in flight across ONE raw s_barrier per chunk (counted vmcnt, never 0 inside the loop);
//   * both operand layouts without a transpose pass: "KC" = K contiguous in memory (x [B,K], W [N,K] of the forward pass):
//     128-byte LDS rows whose 16-byte chunks are XOR-swizzled ON THE SOURCE ADDRESS so that ds_read_b128 of 32 rows is
//     conflict-free — one b128 read feeds 4 MFMAs (the K order inside a chunk is permuted identically for A and B);
//     "MC" = the M / N index contiguous (dz^T, h_prev, W of the backward pass): 256-byte LDS rows read with ds_read_b32,
//     lane = column, conflict-free as it stands;
//   * epilogues: + bias and ReLU (forward); * (h > 0) = ReLU backward in the dh product, and the bias gradient's column
//     sums of what is stored, per 32-row block, in a fixed order (deterministic; finished by k_bias_finalize).
// Numerics: an MFMA accumulator is a k-ordered fp32 fma chain (one rounding per product, cdna_hip_programming.md §3) —
// the same class of result as the library's fp32 GEMM; tests compare against float64.
#pragma once

namespace mlpg {

constexpr int BM = 64, BN = 64, BK = 32, STAGES = 4, THREADS = 512;
constexpr int OPER_FLOATS = 64 * BK;            // 8 KB per operand tile
constexpr int STAGE_FLOATS = 2 * OPER_FLOATS;   // A tile, then B tile
constexpr int LDS_FLOATS = STAGES * STAGE_FLOATS;  // 64 KB: two workgroups per CU fit in 160 KB

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *A;   // KC: A[m * lda + k]   MC: A[k * lda + m]
  int64_t lda;
  const float *B;   // KC: B[n * ldb + k]   MC: B[k * ldb + n]
  int64_t ldb;
  float *C;         // C[m * ldc + n]
  int64_t ldc;
  int M, N, K;      // K % 32 == 0; M, N >= 4 and multiples of 4 (16-byte chunks); partial edge tiles are fine
  const float *bias;   // [N] added per column, or NULL
  int relu;            // C = max(C, 0)
  const float *gate;   // [M][ldg]: C = (gate > 0) ? C : 0  (ReLU backward: gate = the layer's forward output), or NULL
  int64_t ldg;
  float *colsum;       // [ceil(M / 32)][N]: column sums of the STORED values per 32-row block (bias gradient), or NULL
#ifdef MLPG_TIMING
  unsigned long long *dbg;   // timing build: per workgroup 8 words (shader cycles at 4 points, 100 MHz ticks at 4 points)
#endif
};
#ifdef MLPG_TIMING
#define MLPG_STAMP(k) do { if (threadIdx.x == 0 && G.dbg) { G.dbg[(size_t)bid * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                                                            G.dbg[(size_t)bid * 8 + 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MLPG_STAMP(k) do { } while (0)
#endif

#ifndef MLPG_EXP
#define MLPG_EXP 0   // timing experiments of scripts/micro/gemm_f32_test.hip: 1 = no DMA inside the K loop, 2 = no MFMA
#endif
__device__ __forceinline__ void glds16(const float *g, float *lds_wave_base) {
  // 16 bytes per lane, global -> LDS without a VGPR: the LDS address is wave-uniform base + 16 * lane
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// workgroup -> tile: consecutive logical ids (= one XCD under xcd_block) walk a strip of 4 row tiles column by column, so an
// XCD's share of a 16 x 16 tile grid is a 4 x 8 block: 1 MB of A rows + 2 MB of B rows in its 4 MB L2.  Speed only.
__device__ __forceinline__ void tile_of(int bid, int nblk, int tiles_m, int tiles_n, int &tm, int &tn) {
  const int L = (nblk % 8 == 0) ? (bid % 8) * (nblk / 8) + bid / 8 : bid;
  if (tiles_m % 4 == 0) {
    tm = (L / (4 * tiles_n)) * 4 + (L & 3);
    tn = (L >> 2) % tiles_n;
  } else {
    tm = L / tiles_n;
    tn = L - tm * tiles_n;
  }
}

template <bool V>
struct BoolTag { static constexpr bool value = V; };

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_tile(const Args &G, float *lds, int bid, int nblk) {
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (G.M + BM - 1) / BM, tiles_n = (G.N + BN - 1) / BN;
  if (bid >= tiles_m * tiles_n) return;   // (a paired launch is sized for the larger product; whole workgroups leave)
  int tm, tn;
  tile_of(bid, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
  (void)nblk;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nchunks = G.K / BK;

  // ---- staging: a chunk is 16 DMA instructions of 1 KB (8 for the A image, 8 for the B image); wave w issues numbers
  // 2 w and 2 w + 1, so waves 0..3 stage A and waves 4..7 stage B.  Source address = a wave-UNIFORM base that walks K
  // (scalar adds) + a per-lane 32-bit offset fixed for the whole launch.
  const bool stB = w >= 4;                    // (wave-uniform) this wave stages the B operand
  uint32_t off[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int q = (2 * (w & 3) + j) * 64 + lane;   // 16-byte slot of the operand's 8 KB image
    const bool kcl = stB ? B_KC : A_KC;
    const int x0 = stB ? n0 : m0, X = stB ? G.N : G.M;
    const int64_t ld = stB ? G.ldb : G.lda;
    if (kcl) {   // slot q = row r (128 B of K), position p holds logical chunk p ^ swz(r)
      const int r = q >> 3, c = (q & 7) ^ ((r >> 1) & 7);
      const int x = (x0 + r < X) ? x0 + r : X - 1;
      off[j] = (uint32_t)((int64_t)x * ld + 4 * c);
    } else {     // slot q = K row kr (256 B of M / N), chunk ch
      const int kr = q >> 4, ch = q & 15;
      const int x = (x0 + 4 * ch + 4 <= X) ? x0 + 4 * ch : X - 4;
      off[j] = (uint32_t)((int64_t)kr * ld + x);
    }
  }
  const int64_t step = stB ? (B_KC ? (int64_t)BK : (int64_t)BK * G.ldb) : (A_KC ? (int64_t)BK : (int64_t)BK * G.lda);
  const float *const base = stB ? G.B : G.A;
  // Every tile starts its K loop at a DIFFERENT chunk (and wraps): workgroups that walk K in step would all read the same
  // 128-byte column of every row at the same time.  A sum over K does not care where it starts; the start is a function
  // of the tile, so results stay deterministic.
  int kc = (int)(((unsigned)(tm * 5 + tn * 3 + (tn >> 2))) % (unsigned)nchunks);   // chunk the DMA stream is at
  const float *cur = base + (int64_t)kc * step;
  const int dst = (stB ? OPER_FLOATS : 0) + (2 * (w & 3)) * 256;   // float offset of this wave's first 1 KB piece in a stage
  auto advance = [&]() {
    const bool wrap = (kc + 1 == nchunks);
    kc = wrap ? 0 : kc + 1;
    cur = wrap ? base : cur + step;
  };

  // ---- fragments.  Waves 0..3 multiply the first 16 K steps of every chunk, waves 4..7 the last 16 (two waves per SIMD:
  // while one waits for LDS data, issues its DMA or sits at the barrier, the other feeds the matrix pipe); within a half,
  // wave (wm, wn) owns the 32 x 32 quadrant; lane = (i, h): A row / B column i, K half-step h.  The two partial sums meet
  // in LDS at the end.
  const int kh = w >> 2, wq = w & 3, wm = wq >> 1, wn = wq & 1, i = lane & 31, h = lane >> 5;
  const int ra = wm * 32 + i, rb = wn * 32 + i;
  // KC: row r, logical 16-byte chunk 2 g + h at swizzled position ((2 g + h) ^ swz(r)); MC: K row 8 g + 4 h + s, column r
  int fa[2], fb[2];   // float offset of this wave's group gl (g = 2 kh + gl) inside a stage (MC: + 64 per K step)
#pragma unroll
  for (int gl = 0; gl < 2; gl++) {
    const int g = 2 * kh + gl;
    fa[gl] = A_KC ? ra * 32 + (((2 * g + h) ^ ((ra >> 1) & 7)) << 2) : (8 * g + 4 * h) * 64 + ra;
    fb[gl] = OPER_FLOATS + (B_KC ? rb * 32 + (((2 * g + h) ^ ((rb >> 1) & 7)) << 2) : (8 * g + 4 * h) * 64 + rb);
  }
  auto read_group = [&](const float *st, float (&av)[2][4], float (&bv)[2][4], int gl) {
    if (A_KC) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(st + fa[gl]);
      av[gl][0] = v.x; av[gl][1] = v.y; av[gl][2] = v.z; av[gl][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++) av[gl][s] = st[fa[gl] + 64 * s];
    }
    if (B_KC) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(st + fb[gl]);
      bv[gl][0] = v.x; bv[gl][1] = v.y; bv[gl][2] = v.z; bv[gl][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++) bv[gl][s] = st[fb[gl] + 64 * s];
    }
  };

  // two accumulator chains (even / odd K steps), added at the end: an MFMA never waits for the one right before it
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = acc2[r] = 0.0f;

  // ---- the K loop.  Phase c multiplies this wave's half of chunk c from registers (8 MFMAs) and, between them (pinned with
  // sched_barrier(0)): waits for its own DMA pieces of chunk c + 1 and meets the others at the barrier, issues its 2 DMA
  // instructions of chunk c + STAGES into the stage chunk c occupied, reads its fragments of chunk c + 1 into the other
  // register set.
  //   stage of chunk c is free for DMA once every wave has passed the barrier of phase c (its reads of chunk c — issued in
  //   phase c - 1 — are complete: lgkmcnt(0) in front of the barrier);  chunk c + 1 has landed once every wave has waited for its
  //   own DMA instructions of it (counted vmcnt: the STAGES - 2 chunks behind it stay in flight) and passed that same barrier.
  // FULL phases (c + STAGES < nchunks: a next chunk to read AND a chunk to fetch) are branch-free; the last STAGES phases
  // take the conditional form.
#define MLPG_SB() __builtin_amdgcn_sched_barrier(0)
#define MLPG_MF(au, bu, g, s)                                                                                      \
  if (!(MLPG_EXP & 2)) {                                                                                           \
    if ((s) & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(au[g][s], bu[g][s], acc2, 0, 0, 0);                   \
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(au[g][s], bu[g][s], acc, 0, 0, 0);                             \
  } else acc[(4 * g + s) & 15] += au[g][s] * bu[g][s];                                                             \
  MLPG_SB()
  auto phase = [&](auto full_tag, auto late_tag, const float (&au)[2][4], const float (&bu)[2][4], float (&an)[2][4],
                   float (&bn)[2][4], int c, int stage) {
    constexpr bool FULL = decltype(full_tag)::value, LATE = decltype(late_tag)::value;
    const bool next = (MLPG_EXP & 4) ? false : (FULL || c + 1 < nchunks);
    const bool dma = (MLPG_EXP & 1) ? false : (FULL || c + STAGES < nchunks);
    float *st = lds + stage * STAGE_FLOATS;                                        // stage of chunk c
    const float *sn = lds + ((stage + 1 == STAGES) ? 0 : stage + 1) * STAGE_FLOATS;   // stage of chunk c + 1
    MLPG_SB();
    MLPG_MF(au, bu, 0, 0);
    if (next && !(MLPG_EXP & 8)) {
      if (FULL) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      } else {
        const int behind = nchunks - 2 - c;   // chunks issued behind chunk c + 1 (at most STAGES - 2 of them are outstanding)
        if (behind >= STAGES - 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    MLPG_SB();
    MLPG_MF(au, bu, 0, 1);
    // The two waves of a SIMD run the same program and leave the barrier together: if both issued their DMA in the same
    // gaps, the matrix pipe would have nobody to feed it meanwhile (a DMA instruction occupies its wave's issue for ~60
    // cycles).  So the K halves are staggered: waves 0..3 issue DMA first and read fragments later, waves 4..7 read first and
    // issue DMA at the end of the phase, beside the other half's MFMA-only stretch.
    if (!LATE) {
      if (dma) glds16(cur + off[0], st + dst);
      MLPG_SB();
      MLPG_MF(au, bu, 0, 2);
      if (dma) {
        glds16(cur + off[1], st + dst + 256);
        advance();
      }
      MLPG_SB();
      MLPG_MF(au, bu, 0, 3);
      if (next) read_group(sn, an, bn, 0);
      MLPG_SB();
      MLPG_MF(au, bu, 1, 0);
      if (next) read_group(sn, an, bn, 1);
      MLPG_SB();
      MLPG_MF(au, bu, 1, 1);
      MLPG_MF(au, bu, 1, 2);
      MLPG_MF(au, bu, 1, 3);
    } else {
      if (next) read_group(sn, an, bn, 0);
      MLPG_SB();
      MLPG_MF(au, bu, 0, 2);
      if (next) read_group(sn, an, bn, 1);
      MLPG_SB();
      MLPG_MF(au, bu, 0, 3);
      MLPG_MF(au, bu, 1, 0);
      MLPG_MF(au, bu, 1, 1);
      if (dma) glds16(cur + off[0], st + dst);
      MLPG_SB();
      MLPG_MF(au, bu, 1, 2);
      if (dma) {
        glds16(cur + off[1], st + dst + 256);
        advance();
      }
      MLPG_SB();
      MLPG_MF(au, bu, 1, 3);
    }
  };

  float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
  MLPG_STAMP(0);
  for (int c = 0; c < STAGES && c < nchunks; c++) {   // prologue: every stage filled
    glds16(cur + off[0], lds + c * STAGE_FLOATS + dst);
    glds16(cur + off[1], lds + c * STAGE_FLOATS + dst + 256);
    advance();
  }
  {
    const int behind = nchunks - 1;   // chunks issued behind chunk 0
    if (behind >= STAGES - 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_group(lds, a0, b0, 0);
    read_group(lds, a0, b0, 1);
  }
  MLPG_STAMP(1);
  static_assert(STAGES == 4, "the phase pairs below walk the stages 0,1 | 2,3");
  using T = BoolTag<true>;
  using F = BoolTag<false>;
  auto k_loop = [&](auto late_tag) {
    int c = 0;
    for (; c + 1 + STAGES < nchunks; c += 2) {   // both phases of the pair are FULL
      phase(T{}, late_tag, a0, b0, a1, b1, c, c & 3);
      phase(T{}, late_tag, a1, b1, a0, b0, c + 1, (c + 1) & 3);
    }
    for (; c + 1 < nchunks; c += 2) {
      phase(F{}, late_tag, a0, b0, a1, b1, c, c & 3);
      phase(F{}, late_tag, a1, b1, a0, b0, c + 1, (c + 1) & 3);
    }
    if (c < nchunks) phase(F{}, late_tag, a0, b0, a1, b1, c, c & 3);
  };
  if (kh == 0) k_loop(F{});   // (wave-uniform; both forms pass the same barriers)
  else k_loop(T{});
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] += acc2[r];
  MLPG_STAMP(2);
#undef MLPG_MF
#undef MLPG_SB
  // the two K halves meet: waves 4..7 leave their quadrant's partial sums in LDS (register-major: lane-contiguous, no bank
  // conflicts), waves 0..3 add them — in that fixed order — and run the epilogue
  __builtin_amdgcn_s_barrier();   // (every wave is done with the stages: its last fragments are in registers, no DMA is in flight)
  asm volatile("" ::: "memory");
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; r++) lds[(wq * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] += lds[(wq * 16 + r) * 64 + lane];

  // ---- epilogue: accumulator register r of lane (i, h) = C[row (r & 3) + 8 (r >> 2) + 4 h][column i] of the quadrant
  const int n = n0 + wn * 32 + i;
  const bool nv = n < G.N;
  const float bias = (G.bias != nullptr && nv) ? G.bias[n] : 0.0f;
  float csum = 0.0f;
  float gv[16];
  if (G.gate != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      gv[r] = G.gate[(int64_t)((m < G.M) ? m : G.M - 1) * G.ldg + (nv ? n : G.N - 1)];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    float v = acc[r] + bias;
    if (G.relu) v = fmaxf(v, 0.0f);
    if (G.gate != nullptr) v = (gv[r] > 0.0f) ? v : 0.0f;
    const bool ok = nv && m < G.M;
    if (ok) G.C[(int64_t)m * G.ldc + n] = v;
    csum += ok ? v : 0.0f;
  }
  MLPG_STAMP(3);
  if (G.colsum != nullptr) {
    // rows of half 0 and half 1 interleave in blocks of 4: a fixed order (r ascending per half, then half 0 + half 1)
    const float tot = csum + __shfl_xor(csum, 32, 64);
    if (h == 0 && nv) G.colsum[(int64_t)(m0 / 32 + wm) * G.N + n] = tot;
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS) void k_mlp_gemm(Args G) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  gemm_tile<A_KC, B_KC>(G, lds, (int)blockIdx.x, (int)gridDim.x);
}

// the two backward products of one hidden layer in ONE launch (gridDim.y = 2): y = 0: dW = dz^T h_prev (A, B both MC),
// y = 1: dh = dz W with the ReLU-backward gate and column sums (A KC, B MC).  Two workgroups per CU = two waves per SIMD.
__global__ __launch_bounds__(THREADS) void k_mlp_gemm_bwd_pair(Args GW, Args GH) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  if (blockIdx.y == 0) gemm_tile<false, false>(GW, lds, (int)blockIdx.x, (int)gridDim.x);
  else gemm_tile<true, false>(GH, lds, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace mlpg
