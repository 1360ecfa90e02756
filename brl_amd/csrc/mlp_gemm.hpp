// mlp_gemm.hpp — fp32 MFMA GEMMs of one PPO minibatch step (src/update.py:74-242: the forward layers, dh = dz W, dW = dz^T h)
// with the step's elementwise work in their epilogues, hand-written for gfx950 (round 4; the round-3 attempt with LDS-DMA
// staging is scripts/micro/mlp_gemm.hpp, profiles/r03/r03_experiments.txt).
//
//   * 64 x 64 output tile per 256-thread workgroup: a 1024 x 1024 result is 256 workgroups = one per CU, one wave per SIMD;
//     wave (wm, wn) owns a 32 x 32 quadrant as 2 x 2 blocks of v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fma chain;
//     4 independent accumulators cover its 40-cycle dependent latency; the library's own kernel for these shapes uses the
//     same instruction).  The product is formed TRANSPOSED (the instruction's A operand is this kernel's B fragment), so a
//     lane ends with 4 consecutive output COLUMNS of one row: 16-byte stores, 16-byte gate loads.
//   * operands staged global -> registers -> LDS (global_load_dwordx4 + ds_write_b128, one register set = one 32-deep K chunk
//     in flight, two LDS stages), ONE raw s_barrier per chunk.  The pipeline is skewed by half a chunk: the MFMAs of a
//     chunk's second half are issued AFTER the barrier that publishes the next chunk, beside the LDS reads of its first half,
//     so neither the barrier nor the read latency leaves the matrix pipe without queued work.
//   * both operand layouts without a transpose pass: "KC" = the summation index contiguous in memory (x [B,K], W [N,K] of
//     the forward pass, dz as the A operand of dh): 128-byte LDS rows, 16-byte pieces XOR-swizzled by (row >> 1) & 7, one
//     conflict-free ds_read_b128 feeds 4 MFMA K-steps (the K order inside a 16-deep span is permuted identically for A and
//     B); "MC" = the output index contiguous (W of dh, dz^T and h of dW): 256-byte LDS rows with bit 4 of the column
//     XOR-ed with bit 2 of the K row, read with ds_read_b32 (lane = column), conflict-free.
//   * epilogues (template EPI): bias + ReLU / tanh (forward); activation derivative from the layer's forward output
//     (ReLU: h > 0, tanh: 1 - h^2) and the bias gradient's column sums per 64-row tile, in a fixed order (dh); sum of squares
//     of the tile for clip_by_global_norm (dW) — all deterministic: no atomics, partials finished by k_bias_finalize /
//     k_adam_norm in index order.
#pragma once

namespace mg {

constexpr int BM = 64, BN = 64, BK = 32, THREADS = 256;
constexpr int LDS_FLOATS = 2 * (2 * 64 * BK);      // the 64 x 64 form: two stages of an A and a B tile = 32 KB (+ 128 reduction words)

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_NONE = 0, EPI_BIAS_ACT = 1, EPI_GATE_COLSUM = 2, EPI_SQSUM = 3 };

struct Args {
  const float *A;   // KC: A[m * lda + k]   MC: A[k * lda + m]
  int64_t lda;
  const float *B;   // KC: B[n * ldb + k]   MC: B[k * ldb + n]
  int64_t ldb;
  float *C;         // C[m * ldc + n]
  int64_t ldc;
  int M, N, K;      // leading dimensions, N and (for a KC operand) K multiples of 4: 16-byte pieces; edge tiles are fine
  int act;          // 0 = ReLU, 1 = tanh (EPI_BIAS_ACT: the activation; EPI_GATE_COLSUM: its derivative from `gate`)
  const float *bias;   // EPI_BIAS_ACT: [N]
  const float *gate;   // EPI_GATE_COLSUM: [M][ldg] = the layer's forward output h
  int64_t ldg;
  float *colsum;       // EPI_GATE_COLSUM: [ceil(M / 64)][N] column sums of the STORED values per 64-row tile, or NULL
  float *sqsum;        // EPI_SQSUM: [tiles] sum of squares of the stored tile (tile index = tm * tiles_n + tn)
#ifdef MG_TIMING
  unsigned long long *dbg;   // timing build: per workgroup 8 words (shader cycles at 4 points, 100 MHz ticks at 4 points)
#endif
};
#ifdef MG_TIMING
#define MG_STAMP(k) do { if (threadIdx.x == 0 && G.dbg) { G.dbg[(size_t)bid * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                                                          G.dbg[(size_t)bid * 8 + 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MG_STAMP(k) do { } while (0)
#endif
#ifndef MG_EXP
#define MG_EXP 0   // timing experiments (scripts/micro/gemm64_test.hip): 1 = no global loads in the loop, 2 = no MFMA, 4 = no LDS writes,
                   // 8 = no barrier, 16 = no fragment reads (results are wrong in every one of them)
#endif

// workgroup -> tile: consecutive logical ids (= one XCD: blocks b and b + 8 share one) walk a strip of 4 row tiles column by
// column, so an XCD's share of a 16 x 16 tile grid is a 4 x 8 block: 1 MB of A rows + 2 MB of B rows in its 4 MB L2.  Speed only.
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

#define MG_SB() __builtin_amdgcn_sched_barrier(0)

template <int CTRL>
__device__ __forceinline__ float mg_dpp(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), on every lane of the row; fixed order
__device__ __forceinline__ float row16_sum(float v) {
  v += mg_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += mg_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += mg_dpp<0x124>(v);   // row_ror:4
  v += mg_dpp<0x128>(v);   // row_ror:8
  return v;
}

template <bool V>
struct BoolTag { static constexpr bool value = V; };
template <int V>
struct IntTag { static constexpr int value = V; };

// NB = 16-column blocks of B per wave: 2 = 64 x 64 tiles (one workgroup per CU at 1024 x 1024), 1 = 64 x 32 tiles (512 workgroups
// = two per CU, with their own barriers: one's barrier waits, prologue and epilogue sit under the other's MFMAs).
template <bool A_KC, bool B_KC, int EPI, int NB>
__device__ __forceinline__ void gemm_tile(const Args &G, float *lds, int bid, int nblk) {
  constexpr int BNT = 32 * NB;                       // tile columns
  constexpr int OPER_A = 64 * BK, OPER_B = BNT * BK;  // floats per operand tile
  constexpr int STAGE = OPER_A + OPER_B;              // A tile, then B tile
  constexpr int NP = 2 + NB;                          // 16-byte staging pieces per thread and chunk: 2 of A, NB of B
  constexpr int NU = 2 + NB;                          // fragment units per half: A block 0 / 1, B block 0 (/ 1)
  constexpr int HM = 8 * NB, PM = 2 * HM;             // MFMAs per half chunk / per chunk and wave
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (G.M + BM - 1) / BM, tiles_n = (G.N + BNT - 1) / BNT;
  if (bid >= tiles_m * tiles_n) return;
  int tm, tn;
  tile_of(bid, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
  (void)nblk;
  const int m0 = tm * BM, n0 = tn * BNT;
  const int nchunks = (G.K + BK - 1) / BK;

  // ---- staging: a chunk of the A tile is 512 pieces of 16 bytes (two per thread: j = 0, 1), of the B tile 256 NB (j = 2 ..).
  //   KC: piece q = row (q >> 3) of the tile, 16-byte piece (q & 7) of its 128 bytes of K  -> LDS row * 32 + ((p ^ swz(row)) << 2)
  //   MC: a K row is X = 64 (A) / 32 NB (B) floats = X / 4 pieces: piece q = K row q / (X / 4), piece q % (X / 4)
  //                                                                                      -> LDS k * X + ((p ^ (k & 4)) << 2)
  // Source address = descriptor base + a per-lane 32-bit byte offset fixed for the launch + a scalar offset that walks K.
  // (pieces outside the matrix — rows / columns of an edge tile — are CLAMPED to an address inside it: what they contribute
  //  lands in output rows / columns that are never stored; so the steady state has no predicated loads)
  uint32_t go[NP], gs[NP];      // byte offset of this thread's piece from the operand's chunk base; the part of it that selects the K index
  int lo[NP], kk[NP];           // LDS float offset inside a stage; K index of the piece within a chunk (for the K tail)
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const bool isB = j >= 2;
    const int q = tid + 256 * (isB ? j - 2 : j);
    const bool kc = isB ? B_KC : A_KC;
    const int x0 = isB ? n0 : m0, X = isB ? G.N : G.M, XT = isB ? BNT : 64;
    const int64_t ld = isB ? G.ldb : G.lda;
    if (kc) {
      const int r = q >> 3, p = q & 7;
      const int x = (x0 + r < X) ? x0 + r : X - 1;
      go[j] = (uint32_t)(((int64_t)x * ld + 4 * p) * 4);
      gs[j] = (uint32_t)(4 * p * 4);
      lo[j] = (isB ? OPER_A : 0) + r * 32 + ((p ^ ((r >> 1) & 7)) << 2);
      kk[j] = 4 * p;
    } else {
      const int ppr = XT / 4;   // pieces per K row
      const int kr = q / ppr, p = q % ppr;
      const int col = (x0 + 4 * p < X) ? x0 + 4 * p : 0;
      go[j] = (uint32_t)(((int64_t)kr * ld + col) * 4);
      gs[j] = (uint32_t)(((int64_t)kr * ld) * 4);
      lo[j] = (isB ? OPER_A : 0) + kr * XT + ((p ^ (kr & 4)) << 2);
      kk[j] = kr;
    }
  }
  const uint32_t stepa = (uint32_t)((A_KC ? (int64_t)BK : (int64_t)BK * G.lda) * 4), stepb = (uint32_t)((B_KC ? (int64_t)BK : (int64_t)BK * G.ldb) * 4);   // bytes
  const int kfull = G.K / BK;           // chunks [0, kfull) are whole; chunk kfull (if any) is partial: pieces beyond K become zeros
  f32x4 rg[NP];                         // the chunk in flight
  // Piece j of the chunk being requested -> registers, as a BUFFER load: address = descriptor base (4 SGPRs) + the lane's 32-bit
  // byte offset (a VGPR fixed for the launch) + a scalar offset that walks K — no per-lane address arithmetic in the loop.
  // (hipcc forms `uniform pointer + 32-bit lane offset` with a 64-bit VALU add per load, and between f32 MFMAs every VALU
  //  instruction costs the stream ~14 cycles and a global_load with a 64-bit VGPR address 23, against ~6 for a load whose base
  //  is scalar: scripts/micro/mfma_f32_issue.hip.  A builtin, not inline asm: hipcc then counts the loads — vmcnt in front of
  //  every LDS store of the steady state — and never copies a register whose load is still in flight.)
  // The matrices of the step are < 2 GB: 32-bit offsets; num_records covers the whole address range behind the base.
  const __amdgpu_buffer_rsrc_t srda = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.B), (short)0, 0x7FFFFFFF, 0x00020000);
  uint32_t soa = 0, sob = 0;            // scalar byte offsets of the chunk being requested
#define MG_GLOAD(dst, off, isb) dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((isb) ? srdb : srda, (int)(off), (int)((isb) ? sob : soa), 0))
  auto gload1_full = [&](int j) __attribute__((always_inline)) { MG_GLOAD(rg[j], go[j], j >= 2); };
  // a partial last chunk: pieces beyond K are fetched from the chunk's K index 0 (inside the matrix) and zeroed by lfix1
  auto gload1 = [&](int j, int c) __attribute__((always_inline)) {
    const uint32_t off = go[j] - ((c < kfull || kk[j] < G.K - c * BK) ? 0u : gs[j]);   // (arithmetic, not a select of two array elements: that sent both arrays to scratch)
    MG_GLOAD(rg[j], off, j >= 2);
  };
  auto lfix1 = [&](int j, int c) __attribute__((always_inline)) {      // (piece j of chunk c, before it goes to LDS)
    if (c >= kfull && kk[j] >= G.K - c * BK) rg[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  auto gadvance = [&]() __attribute__((always_inline)) { soa += stepa; sob += stepb; };
  auto lwrite1 = [&](int j, float *st) __attribute__((always_inline)) { *reinterpret_cast<f32x4 *>(st + lo[j]) = rg[j]; };

  // ---- fragments.  lane = (c, g): c = lane & 15 = row of the A block / column of the B block, g = lane >> 4 = K group:
  // in the 16-deep half h of a chunk the lane holds K indices 16 h + 4 g + s, s = 0..3 (one MFMA K-step each).
  const int wm = w >> 1, wn = w & 1, c16 = lane & 15, g = lane >> 4;
  // Fragment addresses (float offsets inside a stage), two registers per operand, so that every read of the loop is
  // `register + compile-time constant` (the stage too: the loop is unrolled by two) — no VALU instruction in the K loop:
  //   KC: fx[h] = block 0 of half h (the swizzle moves the half's piece by a lane-dependent amount); block 1: + 16 rows = + 512
  //   MC: fx[b] = block b of half 0 (bit 4 of the column is XOR-ed with the lane's K group); half 1: + 16 K rows; K-step s: + one row
  int fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = wm * 32 + c16, n = wn * 16 * NB + c16;
    fa[i] = A_KC ? r * 32 + (((4 * i + g) ^ ((r >> 1) & 7)) << 2) : (4 * g) * 64 + ((r + 16 * i) ^ ((g & 1) << 4));
    fb[i] = OPER_A + (B_KC ? n * 32 + (((4 * i + g) ^ ((n >> 1) & 7)) << 2) : (4 * g) * BNT + ((n + 16 * i) ^ ((g & 1) << 4)));
  }
  // unit u of a half's fragments: 0, 1 = A block 0 / 1, 2 (, 3) = B block 0 (/ 1) (one ds_read_b128 or four ds_read_b32 each)
  auto read_unit = [&](const float *st, int h, int u, float (&av)[2][4], float (&bv)[NB][4]) __attribute__((always_inline)) {
    if (u < 2) {
      const int b = u;
      if (A_KC) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(st + fa[h] + b * 512);
        av[b][0] = v.x; av[b][1] = v.y; av[b][2] = v.z; av[b][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++) av[b][s] = st[fa[b] + h * 1024 + s * 64];
      }
    } else {
      const int b = u - 2;
      if (B_KC) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(st + fb[h] + b * 512);
        bv[b][0] = v.x; bv[b][1] = v.y; bv[b][2] = v.z; bv[b][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++) bv[b][s] = st[fb[b] + h * 16 * BNT + s * BNT];
      }
    }
  };
  auto read_half = [&](const float *st, int h, float (&av)[2][4], float (&bv)[NB][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NU; u++) read_unit(st, h, u, av, bv);
  };

  f32x4 acc[2][NB];   // [A block][B block]; register i of lane (c, g) = C[row 16 bi + c][column 16 bj + 4 g + i] of the wave's part
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // The MFMA as inline asm with the accumulator tied in place ("+v": plain VGPRs — gfx950's register file is unified): as a
  // builtin, hipcc's register allocator renamed the accumulators at the loop's back edge and repaired that with 24
  // v_accvgpr moves per chunk, each waiting for the MFMA that produced its source.  volatile: the statements keep their order.
  // Hazards hipcc does not pad inside asm (cdna_hip_programming.md §5.7): an accumulator is re-used by every 2 NB-th MFMA
  // (>= 64 cycles after its producer: past the 40-cycle dependent latency); A / B operands come from LDS reads (waited for by
  // s_waitcnt, which hipcc does insert for asm operands); the epilogue's first VALU read of an accumulator sits behind explicit
  // s_nops (MG_ACC_FENCE).
#define MG_MF(av, bv, s, bi, bj)                                                                                      \
  if (!(MG_EXP & 2)) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[bi][bj]) : "v"(bv[bj][s]), "v"(av[bi][s])); \
  else acc[bi][bj][s] += av[bi][s] * bv[bj][s]

  float a0[2][4], b0[NB][4], a1[2][4], b1[NB][4];
  MG_STAMP(0);
#pragma unroll
  for (int j = 0; j < NP; j++) gload1(j, 0);
  gadvance();
#pragma unroll
  for (int j = 0; j < NP; j++) { lfix1(j, 0); lwrite1(j, lds); }
  if (nchunks > 1) {
#pragma unroll
    for (int j = 0; j < NP; j++) gload1(j, 1);
    gadvance();
  }
  // the epilogue's operands (bias / the layer's forward output for the activation derivative) are requested HERE, behind the
  // first two chunks: they land during the first phases, the epilogue finds them in registers (loads cannot sink below the
  // loop: its barriers are compiler memory barriers).  Rows / columns outside the matrix read a clamped address.
  f32x4 ebias[NB], egate[2][NB];
#pragma unroll
  for (int bj = 0; bj < NB; bj++) {
    const int n = n0 + 16 * NB * wn + 16 * bj + 4 * g, nc = n < G.N ? n : 0;
    if (EPI == EPI_BIAS_ACT) ebias[bj] = *reinterpret_cast<const f32x4 *>(G.bias + nc);
#pragma unroll
    for (int bi = 0; bi < 2; bi++) {
      const int m = m0 + 32 * wm + 16 * bi + c16, mc = m < G.M ? m : G.M - 1;
      if (EPI == EPI_GATE_COLSUM) egate[bi][bj] = *reinterpret_cast<const f32x4 *>(G.gate + (int64_t)mc * G.ldg + nc);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_half(lds, 0, a0, b0);
  if (MG_EXP & 16) read_half(lds, 1, a1, b1);
  MG_STAMP(1);
  // ---- the K loop.  Phase c = the PM MFMAs of chunk c with a slot behind each (the slot table is in `phase`).  Piece j of the
  // chunk in flight goes to LDS and is re-requested for the chunk after it in ONE slot: the wave's outstanding loads are then,
  // oldest first, the later pieces of chunk c + 1 and the earlier ones of chunk c + 2 (hipcc's counted wait in front of the store
  // covers exactly piece j).  The four waves run the same stream in step (one barrier per chunk); staggered by W, (NB = 2) no
  // two of them push a 16-byte store through the CU's VGPR -> LDS path in the same slot.
  // What an instruction between two f32 MFMAs of one wave costs the MFMA stream (scripts/micro/mfma_f32_issue.hip, cycles):
  // ds_read_b128 1-4, ds_write_b128 3 (one per 4 MFMAs), global_load with an SGPR base 6, with a 64-bit VGPR address 23,
  // ANY VALU instruction 14, SALU / s_waitcnt 0-1.  Hence: addresses are registers + immediates (the loop is unrolled by two so
  // that the stage is a constant), bases walk K in SGPRs.
  // FULL phases (chunk c + 1 exists and chunk c + 2 is a whole chunk) are branch-free; the last phases take the conditional form.
  auto phase = [&](auto wtag, auto full_tag, auto par_tag, int c) __attribute__((always_inline)) {
    constexpr int W = decltype(wtag)::value, PAR = decltype(par_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    float *st = lds + PAR * STAGE, *sn = lds + (PAR ^ 1) * STAGE;
    const bool nxt = FULL || c + 1 < nchunks, nxt2 = FULL || c + 2 < nchunks;
    // slot i (behind MFMA i of the phase; the first HM MFMAs multiply the chunk's first half from a0 / b0, the others its second
    // half from a1 / b1):                                                NB = 2 (32 MFMAs)            NB = 1 (16 MFMAs)
    //    the second half's fragments (units 0 .. NU-1) <- this stage     slots 0..3                   0..2
    //    wave W's piece j of chunk c + 1 -> the other stage, then its    4 + 4 j + W  (4..19)         3 + 2 j + (W & 1)  (3..8; two
    //      piece j of chunk c + 2 requested into the same registers                                   waves per slot)
    //    every LDS operation of the wave done, barrier (publishes c + 1) 23                           10
    //    the next phase's first-half fragments <- the other stage        24..27                       11..13
    // (the barrier sits LATE in the phase: the LDS stores get slots of their own, no wave stores while the four of them read,
    //  and the last store of a wave is a few MFMAs old when it waits for it)
    constexpr int R0 = 0, W0 = NU, BAR = (NB == 2) ? 23 : 10, R1 = BAR + 1;
    auto slot = [&](int i) __attribute__((always_inline)) {   // (i is a literal at every call: the conditions fold)
      if (i >= R0 && i < R0 + NU && !(MG_EXP & 16)) read_unit(st, 1, i - R0, a1, b1);
      int j = -1;
      if (NB == 2) { if (i >= W0 && i < W0 + 4 * NP && ((i - W0) & 3) == W) j = (i - W0) >> 2; }
      else { if (i >= W0 && i < W0 + 2 * NP && ((i - W0) & 1) == (W & 1)) j = (i - W0) >> 1; }
      if (j >= 0) {
        if (nxt && !(MG_EXP & 4)) {
          if (!FULL) lfix1(j, c + 1);
          lwrite1(j, sn);
        }
        if (!(MG_EXP & 1)) {
          if (FULL) gload1_full(j);
          else if (nxt2) gload1(j, c + 2);
        }
      }
      if (i == BAR) {
        if (FULL || nxt2) gadvance();
        if (!(MG_EXP & 8)) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      if (i >= R1 && i < R1 + NU && nxt && !(MG_EXP & 16)) read_unit(sn, 0, i - R1, a0, b0);
    };
    // MFMA i of the phase: half i / HM; within it K-step s, A block bi, B block bj in the order (s, bi, bj)
#define MG_STEP(i)                                                                                                        \
    if ((i) < PM) {                                                                                                       \
      constexpr int k_ = (i) % HM, s_ = k_ / (2 * NB), bi_ = (k_ / NB) % 2, bj_ = k_ % NB;                                 \
      MG_SB();                                                                                                            \
      if ((i) < HM) { MG_MF(a0, b0, s_, bi_, bj_); } else { MG_MF(a1, b1, s_, bi_, bj_); }                                 \
      MG_SB();                                                                                                            \
      slot(i);                                                                                                            \
    }
    MG_STEP(0) MG_STEP(1) MG_STEP(2) MG_STEP(3) MG_STEP(4) MG_STEP(5) MG_STEP(6) MG_STEP(7)
    MG_STEP(8) MG_STEP(9) MG_STEP(10) MG_STEP(11) MG_STEP(12) MG_STEP(13) MG_STEP(14) MG_STEP(15)
    MG_STEP(16) MG_STEP(17) MG_STEP(18) MG_STEP(19) MG_STEP(20) MG_STEP(21) MG_STEP(22) MG_STEP(23)
    MG_STEP(24) MG_STEP(25) MG_STEP(26) MG_STEP(27) MG_STEP(28) MG_STEP(29) MG_STEP(30) MG_STEP(31)
    MG_SB();
#undef MG_STEP
  };
  auto k_loop = [&](auto wtag) __attribute__((always_inline)) {
    int c = 0;
    for (; c + 3 < kfull; c += 2) {   // both phases of the pair are FULL
      phase(wtag, BoolTag<true>{}, IntTag<0>{}, c);
      phase(wtag, BoolTag<true>{}, IntTag<1>{}, c + 1);
    }
    for (; c < nchunks; c++) {
      if (c & 1) phase(wtag, BoolTag<false>{}, IntTag<1>{}, c);
      else phase(wtag, BoolTag<false>{}, IntTag<0>{}, c);
    }
  };
  // (four copies of the loop, one per wave: the stagger is static, every copy is straight-line code)
  switch (w) {
    case 0: k_loop(IntTag<0>{}); break;
    case 1: k_loop(IntTag<1>{}); break;
    case 2: k_loop(IntTag<2>{}); break;
    default: k_loop(IntTag<3>{}); break;
  }
  // (hipcc pads nothing behind inline asm: the accumulators' first VALU read sits behind these)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[i][j]));
  MG_STAMP(2);
#undef MG_MF
#undef MG_GLOAD

  // ---- epilogue: lane (c, g) holds, per (bi, bj), C[m = m0 + 32 wm + 16 bi + c][n = n0 + 16 NB wn + 16 bj + 4 g + 0..3]
  float csum[NB][4];
  float sq = 0.0f;
#pragma unroll
  for (int bj = 0; bj < NB; bj++)
#pragma unroll
    for (int i = 0; i < 4; i++) csum[bj][i] = 0.0f;
  const bool relu = G.act == 0;   // (wave-uniform: the tanh forms are whole separate loops, no per-element branch)
#pragma unroll
  for (int bi = 0; bi < 2; bi++) {
    const int m = m0 + 32 * wm + 16 * bi + c16;
#pragma unroll
    for (int bj = 0; bj < NB; bj++) {
      const int n = n0 + 16 * NB * wn + 16 * bj + 4 * g;
      const bool ok = m < G.M && n < G.N;   // (N is a multiple of 4: the lane's four columns are inside or outside together)
      f32x4 v = acc[bi][bj];
      if (EPI == EPI_BIAS_ACT) {
        const f32x4 bb = ebias[bj];
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i] + bb[i], 0.0f);
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = tanhf(v[i] + bb[i]);
        }
      }
      if (EPI == EPI_GATE_COLSUM) {
        const f32x4 hh = egate[bi][bj];
        if (relu) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = hh[i] > 0.0f ? v[i] : 0.0f;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = v[i] * (1.0f - hh[i] * hh[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) csum[bj][i] += ok ? v[i] : 0.0f;
      }
      if (EPI == EPI_SQSUM) {
#pragma unroll
        for (int i = 0; i < 4; i++) sq += ok ? v[i] * v[i] : 0.0f;
      }
      if (ok) *reinterpret_cast<f32x4 *>(G.C + (int64_t)m * G.ldc + n) = v;
    }
  }
  float *red = lds + 2 * STAGE;   // 128 floats behind the stages: no wave can still be reading them
  if (EPI == EPI_GATE_COLSUM && G.colsum != nullptr) {
    // column sums of the tile's 64 rows, fixed order: block 0 + block 1 (above), the 16 rows of a block by DPP row operations
    // (quad_perm, quad_perm, row_ror:4, row_ror:8: every lane ends with its 16-lane row's sum), wave wm = 0 + wave wm = 1 through LDS
#pragma unroll
    for (int bj = 0; bj < NB; bj++)
#pragma unroll
      for (int i = 0; i < 4; i++) csum[bj][i] = row16_sum(csum[bj][i]);
    if (c16 == 0) {
#pragma unroll
      for (int bj = 0; bj < NB; bj++)
        *reinterpret_cast<f32x4 *>(red + wm * BNT + wn * 16 * NB + 16 * bj + 4 * g) = f32x4{csum[bj][0], csum[bj][1], csum[bj][2], csum[bj][3]};
    }
    __syncthreads();
    if (tid < BNT && n0 + tid < G.N) G.colsum[(int64_t)tm * G.N + n0 + tid] = red[tid] + red[BNT + tid];
  }
  if (EPI == EPI_SQSUM && G.sqsum != nullptr) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    if (tid == 0) G.sqsum[tm * tiles_n + tn] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  MG_STAMP(3);
}

// LDS floats of a workgroup: two stages + the epilogue's reduction words
template <int NB>
constexpr int lds_floats() { return 2 * (64 * BK + 32 * NB * BK) + 128; }

template <bool A_KC, bool B_KC, int EPI, int NB>
__global__ __launch_bounds__(THREADS) void k_gemm64n(Args G) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<NB>()];
  gemm_tile<A_KC, B_KC, EPI, NB>(G, lds, (int)blockIdx.x, (int)gridDim.x);
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(THREADS) void k_gemm64(Args G) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<2>()];
  gemm_tile<A_KC, B_KC, EPI, 2>(G, lds, (int)blockIdx.x, (int)gridDim.x);
}

// Several products of one layout as ONE launch (the FAIR step's eleven weight gradients: as launches of their own they are
// 0.08-0.5 GFLOP each and cost a launch's latency apiece): workgroup b works on tile b - first[p] of problem p.
constexpr int GROUP_MAX = 16;
struct GroupArgs {
  int n;
  int first[GROUP_MAX + 1];   // prefix sums of the problems' tile counts
  Args g[GROUP_MAX];
};
template <bool A_KC, bool B_KC, int NB>
__global__ __launch_bounds__(THREADS) void k_gemm_group(GroupArgs GA) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<NB>()];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < GA.n && b >= GA.first[p + 1]) p++;
  gemm_tile<A_KC, B_KC, EPI_NONE, NB>(GA.g[p], lds, b - GA.first[p], GA.first[p + 1] - GA.first[p]);
}

#undef MG_SB
}  // namespace mg
