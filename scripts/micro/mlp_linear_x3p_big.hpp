// mlp_linear_x3p_big.hpp — EXPERIMENT (round 6, not adopted): csrc/mlp_linear_x3p.hpp's layer on a 256 x 128 tile with 64 x 64 wave tiles
// (a third less LDS traffic per MFMA, a quarter less operand traffic per output) — at the price of 16-deep K chunks (32-byte LDS rows: four
// stages of 36 KB), i.e. 32-byte global segments per DMA lane pair.  Correct (the library's tests passed on it: r06ah), but 97 us per
// 8192 x 1024 x 1024 against the 128 x 128 kernel's 84: quarter-line global accesses cost more than the tile saves.  The 128 x 128 kernel's
// own accounting (scripts/micro/lx3_exp.hip, r06ai_lx3_exp.txt, constant operands): launch + prologue + epilogue 16.7 us (50 MB of planes
// out), the MFMA stream alone 43 us, the DMA stream alone 47 us (17 TB/s out of the L2s), fragment reads 1 us, barriers 2 us, all together
// 53 us — and 84 with random operands (the clock the chip holds under toggling MFMA inputs).
// To build it again: paste `namespace big` into csrc/mlp_linear_x3p.hpp's namespace lx3 and launch lx3::big::k_linear_x3p_big<NPX> with
// ceil(m / 256) * (n / 128) workgroups.
#pragma once

// ---- the same layer on a 256 x 128 tile (8192 x 1024 outputs = 256 tiles = ONE per CU).  The 128 x 128 kernel above is bound by LDS
// bandwidth, not by the matrix pipe: a 64 x 32 wave tile reads 9 fragments (9 KB) per 12 MFMAs = 96 B / clock over the CU at the MFMA
// rate, + 31 B / clock of DMA writes, of the 128 the LDS moves — it runs where brl_mlp_gemm_x3 runs (84 vs 87 us per 8192 x 1024 x 1024:
// scripts/x3p_layer_probe.py).  Here: waves as 4 (M) x 2 (N), wave tile 64 x 64 = FOUR blocks on 12 fragments per 24 MFMAs = 64 B / clock
// (+ 23 of DMA); 16-deep K chunks (one MFMA K step: 32-byte LDS rows, two 16-byte pieces swizzled by bit 3 of the row) so that four
// stages of 36 KB fit: three chunks in flight across one barrier per chunk.  The output planes leave through LDS in two halves of 128 rows.
namespace big {
constexpr int BM = 256, BN = 128, BK = 16, STAGES = 4;
constexpr int APLANE = BM * 32, BPLANE = BN * 32;        // 8 KB, 4 KB
constexpr int STAGE_BYTES = 3 * APLANE + 3 * BPLANE;     // 36 KB
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;          // 144 KB
static_assert(3 * 128 * C_ROW_BYTES <= LDS_BYTES, "half of the output planes reuses the stages");

// workgroup -> logical id: blocks b, b + 8, .. share an XCD; give them CONSECUTIVE logical ids (bijective for any grid size)
__device__ __forceinline__ int xcd_logical_id(int b, int nblk) {
  const int q = nblk / 8, r = nblk % 8, xcd = b % 8;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

template <int NPX>
__global__ __launch_bounds__(THREADS) void k_linear_x3p_big(Args G) {
  static_assert(NPX == 1 || NPX == 3, "x: one plane (exact in bf16) or three");
  constexpr int NP = NPX == 3 ? 6 : 3;         // products per block and K step
  constexpr int NMF = 4 * NP;                  // MFMAs per chunk and wave: 4 blocks x NP
  constexpr int NFR = 2 * NPX + 6;             // fragments per chunk: 2 x blocks x their planes, 2 W blocks x 3
  constexpr int NIA = NPX, NIB_LO = 2, NIB_HI = 1;   // DMA instructions per wave and chunk: its 32 rows of every x plane; W: 12 over 8 waves
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = G.N / BN;
  // the column tiles walk fastest: an XCD's 32 workgroups = 4 row tiles x all 8 column tiles at N = 1024 (its L2 holds W once)
  const int L = xcd_logical_id((int)blockIdx.x, (int)gridDim.x);
  const int tm = L / tiles_n, tn = L - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nchunks = G.K / BK;
  const bool lo4 = w < 4;                      // waves 0..3 issue two of W's instructions, waves 4..7 one

  // ---- staging: a DMA instruction = 32 rows x 32 B: lane -> row (lane >> 1), LDS slot lane & 1 <- the row's piece (lane & 1) ^ ((row >> 3) & 1)
  uint32_t offx, offw0, offw1;
  int wpl0, wpl1;                              // the W planes of this wave's instructions e = w, w + 8 (plane e / 4, row group e % 4)
  {
    const int r = lane >> 1, pc = (lane & 1) ^ ((r >> 3) & 1);
    const int xr = 32 * w + r, mr = (m0 + xr < G.M) ? m0 + xr : G.M - 1;
    offx = (uint32_t)(((int64_t)mr * G.ldx + 8 * pc) * 2);
    const int e0 = w, e1 = w + 8;
    wpl0 = e0 >> 2; wpl1 = e1 >> 2;
    offw0 = (uint32_t)(((int64_t)(n0 + 32 * (e0 & 3) + r) * G.ldw + 8 * pc) * 2);
    offw1 = (uint32_t)(((int64_t)(n0 + 32 * (e1 & 3) + r) * G.ldw + 8 * pc) * 2);
  }
  int kc = 0;
  // instruction j of this wave's chunk: j < NPX: x plane j; j == NPX: W instruction e = w; j == NPX + 1 (waves 0..3): e = w + 8
  auto stage_one = [&](unsigned char *st, int j) __attribute__((always_inline)) {
    const char *base;
    uint32_t o;
    unsigned char *dst;
    if (j < NPX) {
      base = reinterpret_cast<const char *>(G.x + (int64_t)j * G.sx);
      o = offx;
      dst = st + j * APLANE + w * 1024;
    } else if (j == NPX) {
      base = reinterpret_cast<const char *>(G.w + (int64_t)wpl0 * G.sw);
      o = offw0;
      dst = st + NPX * APLANE + wpl0 * BPLANE + (w & 3) * 1024;
    } else {
      base = reinterpret_cast<const char *>(G.w + (int64_t)wpl1 * G.sw);
      o = offw1;
      dst = st + NPX * APLANE + wpl1 * BPLANE + ((w + 8) & 3) * 1024;
    }
    o += (uint32_t)kc * (BK * 2);
    asm volatile("" : "+v"(o));
    glds16(base + o, dst);
  };
  auto stage_chunk = [&](unsigned char *st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j <= NPX; j++) stage_one(st, j);
    if (lo4) stage_one(st, NPX + 1);
    kc++;
  };
  // wait until at most `left` of this wave's chunks are still in flight
  auto wait_chunks = [&](int left) __attribute__((always_inline)) {
#define LX3_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")
    if (left <= 0) LX3_WAIT(0);
    else if (NPX == 3) {      // 5 instructions per chunk (waves 0..3) / 4
      if (lo4) { if (left == 1) LX3_WAIT(5); else if (left == 2) LX3_WAIT(10); else LX3_WAIT(15); }
      else { if (left == 1) LX3_WAIT(4); else if (left == 2) LX3_WAIT(8); else LX3_WAIT(12); }
    } else {                  // 3 / 2
      if (lo4) { if (left == 1) LX3_WAIT(3); else if (left == 2) LX3_WAIT(6); else LX3_WAIT(9); }
      else { if (left == 1) LX3_WAIT(2); else if (left == 2) LX3_WAIT(4); else LX3_WAIT(6); }
    }
#undef LX3_WAIT
  };

  // ---- fragments: lane (r, hh): the 16-byte piece hh of operand row r
  const int wm = w >> 1, wn = w & 1, r32 = lane & 31, hh = lane >> 5;
  const int pcs = (hh ^ ((r32 >> 3) & 1)) << 4;
  const int fa0 = (64 * wm + r32) * 32 + pcs, fb0 = NPX * APLANE + (64 * wn + r32) * 32 + pcs;
  // fragment u: u < 2 NPX: x block u / NPX, plane u % NPX; else v = u - 2 NPX: W block v / 3, plane v % 3
  auto read_frag = [&](const unsigned char *st, int u) __attribute__((always_inline)) -> bf16x8 {
    int off;
    if (u < 2 * NPX) off = (u % NPX) * APLANE + (u / NPX) * 1024 + fa0;
    else off = ((u - 2 * NPX) % 3) * BPLANE + ((u - 2 * NPX) / 3) * 1024 + fb0;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(st + off));
  };
  f32x16 acc[4][2];     // [block 2 bm + bn][class: 0 = hi.hi, 1 = the smaller products]
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[b][c][e] = 0.0f;
  // MFMA t of a chunk: product t >> 2 (small -> large), block t & 3
  auto mf = [&](const bf16x8 (&f)[NFR], int t) __attribute__((always_inline)) {
    const int p = t >> 2, blk = t & 3, bm = blk >> 1, bn = blk & 1;
    int px, pw;
    if (NPX == 3) {
      px = (p == 0 || p == 3 || p == 5) ? 0 : (p == 2 || p == 4) ? 1 : 2;
      pw = (p == 1 || p == 4 || p == 5) ? 0 : (p == 2 || p == 3) ? 1 : 2;
    } else {
      px = 0;
      pw = 2 - p;
    }
    const int cls = (px == 0 && pw == 0) ? 0 : 1;
    acc[blk][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 * NPX + 3 * bn + pw], f[bm * NPX + px], acc[blk][cls], 0, 0, 0);
  };
  // the order in which a chunk's MFMAs first need its fragments
  auto rorder = [&](int q) __attribute__((always_inline)) -> int {
    if (NPX == 3) {
      constexpr int O[12] = {8, 0, 11, 3, 6, 2, 9, 5, 7, 1, 10, 4};     // W lo 0, x hi 0, W lo 1, x hi 1; W hi, x lo; W mid, x mid
      return O[q];
    }
    constexpr int O[8] = {4, 0, 7, 1, 3, 6, 2, 5};                      // W lo 0, x 0, W lo 1, x 1; W mid 0 / 1; W hi 0 / 1
    return O[q];
  };

  // ---- prologue: every stage requested (chunks 0..3), chunk 0 awaited
  for (int c = 0; c < STAGES && c < nchunks; c++) stage_chunk(lds + c * STAGE_BYTES);
  wait_chunks((nchunks < STAGES ? nchunks : STAGES) - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 f0[NFR], f1[NFR];
#pragma unroll
  for (int q = 0; q < NFR; q++) f0[rorder(q)] = read_frag(lds, rorder(q));

  // ---- phase c: the chunk's MFMAs from registers; behind the first: this wave's DMA pieces of chunk c + 1 have landed, barrier (they have
  // for everybody; nobody reads chunk c's stage any more); then the DMA instructions of chunk c + 4 (into chunk c's stage), one per gap,
  // and the fragment reads of chunk c + 1.  In flight across the barrier: chunks c + 2 and c + 3.
  auto phase = [&](auto full_tag, const bf16x8 (&fu)[NFR], bf16x8 (&fn)[NFR], int c, int stage) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool next = FULL || c + 1 < nchunks;
    const bool dma = FULL || c + 4 < nchunks;
    unsigned char *st = lds + stage * STAGE_BYTES;
    const unsigned char *sn = lds + ((stage + 1) & 3) * STAGE_BYTES;
    __builtin_amdgcn_sched_barrier(0);
    mf(fu, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (next) {
      if (FULL) wait_chunks(2);
      else wait_chunks((nchunks - 1 - (c + 1) < 2) ? nchunks - 1 - (c + 1) : 2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < NMF; t++) {
      mf(fu, t);
      __builtin_amdgcn_sched_barrier(0);
      if (dma) {
        if (t - 1 <= NPX) stage_one(st, t - 1);
        else if (t - 1 == NPX + 1 && lo4) stage_one(st, NPX + 1);
      }
      if (t - 1 < NFR && next) fn[rorder(t - 1 < NFR ? t - 1 : 0)] = read_frag(sn, rorder(t - 1 < NFR ? t - 1 : 0));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dma) kc++;
  };
  static_assert(NFR <= NMF - 1 && NPX + 2 <= NMF - 1, "a fragment read and a DMA instruction per gap");
  {
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    const int nfull = nchunks - 4;     // phases c < nfull: chunk c + 4 exists
    int c = 0, stage = 0;
    for (; c + 1 < nfull; c += 2) {
      phase(T{}, f0, f1, c, stage); stage = (stage + 1) & 3;
      phase(T{}, f1, f0, c + 1, stage); stage = (stage + 1) & 3;
    }
    for (; c + 1 < nchunks; c += 2) {
      phase(F{}, f0, f1, c, stage); stage = (stage + 1) & 3;
      phase(F{}, f1, f0, c + 1, stage); stage = (stage + 1) & 3;
    }
    if (c < nchunks) phase(F{}, f0, f1, c, stage);
  }

  // ---- epilogue: lane holds, per block (bm, bn), row m0 + 64 wm + 32 bm + r32, columns n0 + 64 wn + 32 bn + 8 g + 4 hh + (0..3).  The
  // planes leave through LDS in two halves (bm = 0, 1: tile rows 64 wm + 32 bm + 0..31 -> LDS row 32 wm + r32)
  __syncthreads();
  const float floor_v = G.relu ? 0.0f : -__builtin_inff();
#pragma unroll
  for (int bm = 0; bm < 2; bm++) {
    const int row = 64 * wm + 32 * bm + r32, em = m0 + row;
#pragma unroll
    for (int bn = 0; bn < 2; bn++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int col = 64 * wn + 32 * bn + 8 * g + 4 * hh;
        const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(G.bias + n0 + col);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = fmaxf((acc[2 * bm + bn][1][4 * g + e] + acc[2 * bm + bn][0][4 * g + e]) + bias4[e], floor_v);
        if (G.y != nullptr && em < G.M) *reinterpret_cast<f32x4 *>(G.y + (int64_t)em * G.ldy + n0 + col) = o;
        if (G.yp != nullptr) {
          unsigned h[4], m[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; e++) split3(o[e], h[e], m[e], l[e]);
          unsigned char *p = lds + (32 * wm + r32) * C_ROW_BYTES + col * 2;
          *reinterpret_cast<u32x2 *>(p) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
          *reinterpret_cast<u32x2 *>(p + 128 * C_ROW_BYTES) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
          *reinterpret_cast<u32x2 *>(p + 2 * 128 * C_ROW_BYTES) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
        }
      }
    if (G.yp != nullptr) {
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (3 * 128 * 16) / THREADS; it++) {
        const int idx = it * THREADS + tid, pl = idx >> 11, lr = (idx >> 4) & 127, ch = idx & 15;
        const int trow = 64 * (lr >> 5) + 32 * bm + (lr & 31);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + (pl * 128 + lr) * C_ROW_BYTES + ch * 16);
        if (m0 + trow < G.M) *reinterpret_cast<u32x4 *>(G.yp + (int64_t)pl * G.syp + (int64_t)(m0 + trow) * G.ldyp + n0 + ch * 8) = v;
      }
      if (bm == 0) __syncthreads();
    }
  }
}
}  // namespace big

