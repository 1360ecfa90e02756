// fair_chain.hpp — one PPO minibatch step of the "FAIR" network (src/models.py:34-69: eleven 200-wide hk.Linear's in four residual
// blocks, the observation concatenated back in front of the seventh, actor / critic heads on the last block's output) WITHOUT its
// ~55 small launches: forward, `_loss_fn` (src/update.py:90-167) and the whole backward chain of 16 samples by one workgroup.
//
// Why one workgroup per 16 rows: every product of this network is 16 x 200 x 200 per 16 samples — rows are independent all the way
// from the observation to d(loss)/d(pre-activations); only the WEIGHT gradients sum over the minibatch.  As launches (library
// GEMMs + elementwise kernels) the step was bound by launch count (~70 x 4.7 us, profiles/r05/r05f_fair_step_timeline.txt).  Here
// the activations of a row block never leave the CU (LDS), the 1.8 MB of weights stream from L2 (every workgroup reads all of
// them: 64 x 24 x 160 KB per step, L2-resident), products are v_mfma_f32_16x16x4_f32 tiles (13 column tiles of 16 over 8 waves), and
// what the weight-gradient products need afterwards — every layer's input and pre-activation gradient — is written out once
// (k_fair_chain<true>; <false> = the forward half alone for rollouts / evaluators: brl_fair_forward):
//   inp [9][B][200]  inputs of the square layers 1,2,3,4,5,7,8,9,10      dzs [9][B][200]  their pre-activation gradients
//   cat6 [B][680] = [z5 | x0]   dz6, dz0, x4 [B][200]   dheads [B][40]   gates [4][B][200] = h2, h4, h8, h10 (activation outputs
//   the backward re-reads)   tiles [11][B/16][200] + [B/16][39]: column sums of dz_l (l = 0..10) and of dheads per workgroup = the
//   bias gradients' partials (finished by k_bias_finalize), partials [B/16][8] + gram_partials [B/16][1444]: the statistics.
// MFMA operand order: the weight fragment is the instruction's A operand, the sample fragment its B operand, so that lane
// (c = lane & 15, g = lane >> 4) ends with row c, columns 4 g .. 4 g + 3 of the 16 x 16 tile: float4 epilogues, float4 stores.
// Both fragments hold K indices 16 j + 4 g + s (s = register component) of chunk j — any order is a valid sum.
// Included by brl_ppo.hip after ppo_heads.hpp (wave_sum_f, dpp_move_f, PpoArgs, ppo_loss_sample, BiasSegs).
#pragma once

#ifndef FAIR_EXP
#define FAIR_EXP 0   // experiment builds (scripts/fair_chain_probe.py): 1 no MFMAs, 2 no weight loads, 4 no LDS fragment reads,
                     // 8 no global stores of activations / gradients, 16 filler jobs load real weights, 32 / 64 no backward / forward
                     // weight loads, 256 every weight load out of range (issued, answered with 0 without a memory access)
#endif
#ifdef FAIR_TIMING   // scripts/fair_chain_probe.py --stamps: shader-clock stamps of wave 0 / lane 0 behind every phase
__device__ unsigned long long *g_fair_dbg = nullptr;
#define FAIR_STAMP(k) do { if (threadIdx.x == 0 && g_fair_dbg) g_fair_dbg[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FAIR_STAMP(k) do { } while (0)
#endif
namespace fair {
constexpr int H = 200, OBS = 480, CAT = 680;
constexpr int R = 16;               // rows (samples) per workgroup
constexpr int NW = 8;               // waves per workgroup
constexpr int NT = 13;              // column tiles of 16 (208 >= 200)
constexpr int LDA = 208;            // LDS row stride of an activation buffer (floats) = 13 groups of four 16-byte pieces; columns 200..207
                                    // stay ZERO (the K tail of a product reads them).  Inside a group the piece a lane of K group g
                                    // touches is g ^ f(c >> 2), f = (0, 2, 3, 1): a ds_read_b128 is served in the lane groups
                                    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): sixteen lanes of mixed g — with the round-5
                                    // stride of 212 two of them shared a bank group in most (SQ_LDS_BANK_CONFLICT 45 % of
                                    // SQ_LDS_IDX_ACTIVE, profiles/r05/r05x_fair_step_pmc.txt); this map gives each its own.
constexpr int LDX = 484;            // ... of the observation rows
constexpr int LDH = 48;             // ... of the heads / d(heads) rows (39 used, the rest zero)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Net {
  const float *w[11], *b[11];       // nn.Linear layout [out, in]: w[0] [200,480], w[6] [200,680], the others [200,200]
  const float *wh, *bh;             // [39,200] = actor rows, then the critic row; [39]
};
struct Bufs {
  float *inp, *dzs, *gates, *cat6, *x4, *dz0, *dz6, *dheads, *tiles, *partials, *gram_partials;
};
struct Args {
  Net net;
  Bufs o;
  const float *x0;                  // [B,480]
  PpoArgs P;                        // the loss inputs (logits / value / outputs unused: they live in LDS here)
  int act;                          // 0 ReLU, 1 tanh
  int reward_scaling;
  // inference (k_fair_chain<false>): rows of x0, where the heads go; Bufs / P unused
  int64_t nrows;
  float *logits_out, *value_out;    // [nrows,38], [nrows]
};

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
  if (act == 0) return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
  return f32x4{tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)};
}
// d * act'(.) from the activation's OUTPUT h: ReLU h > 0, tanh 1 - h^2 (src/models.py:16)
__device__ __forceinline__ f32x4 dact4(f32x4 d, f32x4 h, int act) {
  if (act == 0) return f32x4{h.x > 0.f ? d.x : 0.f, h.y > 0.f ? d.y : 0.f, h.z > 0.f ? d.z : 0.f, h.w > 0.f ? d.w : 0.f};
  return f32x4{d.x * (1.f - h.x * h.x), d.y * (1.f - h.y * h.y), d.z * (1.f - h.z * h.z), d.w * (1.f - h.w * h.w)};
}
// the sum over the 16 lanes of a DPP row (= the tile's 16 rows of one column), on every lane of the row; fixed order
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move_f<0xB1>(0.0f, v);
  v += dpp_move_f<0x4E>(0.0f, v);
  v += dpp_move_f<0x124>(0.0f, v);
  v += dpp_move_f<0x128>(0.0f, v);
  return v;
}
__device__ __forceinline__ f32x4 row16_sum4(f32x4 v) { return f32x4{row16_sum(v.x), row16_sum(v.y), row16_sum(v.z), row16_sum(v.w)}; }

// ---- products: a "job" = one 16-column output tile times <= 240 K indices, its weight fragments held in registers ---------------
// Weight fragments are requested one job AHEAD of the MFMAs that use them (two register sets in alternation, also across the
// workgroup barriers between layers: weights do not depend on activations), one chunk between every four MFMAs of the current
// job (slot_run below).  History of the launch at minibatch 1024 (profiles/r05/r05_experiments.txt section 15): loads and MFMAs
// of a tile issued together 163 us, all loads of a tile first 141, a job ahead but in one burst 140, interleaved 114, the LDS reads a chunk ahead 108.
constexpr int FCH = 15;                 // fragments per job: 15 x 16 K (13 for the 200-wide products: the last one half empty)
struct Frag { f32x4 w[FCH]; };
struct Job {                            // wave-uniform
  const float *W;                       // NN == 0: W[n][k] (forward; offset to the job's first k), 1: W[k][n] (backward)
  int ldw, nch, nn, n0, nmax;           // nch: 13 (K = 200, tail of 8) or 15 (K = 240, exact)
  int real;                             // 0: a filler job (a wave without a tile in this slot): its loads touch no memory (they return 0)
};

// (buffer loads: descriptor + 32-bit lane offset + scalar / immediate offset — no VALU instruction per load and a quarter of the
//  issue cost of a 64-bit VGPR address; profiles/r04: every VALU instruction between f32 MFMAs costs ~14 cycles of issue)
// a job's addressing, formed once: descriptor + this lane's byte offsets
struct JobAddr {
  __amdgpu_buffer_rsrc_t rs;
  int voff, vtail;          // byte offset of the lane's piece from the job's base; the same for chunk 12 (the K tail: clamped lanes)
  int nn, wide, nch;        // backward (column access: four dwords per chunk); ldw == 680; chunks
};
__device__ __forceinline__ JobAddr job_addr(const Job &J, const int c, const int g) {
  JobAddr a;
  const int n = (J.n0 + c < J.nmax) ? J.n0 + c : J.nmax - 1;
  const bool tail = J.nch == 13 && g >= 2;     // this lane's K indices of chunk 12 do not exist: any valid address (X is zero there)
  // (a filler job's descriptor holds no records: every load is out of range = 0, without a memory access)
  a.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(J.W), (short)0,
                                           (FAIR_EXP & 256) ? 0 : (J.real != 0 || (FAIR_EXP & 16) != 0) ? 0x7FFFFFFF : 0, 0x00020000);
  a.nn = J.nn; a.wide = J.ldw != H; a.nch = J.nch;
  if (!J.nn) {
    a.voff = (n * J.ldw + 4 * g) * 4;
    a.vtail = tail ? a.voff : a.voff + 192 * 4;
  } else {
    a.voff = (4 * g * J.ldw + n) * 4;
    a.vtail = tail ? n * 4 : a.voff;      // (row offsets are scalar: the clamped lane drops its own 4 g rows -> rows 192 + s)
  }
  return a;
}
// chunk j (compile-time) of a job's weight fragments.  Buffer loads: descriptor + 32-bit lane offset + scalar / immediate offset —
// no VALU instruction per load (profiles/r04: every VALU instruction between f32 MFMAs costs ~14 cycles of issue).
template <int j>
__device__ __forceinline__ void load_chunk(Frag &F, const JobAddr &a) {
  if (FAIR_EXP & 2) return;
  if ((FAIR_EXP & 32) && a.nn) return;      // (32: no backward weight loads, 64: no forward weight loads)
  if ((FAIR_EXP & 64) && !a.nn) return;
  if (!a.nn) {
    if (j == 12) F.w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a.rs, a.vtail, 0, 0));
    else F.w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a.rs, a.voff, 64 * j, 0));
  } else {
    const int v = (j == 12) ? a.vtail : a.voff;
    if (!a.wide) {
#pragma unroll
      for (int s = 0; s < 4; s++) F.w[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a.rs, v, (16 * j + s) * H * 4, 0));
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++) F.w[j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a.rs, v, (16 * j + s) * CAT * 4, 0));
    }
  }
}
template <int j>
__device__ __forceinline__ void mac_chunk(f32x4 &acc, const Frag &F, const f32x4 &xv) {
#pragma unroll
  for (int s = 0; s < 4; s++) {
    if (FAIR_EXP & 1) acc[s] += F.w[j][s] * xv[s];
    else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(F.w[j][s], xv[s], acc, 0, 0, 0);
  }
}
// One slot of the pipeline: the NEXT job's weight fragments are requested chunk by chunk BETWEEN the products of the current job
// (fragments in `use`, loaded a slot ago).  All at once in front of the products, the eight waves of the workgroup — in step behind
// every barrier — queued 104 KB at the CU's one vector-memory pipe (64 B per clock) and waited for it, then all multiplied while it
// sat idle: 12-14 k cycles per 200 x 200 layer against 6.7 k of MFMA issue (in-kernel stamps, profiles/r05).
// (the sample fragment of chunk j + 1 is read from LDS in front of chunk j's MFMAs: its latency sits under them)
template <int j>
__device__ __forceinline__ void slot_step(f32x4 &acc, const Frag &use, Frag &pre, const JobAddr &na, const int nch, const float *xrow,
                                          f32x4 &xv) {
  f32x4 xn = xv;
  if (!(FAIR_EXP & 4) && (j + 1 < 13 || (j + 1 < FCH && nch > 13))) xn = ld4(xrow + 16 * (j + 1));
  if (j < 13 || na.nch > 13) load_chunk<j>(pre, na);
  if (j < 13 || nch > 13) mac_chunk<j>(acc, use, xv);
  __builtin_amdgcn_sched_barrier(0);
  xv = xn;
}
__device__ __forceinline__ void slot_run(f32x4 &acc, const Frag &use, Frag &pre, const Job &cur, const Job &nxt, const float *X,
                                         const int ldx, const int c, const int g, const int gs) {
  const JobAddr na = job_addr(nxt, c, g);
  const float *xrow = X + c * ldx + 4 * (ldx == LDA ? gs : g);     // (activation buffers: the swizzled piece; the observation rows: plain)
  const int nch = cur.nch;
  f32x4 xv = (FAIR_EXP & 4) ? f32x4{1.f, 1.f, 1.f, 1.f} : ld4(xrow);
  __builtin_amdgcn_sched_barrier(0);
  slot_step<0>(acc, use, pre, na, nch, xrow, xv);
  slot_step<1>(acc, use, pre, na, nch, xrow, xv);
  slot_step<2>(acc, use, pre, na, nch, xrow, xv);
  slot_step<3>(acc, use, pre, na, nch, xrow, xv);
  slot_step<4>(acc, use, pre, na, nch, xrow, xv);
  slot_step<5>(acc, use, pre, na, nch, xrow, xv);
  slot_step<6>(acc, use, pre, na, nch, xrow, xv);
  slot_step<7>(acc, use, pre, na, nch, xrow, xv);
  slot_step<8>(acc, use, pre, na, nch, xrow, xv);
  slot_step<9>(acc, use, pre, na, nch, xrow, xv);
  slot_step<10>(acc, use, pre, na, nch, xrow, xv);
  slot_step<11>(acc, use, pre, na, nch, xrow, xv);
  slot_step<12>(acc, use, pre, na, nch, xrow, xv);
  if (nch > 13 || na.nch > 13) {
    slot_step<13>(acc, use, pre, na, nch, xrow, xv);
    slot_step<14>(acc, use, pre, na, nch, xrow, xv);
  }
}
__device__ __forceinline__ void load_job(Frag &F, const Job &J, const int c, const int g) {     // (the very first job only)
  const JobAddr a = job_addr(J, c, g);
  load_chunk<0>(F, a); load_chunk<1>(F, a); load_chunk<2>(F, a); load_chunk<3>(F, a); load_chunk<4>(F, a); load_chunk<5>(F, a);
  load_chunk<6>(F, a); load_chunk<7>(F, a); load_chunk<8>(F, a); load_chunk<9>(F, a); load_chunk<10>(F, a); load_chunk<11>(F, a);
  load_chunk<12>(F, a);
  if (a.nch > 13) { load_chunk<13>(F, a); load_chunk<14>(F, a); }
}

// TRAIN == false: the forward pass alone (`actor(x), critic(x)` for rollouts / evaluators: brl_fair_forward) — nothing but the heads
// is written, any number of rows (a partial last block reads clamped rows), 75 KB of LDS = two workgroups per CU.
template <bool TRAIN>
__global__ __launch_bounds__(NW * 64) void k_fair_chain(Args A) {
  __shared__ __attribute__((aligned(16))) float X0[R * LDX];
  __shared__ __attribute__((aligned(16))) float AB[3][R * LDA];
  __shared__ __attribute__((aligned(16))) float HD[R * LDH], DH[TRAIN ? R * LDH : 4];
  __shared__ float illp_s[TRAIN ? R : 1][BRL_NUM_ACTIONS], part_s[TRAIN ? R : 1][8];
  __shared__ float rs_red[NW], rs_stat[2];
  const int tid = (int)threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int gs = g ^ ((0x78 >> (2 * (c >> 2))) & 3);        // this lane's piece inside a four-piece group of an activation row (LDA)
  const int sw4 = 4 * (gs - g);                             // ... as a correction to a column index n0 + 4 g
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the compiler must know the wave index is uniform: jobs live in SGPRs)
  const int64_t B = TRAIN ? A.P.B : A.nrows, row0 = (int64_t)blockIdx.x * R, nwg = gridDim.x;
  const int act = A.act;
  const Net &N = A.net;
  const Bufs &O = A.o;
  const int64_t BH = B * H;
  float *const a0 = AB[0], *const a1 = AB[1], *const a2 = AB[2];

  // the two tiles of this wave: ADJACENT ones (waves 0..5: tiles 2 w, 2 w + 1; wave 6: tile 12) — in the backward products a tile's
  // weights are 64 B of every row, so a wave's two tiles share the 128-byte lines.  Waves without a tile in a slot run a filler job
  // (no memory traffic, result dropped): straight-line code is what keeps the compiler from sinking the prefetch loads to their
  // uses, and the SIMD they sit on is not the critical one (13 tiles on 4 SIMDs: one of them has 4 either way).
  const bool hasA = w < 7, hasB = w < 6;
  const int tA = hasA ? 32 * w : 0, tB = hasB ? tA + 16 : tA;
  auto job_fwd = [&](const float *W, int ldw, int nch, bool B) { return Job{W, ldw, nch, 0, B ? tB : tA, H, (B ? hasB : hasA) ? 1 : 0}; };
  auto job_bwd = [&](const float *W, int ldw, bool B) { return Job{W, ldw, 13, 1, B ? tB : tA, H, (B ? hasB : hasA) ? 1 : 0}; };
  Frag FP, FQ;
  int stamp_i = 0;
  (void)stamp_i;
  FAIR_STAMP(stamp_i++);
  // ---- the observation rows -> LDS (and into the right block of cat6 = jnp.concatenate([x, input]), src/models.py:51): requested
  // first (memory operations retire in order), then the first job's weights; the LDS images are cleared while both travel
  constexpr int XP = (R * (OBS / 4) + NW * 64 - 1) / (NW * 64);      // 16-byte pieces per thread: 4 (the last one partly)
  f32x4 xv[XP];
#pragma unroll
  for (int i = 0; i < XP; i++) {
    const int e = tid + i * NW * 64, ec = (e < R * (OBS / 4)) ? e : 0;
    const int r = ec / (OBS / 4), q = ec - r * (OBS / 4);
    const int64_t row = (TRAIN || row0 + r < B) ? row0 + r : B - 1;
    xv[i] = ld4(A.x0 + row * OBS + 4 * q);
  }
  load_job(FP, job_fwd(N.w[0], OBS, 15, false), c, g);
  for (int e = tid; e < R * LDH; e += NW * 64) {
    HD[e] = 0.f;
    if (TRAIN) DH[e] = 0.f;
  }
  for (int e = tid; e < 3 * R * LDA / 4; e += NW * 64) st4(AB[0] + 4 * e, f32x4{0.f, 0.f, 0.f, 0.f});   // (pad columns 200..211: never written again)
#pragma unroll
  for (int i = 0; i < XP; i++) {
    const int e = tid + i * NW * 64;
    if (e < R * (OBS / 4)) {
      const int r = e / (OBS / 4), q = e - r * (OBS / 4);
      st4(X0 + r * LDX + 4 * q, xv[i]);
      if (TRAIN) st4(O.cat6 + (row0 + r) * CAT + H + 4 * q, xv[i]);
    }
  }
  // the minibatch's advantage statistics (reward_scaling: src/update.py:31-44, jnp std = ddof 0) — every workgroup forms them
  // itself in the same fixed order: identical everywhere, no hand-off
  float adv_mean = 0.0f, adv_inv = 1.0f;
  if (TRAIN && A.reward_scaling) {
    float s = 0.0f;
    for (int64_t i = tid; i < B; i += NW * 64) s += A.P.gae[i];
    s = wave_sum_f(s);
    if (lane == 0) rs_red[w] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int k = 0; k < NW; k++) t += rs_red[k];
      rs_stat[0] = t / (float)B;
    }
    __syncthreads();
    adv_mean = rs_stat[0];
    float q = 0.0f;
    for (int64_t i = tid; i < B; i += NW * 64) {
      const float d = A.P.gae[i] - adv_mean;
      q += d * d;
    }
    q = wave_sum_f(q);
    __syncthreads();
    if (lane == 0) rs_red[w] = q;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int k = 0; k < NW; k++) t += rs_red[k];
      rs_stat[1] = 1.0f / (sqrtf(t / (float)B) + 1e-8f);
    }
    __syncthreads();
    adv_inv = rs_stat[1];
  }
  __syncthreads();
  FAIR_STAMP(stamp_i++);

  // position of this lane's four outputs in the tile at column n0: row c, columns n0 + 4 g .. + 3 (all four exist or none)
  const int64_t grow = (row0 + c) * H;          // offset of row c in a [B,200] array
  float *const sink = AB[0];   // (FAIR_EXP & 8: global stores land in LDS instead)
  auto gptr = [&](float *base, int col) { return (FAIR_EXP & 8) ? sink + c * LDA + (col % 200) : base + grow + col; };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto gst = [&](float *p, const f32x4 &v) { if (TRAIN) st4(p, v); };   // (the forward pass alone keeps nothing but the heads)

  // one slot of the pipeline: request the NEXT job's weights into `pre`, then multiply the current job (fragments in `use`) into acc.
  // `e4` (may be nullptr-free: always a valid address): the epilogue's global operand (bias / gate values) of THIS tile, requested
  // FIRST — memory operations retire in order (one vmcnt counter), so an epilogue load issued behind the prefetch would wait for
  // all of it.
  auto slot = [&](Frag &use, Frag &pre, const Job &cur, const Job &nxt, const float *X, int ldx, f32x4 &acc, const float *e4) {
    const f32x4 e = ld4(e4);
    slot_run(acc, use, pre, cur, nxt, X, ldx, c, g, gs);
    return e;
  };
  struct E2 { f32x4 a, b; };
  auto slot2 = [&](Frag &use, Frag &pre, const Job &cur, const Job &nxt, const float *X, int ldx, f32x4 &acc, const float *e4a,
                   const float *e4b) {
    E2 e;
    e.a = ld4(e4a);
    e.b = ld4(e4b);
    slot_run(acc, use, pre, cur, nxt, X, ldx, c, g, gs);
    return e;
  };
  // this lane's columns in the tile at n0, clamped to existing ones (the clamped lanes' results are dropped)
  auto colc = [&](int n0) { const int col = n0 + 4 * g; return (col < H) ? col : H - 4; };
  // a 200 x 200 product phase: two slots (tile A, tile B), the epilogue `ep(acc, e, n0, valid)` behind each, the barrier;
  // `next` = the first job of the phase behind it; eptr(n0) = where the epilogue's global operand of a tile lives.
  // (F1 holds tile A's weights on entry and `next`'s on exit: FP in the forward pass, FQ in the backward pass — the single-slot
  //  heads phase between them flips the parity)
  auto phase200 = [&](Frag &F1, Frag &F2, const Job &jA, const Job &jB, const float *X, const Job &next, auto &&eptr, auto &&ep) {
    f32x4 acc = zero4;
    f32x4 e = slot(F1, F2, jA, jB, X, LDA, acc, eptr(jA.n0));
    ep(acc, e, jA.n0, hasA);
    acc = zero4;
    e = slot(F2, F1, jB, next, X, LDA, acc, eptr(jB.n0));
    ep(acc, e, jB.n0, hasB);
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  };
  auto fwd_jobs = [&](int l, Job &jA, Job &jB) {
    jA = job_fwd(N.w[l], H, 13, false);
    jB = job_fwd(N.w[l], H, 13, true);
  };
  auto bwd_jobs = [&](int l, int ldw, Job &jA, Job &jB) {
    jA = job_bwd(N.w[l], ldw, false);
    jB = job_bwd(N.w[l], ldw, true);
  };

  // ================================================= forward (src/models.py:34-69)
  // L0 (K = 480 = two jobs per tile): z0 (the shortcut is the PRE-activation) -> a0, h0 = act(z0) -> a1 (= inp[0], the input of L1)
  {
    auto ep = [&](const f32x4 &acc, const f32x4 &bias, int n0, bool valid) {
      const int col = n0 + 4 * g;
      if (col < H && valid) {
        const f32x4 z = acc + bias, h = act4(z, act);
        st4(a0 + c * LDA + col + sw4, z);
        st4(a1 + c * LDA + col + sw4, h);
        gst(gptr(O.inp + 0 * BH, col), h);
      }
    };
    const Job jA0 = job_fwd(N.w[0], OBS, 15, false), jA1 = job_fwd(N.w[0] + 240, OBS, 15, false);
    const Job jB0 = job_fwd(N.w[0], OBS, 15, true), jB1 = job_fwd(N.w[0] + 240, OBS, 15, true);
    f32x4 acc = zero4;
    f32x4 e = slot(FP, FQ, jA0, jA1, X0, LDX, acc, N.b[0] + colc(tA));
    (void)slot(FQ, FP, jA1, jB0, X0 + 240, LDX, acc, N.b[0]);
    ep(acc, e, tA, hasA);
    acc = zero4;
    e = slot(FP, FQ, jB0, jB1, X0, LDX, acc, N.b[0] + colc(tB));
    (void)slot(FQ, FP, jB1, job_fwd(N.w[1], H, 13, false), X0 + 240, LDX, acc, N.b[0]);
    ep(acc, e, tB, hasB);
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  }
  // a plain layer: out = act(in W^T + b) -> LDS `out` and the global array `gout`
  auto layer = [&](const float *in, float *out, int l, float *gout, const Job &next) {
    Job jA, jB;
    fwd_jobs(l, jA, jB);
    phase200(FP, FQ, jA, jB, in, next, [&](int n0) { return N.b[l] + colc(n0); },
             [&](const f32x4 &acc, const f32x4 &bias, int n0, bool valid) {
      const int col = n0 + 4 * g;
      if (col < H && valid) {
        const f32x4 h = act4(acc + bias, act);
        st4(out + c * LDA + col + sw4, h);
        gst(gptr(gout, col), h);
      }
    });
  };
  // the second layer of a residual block: h = act(in W^T + b) -> gate array; x = h + res (in place in `res`); then either
  // act(x) -> `out` + gout (the next block's input) or x itself -> gout
  auto layer_res = [&](const float *in, float *res, int l, float *ggate, float *out, float *gout, bool act_out, const Job &next) {
    Job jA, jB;
    fwd_jobs(l, jA, jB);
    phase200(FP, FQ, jA, jB, in, next, [&](int n0) { return N.b[l] + colc(n0); },
             [&](const f32x4 &acc, const f32x4 &bias, int n0, bool valid) {
      const int col = n0 + 4 * g;
      if (col < H && valid) {
        const f32x4 h = act4(acc + bias, act);
        gst(gptr(ggate, col), h);
        const f32x4 x = h + ld4(res + c * LDA + col + sw4);
        st4(res + c * LDA + col + sw4, x);
        if (act_out) {
          const f32x4 gg = act4(x, act);
          st4(out + c * LDA + col + sw4, gg);
          gst(gptr(gout, col), gg);
        } else {
          gst(gptr(gout, col), x);
        }
      }
    });
  };
  auto first_fwd = [&](int l) { return job_fwd(N.w[l], H, 13, false); };
  layer(a1, a2, 1, O.inp + 1 * BH, first_fwd(2));                                              // h1
  layer_res(a2, a0, 2, O.gates + 0 * BH, a1, O.inp + 2 * BH, true, first_fwd(3));              // h2; x1 = h2 + z0 (a0); g1 = act(x1) (a1)
  layer(a1, a2, 3, O.inp + 3 * BH, first_fwd(4));                                              // h3
  layer_res(a2, a0, 4, O.gates + 1 * BH, nullptr, O.inp + 4 * BH, false, first_fwd(5));        // h4; x2 = h4 + x1 (a0) = inp[4]
  // L5: z5 = x2 W5^T + b5 (no activation) -> a1 and the left block of cat6
  {
    Job jA, jB;
    fwd_jobs(5, jA, jB);
    phase200(FP, FQ, jA, jB, a0, job_fwd(N.w[6], CAT, 13, false), [&](int n0) { return N.b[5] + colc(n0); },
             [&](const f32x4 &acc, const f32x4 &bias, int n0, bool valid) {
      const int col = n0 + 4 * g;
      if (col < H && valid) {
        const f32x4 z = acc + bias;
        st4(a1 + c * LDA + col + sw4, z);
        gst(O.cat6 + (row0 + c) * CAT + col, z);
      }
    });
  }
  // L6 on [z5 | x0] (three jobs per tile: K = 200 of z5, 2 x 240 of x0): z6 -> a0 (shortcut_3), h6 = act(z6) -> a2 (= inp[5])
  {
    auto ep = [&](const f32x4 &acc, const f32x4 &bias, int n0, bool valid) {
      const int col = n0 + 4 * g;
      if (col < H && valid) {
        const f32x4 z = acc + bias, h = act4(z, act);
        st4(a0 + c * LDA + col + sw4, z);
        st4(a2 + c * LDA + col + sw4, h);
        gst(gptr(O.inp + 5 * BH, col), h);
      }
    };
    const float *W6 = N.w[6];
    const Job jA0 = job_fwd(W6, CAT, 13, false), jA1 = job_fwd(W6 + H, CAT, 15, false), jA2 = job_fwd(W6 + H + 240, CAT, 15, false);
    const Job jB0 = job_fwd(W6, CAT, 13, true), jB1 = job_fwd(W6 + H, CAT, 15, true), jB2 = job_fwd(W6 + H + 240, CAT, 15, true);
    f32x4 acc = zero4;
    f32x4 e = slot(FP, FQ, jA0, jA1, a1, LDA, acc, N.b[6] + colc(tA));
    (void)slot(FQ, FP, jA1, jA2, X0, LDX, acc, N.b[6]);
    (void)slot(FP, FQ, jA2, jB0, X0 + 240, LDX, acc, N.b[6]);
    ep(acc, e, tA, hasA);
    acc = zero4;
    e = slot(FQ, FP, jB0, jB1, a1, LDA, acc, N.b[6] + colc(tB));
    (void)slot(FP, FQ, jB1, jB2, X0, LDX, acc, N.b[6]);
    (void)slot(FQ, FP, jB2, first_fwd(7), X0 + 240, LDX, acc, N.b[6]);
    ep(acc, e, tB, hasB);
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  }
  layer(a2, a1, 7, O.inp + 6 * BH, first_fwd(8));                                              // h7
  layer_res(a1, a0, 8, O.gates + 2 * BH, a2, O.inp + 7 * BH, true, first_fwd(9));              // h8; x3 = h8 + z6 (a0); g3 (a2)
  layer(a2, a1, 9, O.inp + 8 * BH, first_fwd(10));                                             // h9
  // (behind L10: the heads' product, then — across the loss — the first backward product's weights)
  const Job jheads = Job{N.wh, H, 13, 0, 16 * ((w < 3) ? w : 2), HD_NOUT, (w < 3) ? 1 : 0};      // (waves 3..7: a filler job)
  // what the loss of this wave's two samples reads from global memory: requested two layers ahead of its use
  PpoSampleIn smp[2];
  if (TRAIN) {
    smp[0] = ppo_sample_load(A.P, row0 + 2 * w, true, lane);
    smp[1] = ppo_sample_load(A.P, row0 + 2 * w + 1, true, lane);
  }
  layer_res(a1, a0, 10, O.gates + 3 * BH, nullptr, O.x4, false, jheads);                       // h10; x4 = h10 + x3 (a0)
  // heads = x4 Wh^T + bh: 39 columns = 3 tiles
  {
    Job bA, bB;
    bwd_jobs(10, H, bA, bB);
    (void)bB;
    if (!TRAIN) bA.real = 0;
    f32x4 acc = zero4;
    (void)slot(FP, FQ, jheads, bA, a0, LDA, acc, N.b[0]);     // dz9 = (dz10 W10) ...: its tile-A weights (FQ) travel while the loss is computed
    if (w < 3) {
      const int col = 16 * w + 4 * g;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (col + i < HD_NOUT) HD[c * LDH + col + i] = acc[i] + N.bh[col + i];
    }
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  }

  if (!TRAIN) {   // the forward pass alone: logits [nrows,38], value [nrows]
    for (int e = tid; e < R * HD_NOUT; e += NW * 64) {
      const int r = e / HD_NOUT, n = e - r * HD_NOUT;
      if (row0 + r < B) {
        if (n < BRL_NUM_ACTIONS) A.logits_out[(row0 + r) * BRL_NUM_ACTIONS + n] = HD[r * LDH + n];
        else A.value_out[row0 + r] = HD[r * LDH + n];
      }
    }
    return;
  }

  // (what the first backward product, dx = d(heads) Wh, reads from global memory — its K = 39 is three chunks, not worth a place in
  //  the pipeline — travels while the loss is computed: the weight fragments of this wave's tiles w, w + 8 and their gate values)
  f32x4 hw_[2][3], hg_[2];
  {
    const float *g3 = O.gates + 3 * BH;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int t = (w + 8 * i < NT) ? w + 8 * i : w, n = (16 * t + c < H) ? 16 * t + c : H - 1;
#pragma unroll
      for (int j = 0; j < 3; j++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k = 16 * j + 4 * g + q;
          hw_[i][j][q] = N.wh[(int64_t)((k < HD_NOUT) ? k : HD_NOUT - 1) * H + n];
        }
      }
      hg_[i] = ld4(g3 + grow + colc(16 * t));
    }
  }

  // ================================================= `_loss_fn` (src/update.py:90-167): wave w takes rows 2 w, 2 w + 1
  {
    PpoArgs P = A.P;
    P.dlogits = DH; P.dls = LDH; P.dvalue = DH + BRL_NUM_ACTIONS; P.dvs = LDH; P.illp = &illp_s[0][0];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = 2 * w + i;
      const PpoSampleIn S = smp[i];
      const float lg = (lane < HD_NOUT) ? HD[r * LDH + lane] : 0.0f;
      const float v = HD[r * LDH + BRL_NUM_ACTIONS];
      const float adv = A.reward_scaling ? (S.gae - adv_mean) * adv_inv : S.gae;
      float st[5], ill;
      ppo_loss_sample(P, S, r, true, lane, lg, v, adv, st, ill);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) part_s[r][k] = (k < 5) ? st[k] : 0.0f;
      }
    }
  }
  __syncthreads();
  FAIR_STAMP(stamp_i++);
  if (tid < 8) {   // the workgroup's statistics partials, rows in order (deterministic)
    float s = 0.0f;
    for (int r = 0; r < R; r++) s += part_s[r][tid];
    O.partials[(int64_t)blockIdx.x * 8 + tid] = s;
  }
  if (O.gram_partials != nullptr) {   // P^T P of the workgroup's illegal-action probabilities (src/update.py:136-141)
    for (int e = tid; e < HD_GRAM; e += NW * 64) {
      const int i = e / BRL_NUM_ACTIONS, j = e - i * BRL_NUM_ACTIONS;
      float s = 0.0f;
#pragma unroll
      for (int r = 0; r < R; r++) s += illp_s[r][i] * illp_s[r][j];
      O.gram_partials[(int64_t)blockIdx.x * HD_GRAM + e] = s;
    }
  }
  for (int e = tid; e < R * 40; e += NW * 64) {      // [B][40]: column 39 = 0 (a 16-byte-piece row for the weight-gradient product)
    const int r = e / 40, n = e - r * 40;
    O.dheads[(row0 + r) * 40 + n] = DH[r * LDH + n];
  }
  float *const tiles_wg = O.tiles + (int64_t)blockIdx.x * H;   // + l * nwg * H: this workgroup's row of layer l's partials
  const int64_t tstride = nwg * H;
  if (tid < HD_NOUT) {   // the heads' bias gradient: column sums of d(heads)
    float s = 0.0f;
    for (int r = 0; r < R; r++) s += DH[r * LDH + tid];
    O.tiles[11 * tstride + (int64_t)blockIdx.x * HD_NOUT + tid] = s;   // (segment 11: [B/16][39], compact)
  }

  // ================================================= backward.  dx = a0, dz_b = a1, dz_a = a2
  // column sums of the tile (valid columns only) -> layer l's partials
  auto colsum = [&](f32x4 v, int col, int l) {
    const f32x4 s = row16_sum4(v);
    if (c == 0 && col < H) st4(tiles_wg + l * tstride + col, s);
  };
  // What follows a finished dx (the gradient w.r.t. a block's output, in a0) is always the same step — dz of the layer above it:
  // dz = dx * act'(gate2) (gate2 == nullptr: dz = dx) -> LDS dst2 (may be nullptr), its global copy gout2 and layer lsum2's column
  // sums.  It rides in the epilogue of the product that finishes dx (a pass of its own would be a barrier, an LDS round trip and a
  // global load that waits for the whole weight prefetch).
  struct Then { const float *gate2; float *dst2, *gout2; int lsum2; };
  auto then_step = [&](const f32x4 &dx, const f32x4 &g2, int col, bool valid, const Then &T) {
    f32x4 v = zero4;
    if (col < H && valid) {
      v = (T.gate2 != nullptr) ? dact4(dx, g2, act) : dx;
      if (T.dst2 != nullptr) st4(T.dst2 + c * LDA + col + sw4, v);
      st4(gptr(T.gout2, col), v);
    }
    if (valid) colsum(v, col, T.lsum2);
  };
  // dx = d(heads) Wh -> a0; then dz10 = dx * act'(h10)
  {
    const Then T{O.gates + 3 * BH, a1, O.dzs + 8 * BH, 10};
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool valid = w + 8 * i < NT;
      const int t = valid ? w + 8 * i : w, col = 16 * t + 4 * g;
      f32x4 acc = zero4;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const f32x4 xv = ld4(DH + c * LDH + 16 * j + 4 * g);       // (columns 39..47 of DH are zero)
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hw_[i][j][q], xv[q], acc, 0, 0, 0);
      }
      if (col < H && valid) st4(a0 + c * LDA + col + sw4, acc);
      then_step(acc, hg_[i], col, valid, T);
    }
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  }
  // dst = (src W_l) * act'(gate) [+ the old dst when `accumulate`]; gout / lsum: global copy + column sums (dz of layer lsum) or none;
  // T (dst is a finished dx): the step above
  auto back = [&](const float *src, float *dst, int l, int ldw, const float *ggate, float *gout, int lsum, bool accumulate,
                  const Job &next, const Then *T) {
    Job jA, jB;
    bwd_jobs(l, ldw, jA, jB);
    const float *gbase = (ggate != nullptr) ? ggate : O.inp;      // (no gate: any valid address, the values are ignored)
    const float *g2base = (T != nullptr && T->gate2 != nullptr) ? T->gate2 : O.inp;
    auto ep = [&](const f32x4 &acc, const E2 &e, int n0, bool valid) {
      const int col = n0 + 4 * g;
      f32x4 v = zero4;
      if (col < H && valid) {
        v = acc;
        if (ggate != nullptr) v = dact4(v, e.a, act);
        if (accumulate) v += ld4(dst + c * LDA + col + sw4);
        st4(dst + c * LDA + col + sw4, v);
        if (gout != nullptr) st4(gptr(gout, col), v);
      }
      if (lsum >= 0 && valid) colsum(v, col, lsum);
      if (T != nullptr) then_step(v, e.b, col, valid, *T);
    };
    f32x4 acc = zero4;
    E2 e = slot2(FQ, FP, jA, jB, src, LDA, acc, gbase + grow + colc(jA.n0), g2base + grow + colc(jA.n0));
    ep(acc, e, jA.n0, hasA);
    acc = zero4;
    e = slot2(FP, FQ, jB, next, src, LDA, acc, gbase + grow + colc(jB.n0), g2base + grow + colc(jB.n0));
    ep(acc, e, jB.n0, hasB);
    __syncthreads();
    FAIR_STAMP(stamp_i++);
  };
  auto first_bwd = [&](int l, int ldw) { return job_bwd(N.w[l], ldw, false); };
  // a residual block  x_out = act(L_b(act(L_a(g)))) + x_in, g = act(x_in or a pre-activation), its dz_b already formed (a1):
  // dz_a = (dz_b W_b) act'(h_a) (kept for the weight gradients), dx += (dz_a W_a) act'(g), then T on the new dx
  auto block = [&](int lb, int la, int ia, const float *ga, const float *gin, const Job &next, const Then &T, float *zb, float *za) {
    back(zb, za, lb, H, ga, O.dzs + (int64_t)ia * BH, la, false, first_bwd(la, H), nullptr);
    back(za, a0, la, H, gin, nullptr, -1, true, next, &T);
  };
  // (T.dst2 — the next dz_b — must not be the buffer the product that carries T reads: a1 above L6, a2 below it)
  const Then T8{O.gates + 2 * BH, a1, O.dzs + 6 * BH, 8}, T6{nullptr, nullptr, O.dz6, 6}, T4{O.gates + 1 * BH, a2, O.dzs + 3 * BH, 4},
      T2{O.gates + 0 * BH, a2, O.dzs + 1 * BH, 2}, T0{nullptr, nullptr, O.dz0, 0};
  block(10, 9, 7, O.inp + 8 * BH, O.inp + 7 * BH, first_bwd(8, H), T8, a1, a2);   // -> d/dx3, dz8 (a1)
  block(8, 7, 5, O.inp + 6 * BH, O.inp + 5 * BH, first_bwd(6, CAT), T6, a1, a2);  // -> d/dz6 = dz6
  back(a0, a1, 6, CAT, nullptr, O.dzs + 4 * BH, 5, false, first_bwd(5, H), nullptr);      // dz5 = dz6 W6[:, :200] (L5: no activation)
  back(a1, a0, 5, H, nullptr, nullptr, -1, false, first_bwd(4, H), &T4);                  // d/dx2 = dz5 W5, dz4 (a2)
  block(4, 3, 2, O.inp + 3 * BH, O.inp + 2 * BH, first_bwd(2, H), T2, a2, a1);    // -> d/dx1, dz2 (a2)
  block(2, 1, 0, O.inp + 1 * BH, O.inp + 0 * BH, first_bwd(1, H), T0, a2, a1);    // -> d/dz0 = dz0 (the last prefetch is never used)
}
}  // namespace fair
