// mlp16_fused.hpp — EXPERIMENT, not built into libbrl_hip.so (profiles/r03/r03_experiments.txt, "all hidden layers in one launch").
// The multi-layer form of csrc/mlp_infer.hpp: include it inside namespace lin16 behind linear_tile / xcd_logical_id.  Bit-identical
// to the layer-by-layer launches on every shape tried, 5 us SLOWER per 4-layer forward at 8192 tables — kept as the record of
// what was measured: the in-launch meeting (counter in global memory + L1 invalidate) costs what a graph node boundary costs.
// ---- the whole stack of hidden layers in ONE launch (src/models.py:23-33: 480 -> 1024 -> 1024 -> .. with relu).
// Layer l + 1's tile (tm, tn) reads all of row tile tm of layer l, i.e. the output of the `tiles_n` workgroups (tm, *): they form
// a GROUP that meets at a counter in global memory between two layers — nobody else waits for anybody.  The grid is at most one
// workgroup per CU (144 KB of LDS each: one fits), so every workgroup of a group is resident and the meeting cannot deadlock; a
// group with more than one row tile walks them one after the other.
// Memory order.  y always leaves WRITE-THROUGH (sc0 sc1: in memory, no dirty line anywhere), so a member's release is
// s_waitcnt vmcnt(0).  The acquire depends on where the members run: every workgroup ORs its XCC id (hardware register) into
// the group's mask before it computes anything; after a meeting the mask is complete, and
//   * one bit (blockIdx -> XCD round robin + the logical ids below: the normal case): the members share an L2, which is the
//     point of coherence of its CUs — only the CU's own vector L1 can be stale: buffer_inv sc0;
//   * several bits: agent-scope acquire (buffer_inv sc1: also the L2's copies), ~10 us per meeting but correct anywhere.
// (Measured: agent-scope release + acquire at every meeting made the launch 2.4x slower than four separate launches.)
// The counter only grows: a launch adds  group size x meetings  to it, so at the start of a launch its value is a multiple of
// the group size plus however many members already arrived at the first meeting (< group size; group size is a power of two).
// A poll gives up after ~0.3 s (MLP_SPIN_LIMIT) and raises *err: the grid always drains.
constexpr int MLP_MAX_LAYERS = 8;
constexpr int MLP_SPIN_LIMIT = 1 << 20;

struct MlpArgs {
  const uint16_t *x0;      // [M][ldx0] the observation in 16 bits
  int64_t ldx0;
  int k0;                  // its width (480)
  int nl;                  // layers
  const uint16_t *w[MLP_MAX_LAYERS];   // [H][ldw[l]]
  int64_t ldw[MLP_MAX_LAYERS];
  const float *bias[MLP_MAX_LAYERS];
  uint16_t *buf[2];        // [M][ldh] ping-pong for the layers in between
  uint16_t *out;           // [M][ldh] the last layer's output
  int64_t ldh;
  int M, H;                // H % 128 == 0, H / 128 a power of two
  uint32_t *sync;          // [groups] meeting counters (never reset)
  uint32_t *xcc_mask;      // [groups] XCC ids the group's members have run on (never reset: only ever more careful)
  int *err;
  int force_agent;         // experiment: agent-scope acquire at every meeting
  int l2_stores;           // experiment switch: plain stores between members that share an L2
  int groups;              // row-tile groups of this launch (<= 8 x floor(CUs per XCD / group size))
};

// returns (through *shared) bit 0: the careful acquire was taken (members on several XCDs); bits 8..: nothing
__device__ __forceinline__ void group_meet(uint32_t *ctr, uint32_t *target, bool first, int gs, const uint32_t *mask, int force_agent,
                                           int *err, uint32_t *shared) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's y stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // first meeting of the launch: the counter was a multiple of the group size when the launch began, and fewer than `gs`
    // members can have arrived before this one
    const uint32_t tgt = first ? (old & ~(uint32_t)(gs - 1)) + (uint32_t)gs : *target;
    int spins = 0;
    while ((int32_t)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tgt) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > MLP_SPIN_LIMIT) {
        *err = 1;
        break;
      }
    }
    const uint32_t m = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    shared[0] = (force_agent || (m & (m - 1)) != 0) ? 1u : 0u;
    shared[1] = tgt;
  }
  __syncthreads();
  *target = shared[1];
  if (shared[0]) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else asm volatile("buffer_inv sc0" ::: "memory");
}

template <int FMT>
__global__ __launch_bounds__(THREADS) void k_mlp16(MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ float bias_s[BN];
  __shared__ uint32_t meet_s[2];
  const int gs = A.H / BN;                               // group size = column tiles
  // blocks b, b + 8, .. share an XCD (round-robin dispatch): XCD x hosts the groups x, x + 8, .. — `gs` consecutive ones of its
  // blocks each.  The grid is padded to 8 x gs x ceil(groups / 8) blocks; the surplus leaves at once.
  const int groups = A.groups;
  const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
  const int g = xcd + 8 * (j / gs), tn = j % gs;
  if (g >= groups) return;
  const int tiles_m = (A.M + BM - 1) / BM;
  if (threadIdx.x == 0) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __hip_atomic_fetch_or(A.xcc_mask + g, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (before its first arrival)
  }
  uint32_t target = 0;
  bool first = true, shared_l2 = false;   // shared_l2: known (from a meeting of this launch) that the group's members share an L2
  for (int tm = g; tm < tiles_m; tm += groups) {
    for (int l = 0; l < A.nl; l++) {
      Args G;
      G.x = (l == 0) ? A.x0 : A.buf[(l - 1) & 1];
      G.ldx = (l == 0) ? A.ldx0 : A.ldh;
      G.w = A.w[l];
      G.ldw = A.ldw[l];
      G.bias = A.bias[l];
      G.y = (l + 1 == A.nl) ? A.out : A.buf[l & 1];
      G.ldy = A.ldh;
      G.M = A.M;
      G.N = A.H;
      G.K = (l == 0) ? A.k0 : A.H;
      G.relu = 1;
      // y leaves write-through (what a meeting's release relies on wherever the members run) — except between members known
      // to share an L2: there a plain store, acknowledged by that L2, is enough and the next layer reads it from there
      G.store_mode = (l + 1 < A.nl && shared_l2 && A.l2_stores) ? 0 : 2;
#ifdef LIN16_TIMING
      G.dbg = nullptr;
#endif
      linear_tile<FMT>(G, lds, bias_s, tm, tn);
      if (l + 1 < A.nl) {
        target += (uint32_t)gs;
        group_meet(A.sync + g, &target, first, gs, A.xcc_mask + g, A.force_agent, A.err, meet_s);
        first = false;
        shared_l2 = meet_s[0] == 0;
        __syncthreads();   // (meet_s is rewritten at the next meeting)
      } else {
        __syncthreads();   // (the next row tile's first DMA overwrites the output tile in LDS)
      }
    }
  }
}

