// peer_adam.hip — PROTOTYPE (design + functional proof only; never the default before a node has timed it):
// the multi-rank PPO step's gradient collective FUSED into the optimizer launches.  Instead of RCCL's reduce-scatter / all-gather
// kernels on a second branch of the step's hipGraph (each cross-stream edge costs the graph ~10 us: profiles/r05/r05r_*), every rank
// maps its peers' flat gradient and parameter buffers (hipIpcOpenMemHandle: plain pointers over xGMI) and
//   k_peer_reduce_norm   reads slice r of EVERY rank's gradient (rank order: a deterministic sum), writes the reduced slice locally
//                        and its square-sum partials into every peer's partials table,
//   k_peer_apply         adds everybody's partials in one fixed order (every rank forms the SAME norm), clips, runs Adam on ITS slice
//                        (moments stay local: 1/world of the sweep) and stores the updated parameters into EVERY rank's buffer.
// Ordering between GPUs is three monotone flag words per peer — "my gradient of step s is complete", "my partials of step s are
// written", "my parameter slice of step s is written everywhere" — set by a one-wave signal launch behind the producing kernels (the
// kernel boundary is the release) and awaited by a one-wave wait launch in front of the consumers (a BOUNDED spin: on a time-out it
// raises an error word instead of hanging the device).  Step ids only grow: nothing is ever reset.  In the product form the waits
// would sit at the head of the two data kernels and the signals behind their last block (two launches per step instead of ten);
// the prototype keeps them apart so that no data kernel ever spins.
// Same arithmetic as csrc/adam_role.hpp (torch.optim.Adam + clip_grad_norm_), one bucket = the whole buffer in `world` slices.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -o scripts/micro/libpeer_adam.so scripts/micro/peer_adam.hip
//   python scripts/peer_adam_probe.py        (two processes on ONE GPU, buffers shared through hipIpcGetMemHandle)
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MAXW 8
constexpr int J = 64;                 // partial-sum blocks per rank
constexpr int THREADS = 256;

struct PeerCtl {                      // one per rank, in that rank's memory, mapped by every peer
  unsigned long long flag_grad[MAXW], flag_norm[MAXW], flag_param[MAXW];   // [q]: written by rank q
  unsigned int error;                 // a wait timed out
  unsigned int pad[7];
  float norm_part[MAXW][J];           // [q][j]: rank q's partials of ITS slice
};
struct Peers {
  int rank, world;
  const float *G[MAXW];               // every rank's gradient buffer
  float *P[MAXW];                     // every rank's parameter buffer
  PeerCtl *ctl[MAXW];                 // every rank's control block
};

// which: 0 flag_grad, 1 flag_norm, 2 flag_param.  One wave: lane q writes this rank's word in rank q's control block.
__global__ void k_peer_signal(Peers X, int which, unsigned long long step) {
  __threadfence_system();
  const int q = (int)threadIdx.x;
  if (q < X.world) {
    unsigned long long *f = which == 0 ? X.ctl[q]->flag_grad : which == 1 ? X.ctl[q]->flag_norm : X.ctl[q]->flag_param;
    __hip_atomic_store(&f[X.rank], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// One wave: lane q waits until rank q's word in THIS rank's control block has reached `step` (bounded: ~2 s), then acquires.
__global__ void k_peer_wait(Peers X, int which, unsigned long long step) {
  const int q = (int)threadIdx.x;
  PeerCtl *c = X.ctl[X.rank];
  if (q < X.world) {
    unsigned long long *f = which == 0 ? c->flag_grad : which == 1 ? c->flag_norm : c->flag_param;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    bool ok = false;
    while (!ok) {
      ok = __hip_atomic_load(&f[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= step;
      if (!ok) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { atomicOr(&c->error, 1u << which); break; }
        __builtin_amdgcn_s_sleep(32);
      }
    }
  }
  __threadfence_system();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block j of J: sub-block j of slice `rank`: sum over the ranks in rank order -> gred (local), square sums (of the MEAN) -> every peer
__global__ __launch_bounds__(THREADS) void k_peer_reduce_norm(Peers X, float *gred, int64_t n4, float gscale) {
  __shared__ float red[THREADS / 64];
  const int r = X.rank, W = X.world, j = (int)blockIdx.x;
  const int64_t len = n4 / W, chunk = (len + J - 1) / J, base = (int64_t)r * len;
  const int64_t lo = base + (int64_t)j * chunk, hi = (lo + chunk < base + len) ? lo + chunk : base + len;
  float s = 0.0f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += THREADS) {
    float4 a = reinterpret_cast<const float4 *>(X.G[0])[i];
    for (int q = 1; q < W; q++) {
      const float4 b = reinterpret_cast<const float4 *>(X.G[q])[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4 *>(gred)[i] = a;
    a.x *= gscale; a.y *= gscale; a.z *= gscale; a.w *= gscale;
    s += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x < (unsigned)W) X.ctl[threadIdx.x]->norm_part[r][j] = (red[0] + red[1]) + (red[2] + red[3]);
}

// block b of nb: clip + Adam on this rank's slice; the new parameters go to every rank
__global__ __launch_bounds__(THREADS) void k_peer_apply(Peers X, const float *gred, float *m, float *v, int64_t n4, float t, float lr, float b1,
                                                       float b2, float eps, float max_norm, float gscale, float *norm_out) {
  __shared__ float red[8];
  const int r = X.rank, W = X.world, tid = (int)threadIdx.x, nb = (int)gridDim.x, b = (int)blockIdx.x;
  {
    const PeerCtl *c = X.ctl[r];
    float s = 0.0f;
    for (int i = tid; i < W * J; i += THREADS) s += c->norm_part[i / J][i % J];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
      red[4] = (max_norm > 0.0f) ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (b == 0 && norm_out) *norm_out = norm;
    }
    __syncthreads();
  }
  const float scale = red[4] * gscale;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  const int64_t len = n4 / W, base = (int64_t)r * len, chunk = (len + nb - 1) / nb;
  const int64_t lo = base + (int64_t)b * chunk, hi = (lo + chunk < base + len) ? lo + chunk : base + len;
  for (int64_t i = lo + tid; i < hi; i += THREADS) {
    const float4 g4 = reinterpret_cast<const float4 *>(gred)[i];
    float4 m4 = reinterpret_cast<float4 *>(m)[i], v4 = reinterpret_cast<float4 *>(v)[i], p4 = reinterpret_cast<float4 *>(X.P[r])[i];
    const float gs[4] = {g4.x * scale, g4.y * scale, g4.z * scale, g4.w * scale};
    float ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w}, ps[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      ms[k] = ms[k] + (gs[k] - ms[k]) * (1.0f - b1);
      vs[k] = vs[k] * b2 + gs[k] * gs[k] * (1.0f - b2);
      ps[k] -= step_size * (ms[k] / (sqrtf(vs[k]) / bc2_sqrt + eps));
    }
    reinterpret_cast<float4 *>(m)[i] = make_float4(ms[0], ms[1], ms[2], ms[3]);
    reinterpret_cast<float4 *>(v)[i] = make_float4(vs[0], vs[1], vs[2], vs[3]);
    const float4 pn = make_float4(ps[0], ps[1], ps[2], ps[3]);
    for (int q = 0; q < W; q++) reinterpret_cast<float4 *>(X.P[q])[i] = pn;
  }
}

// ---- C entry points for the probe (plain pointers; `peers` = world pointers each) --------------------------------------------
extern "C" int peer_ctl_bytes() { return (int)sizeof(PeerCtl); }

static Peers make(int rank, int world, void *const *G, void *const *P, void *const *ctl) {
  Peers X{};
  X.rank = rank; X.world = world;
  for (int q = 0; q < world; q++) { X.G[q] = (const float *)G[q]; X.P[q] = (float *)P[q]; X.ctl[q] = (PeerCtl *)ctl[q]; }
  return X;
}

// one optimizer step of rank `rank`: its gradient (G[rank]) is complete in stream order when this is called
extern "C" int peer_adam_step(int rank, int world, void *const *G, void *const *P, void *const *ctl, float *gred, float *m, float *v,
                              int64_t n, unsigned long long step, float lr, float b1, float b2, float eps, float max_norm, float *norm_out,
                              void *stream) {
  if (world < 1 || world > MAXW || n % (4 * world)) return 1;
  const Peers X = make(rank, world, G, P, ctl);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = n / 4;
  const float gscale = 1.0f / (float)world;
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, X, 0, step);            // my gradient is complete
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, X, 0, step);              // ... and everybody's
  hipLaunchKernelGGL(k_peer_reduce_norm, dim3(J), dim3(THREADS), 0, s, X, gred, n4, gscale);
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, X, 1, step);            // my partials are everywhere
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, X, 1, step);
  const int nb = (int)((n4 / world + 1023) / 1024);
  hipLaunchKernelGGL(k_peer_apply, dim3(nb > 0 ? nb : 1), dim3(THREADS), 0, s, X, gred, m, v, n4, (float)step, lr, b1, b2, eps, max_norm, gscale,
                     norm_out);
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, X, 2, step);            // my parameter slice is everywhere
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, X, 2, step);              // ... and everybody's is here: the next forward may start
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
