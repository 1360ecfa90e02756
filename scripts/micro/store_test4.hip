// What a rollout launch could reach if its followers were never starved: the store probe of store_test3.hip (obs + mask
// bytes of 33 slots = 140 MB, output rotating over 3 buffers = 433 MB > Infinity Cache) with the stores PACED the way the
// logic wave makes rows available:
//   pace 0: no pacing (pure store floor)
//   pace 1: "flag" hand-off: slot s may be stored from  pro + per * s  cycles after the wave started
//   pace 2: batches 1,3,4,8,8,..: slot s may be stored once its whole batch is posted (what k_rollout_ws does)
//   pace 3: batches 1,1,2,4,8,8,.. (round 1)
// plus store flavours: P0 the kernel's lane mapping (2 x 16 B per lane at a 32-B stride), P1 fully contiguous 960-B
// instructions; plain / nontemporal / sc1 (write-through) stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16(uint8_t *p, u32x4 v, int flavour) {
  if (flavour == 0) *reinterpret_cast<u32x4 *>(p) = v;
  else if (flavour == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ int batch_end(int s, int sched) {  // last slot of the batch that holds slot s
  if (sched == 2) { if (s < 1) return 0; if (s < 4) return 3; if (s < 8) return 7; return (s / 8) * 8 + 7; }
  if (s < 1) return 0; if (s < 2) return 1; if (s < 4) return 3; if (s < 8) return 7; return (s / 8) * 8 + 7;
}

template <int P, int F>
__global__ __launch_bounds__(512) void k_store(uint8_t *obs, uint8_t *mask, int64_t n, int T, int pace, int pro, int per) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane / 15, ch = lane % 15;
  u32x4 v = {(uint32_t)lane, (uint32_t)wave, 1u, 0x01010101u};
  const int vb = blockIdx.x, nvb = gridDim.x;
  const int64_t table0 = (int64_t)((vb % 8) * (nvb / 8) + vb / 8) * 32;
  for (int s = 0; s < T; s++) {
    if (pace) {
      int sa = (pace == 1) ? s : batch_end(s, pace);
      if (sa > T - 1) sa = T - 1;
      const unsigned long long avail = (unsigned long long)pro + (unsigned long long)per * (unsigned long long)sa;
      while (__builtin_amdgcn_s_memtime() - t0 < avail) __builtin_amdgcn_s_sleep(2);
    }
    int64_t row = (int64_t)s * n + table0 + 4 * wave;
    uint8_t *base = obs + row * 480;
    if (P == 0) {
      if (r < 4) { st16(base + r * 480 + ch * 32, v, F); st16(base + r * 480 + ch * 32 + 16, v, F); }
    } else {
      if (lane < 60) { st16(base + 16 * lane, v, F); st16(base + 960 + 16 * lane, v, F); }
    }
    if (lane < 38) reinterpret_cast<uint32_t *>(mask + row * 38)[lane] = v.w;
    v.x += 1;
  }
}

template <int P, int F>
static float run(uint8_t **obs, uint8_t **mask, int nb, int pace, int pro, int per) {
  const int64_t n = 8192; const int T = 33;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < 64; it++)
      hipLaunchKernelGGL((k_store<P, F>), dim3(256), dim3(512), 0, 0, obs[it % nb], mask[it % nb], n, T, pace, pro, per);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms / 64 < best) best = ms / 64;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  return best * 1e3f;
}

int main() {
  const int64_t n = 8192; const int T = 33;
  for (int nb : {3, 6}) {
    uint8_t *obs[6], *mask[6];
    for (int i = 0; i < nb; i++) { hipMalloc(&obs[i], n * T * 480 + 4096); hipMalloc(&mask[i], n * T * 38 + 64); }
    printf("== %d rotating buffers\n", nb);
    printf("P0 plain %.1f  nt %.1f  sc0sc1 %.1f\n", run<0, 0>(obs, mask, nb, 0, 0, 0), run<0, 1>(obs, mask, nb, 0, 0, 0), run<0, 2>(obs, mask, nb, 0, 0, 0));
    printf("P1 plain %.1f  nt %.1f  sc0sc1 %.1f\n", run<1, 0>(obs, mask, nb, 0, 0, 0), run<1, 1>(obs, mask, nb, 0, 0, 0), run<1, 2>(obs, mask, nb, 0, 0, 0));
    for (int pro : {2400, 4800}) {
      for (int per : {800, 1170}) {
        printf("pro %d per %d:  P0 flag %.1f  b1348 %.1f  b11248 %.1f |  P1 flag %.1f  b1348 %.1f  b11248 %.1f | P1nt flag %.1f b1348 %.1f\n", pro, per,
               run<0, 0>(obs, mask, nb, 1, pro, per), run<0, 0>(obs, mask, nb, 2, pro, per), run<0, 0>(obs, mask, nb, 3, pro, per),
               run<1, 0>(obs, mask, nb, 1, pro, per), run<1, 0>(obs, mask, nb, 2, pro, per), run<1, 0>(obs, mask, nb, 3, pro, per),
               run<1, 1>(obs, mask, nb, 1, pro, per), run<1, 1>(obs, mask, nb, 2, pro, per));
        fflush(stdout);
      }
    }
    for (int i = 0; i < nb; i++) { hipFree(obs[i]); hipFree(mask[i]); }
  }
  return 0;
}
