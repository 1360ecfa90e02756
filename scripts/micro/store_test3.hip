// Store floor of ONE rollout launch's bytes when they really go to HBM: the P0 / P1 patterns of store_test2.hip, but the
// output rotates over NB buffers of 144 MB (NB x 144 MB > the 256 MB Infinity Cache), 64 back-to-back launches per event pair.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int P>
__global__ __launch_bounds__(512) void k_store(uint8_t *obs, uint8_t *mask, int64_t n, int T) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane / 15, ch = lane % 15;
  uint4 v = make_uint4(lane, wave, 1, 0x01010101u);
  const int vb = blockIdx.x, nvb = gridDim.x;
  for (int s = 0; s < T; s++) {
    const int64_t table0 = (int64_t)((vb % 8) * (nvb / 8) + vb / 8) * 32;
    int64_t row = (int64_t)s * n + table0 + 4 * wave;
    uint8_t *base = obs + row * 480;
    if (P == 0) {
      if (r < 4) {
        uint4 *dst = reinterpret_cast<uint4 *>(base + r * 480 + ch * 32);
        dst[0] = v;
        dst[1] = v;
      }
    } else {
      if (lane < 60) {
        *reinterpret_cast<uint4 *>(base + 16 * lane) = v;
        *reinterpret_cast<uint4 *>(base + 960 + 16 * lane) = v;
      }
    }
    if (lane < 38) reinterpret_cast<uint32_t *>(mask + row * 38)[lane] = v.w;
    v.x += 1;
  }
}
int main() {
  const int64_t n = 8192; const int T = 33;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int nb : {1, 3, 6}) {
    uint8_t *obs[6], *mask[6];
    for (int i = 0; i < nb; i++) { hipMalloc(&obs[i], n * T * 480 + 4096); hipMalloc(&mask[i], n * T * 38 + 64); }
    for (int P = 0; P < 2; P++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int it = 0; it < 64; it++) {
          if (P == 0) hipLaunchKernelGGL((k_store<0>), dim3(256), dim3(512), 0, 0, obs[it % nb], mask[it % nb], n, T);
          else hipLaunchKernelGGL((k_store<1>), dim3(256), dim3(512), 0, 0, obs[it % nb], mask[it % nb], n, T);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms / 64 < best) best = ms / 64;
      }
      double bytes = (double)n * T * 518;
      printf("buffers %d pattern P%d: %.1f us per launch  %.0f GB/s (518 B/row x 33 slots = %.0f MB)\n", nb, P, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / 1e6);
    }
    for (int i = 0; i < nb; i++) { hipFree(obs[i]); hipFree(mask[i]); }
  }
  return 0;
}
