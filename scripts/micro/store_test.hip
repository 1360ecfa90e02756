// Pure store-throughput probe with the rollout kernel's emission pattern (timing experiment only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NW, int NE0, int NT = 0>
__global__ __launch_bounds__(NW * 64) void k_store(uint8_t *obs, uint8_t *mask, int64_t n, int T, int mode, float *cols = nullptr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (cols && wave == 2) {  // the scorer's five scalar columns: 4 x 128 B + 32 B per sub-step
    const int64_t t0 = (int64_t)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) * 32;
    for (int s = 0; s < T; s++) {
      if (lane < 32) {
        int64_t row = (int64_t)s * n + t0 + lane;
        cols[row] = 1.0f; cols[n * T + row] = 2.0f; cols[2 * n * T + row] = 3.0f; cols[3 * n * T + row] = 4.0f;
        reinterpret_cast<uint8_t *>(cols + 4 * n * T)[row] = 1;
      }
      __builtin_amdgcn_s_sleep(10);
    }
    return;
  }
  if (wave < NE0) return;
  const int NE = NW - NE0;
  const int r = lane / 15, ch = lane % 15;
  const int64_t table0 = (int64_t)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) * 32;
  uint4 v = make_uint4(lane, wave, 1, 0x01010101u);
  for (int s = 0; s < T; s++) {
    for (int g = wave - NE0; g < 8; g += NE) {
      int64_t row = (int64_t)s * n + table0 + 4 * g;
      if (r < 4) {
        if (NT) {
          u32x4 *dst = reinterpret_cast<u32x4 *>(obs + row * 480 + r * 480 + ch * 32);
          u32x4 vv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(vv, dst);
          __builtin_nontemporal_store(vv, dst + 1);
        } else {
          uint4 *dst = reinterpret_cast<uint4 *>(obs + row * 480 + r * 480 + ch * 32);
          dst[0] = v;
          dst[1] = v;
        }
      }
      if (lane < 38) reinterpret_cast<uint32_t *>(mask + row * 38)[lane] = v.w;
      v.x += 1;  // keep the loop from collapsing
      if (mode >= 2) {  // dependent VALU work between the stores (mode = number of 10-op rounds)
        uint32_t y = v.y;
        for (int q = 0; q < mode; q++) {
#pragma unroll
          for (int z = 0; z < 10; z++) y = y * 0x9E3779B1u + (y >> 7);
        }
        v.y = y;
      }
    }
  }
}
int main(int argc, char **argv) {
  const int64_t n = 8192; const int T = 32;
  uint8_t *obs, *mask;
  hipMalloc(&obs, n * T * 480); hipMalloc(&mask, n * T * 38 + 64);
  float *cols; hipMalloc(&cols, n * T * 17 + 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int cfg = 0; cfg < 11; cfg++) {
    float best = 1e9;
    for (int it = 0; it < 20; it++) {
      hipEventRecord(a);
      if (cfg == 0) hipLaunchKernelGGL((k_store<11, 3>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 1) hipLaunchKernelGGL((k_store<8, 0>), dim3(256), dim3(8 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 2) hipLaunchKernelGGL((k_store<16, 0>), dim3(256), dim3(16 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 3) hipLaunchKernelGGL((k_store<4, 0>), dim3(256), dim3(4 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 4) hipLaunchKernelGGL((k_store<11, 3>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 5);
      if (cfg == 5) hipLaunchKernelGGL((k_store<11, 3>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 10);
      if (cfg == 6) hipLaunchKernelGGL((k_store<11, 3>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 20);
      if (cfg == 7) hipLaunchKernelGGL((k_store<7, 3>), dim3(256), dim3(7 * 64), 0, 0, obs, mask, n, T, 10);
      if (cfg == 8) hipLaunchKernelGGL((k_store<11, 3, 1>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 9) hipLaunchKernelGGL((k_store<16, 0, 1>), dim3(256), dim3(16 * 64), 0, 0, obs, mask, n, T, 1);
      if (cfg == 10) hipLaunchKernelGGL((k_store<11, 3>), dim3(256), dim3(11 * 64), 0, 0, obs, mask, n, T, 1, cols);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    double bytes = (double)n * T * 518;
    printf("cfg %d: %.1f us  %.0f GB/s\n", cfg, best * 1e3, bytes / best / 1e6);
  }
  return 0;
}
