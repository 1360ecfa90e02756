// Store-pattern probe: the rollout's obs/mask bytes written with different lane->address mappings.
//   P0: lane (r,ch) writes 2 x 16 B at row r, byte 32 ch (each instruction covers every other 16 B)
//   P1: lane i writes 16 B at 16 i, two instructions cover [0,960) and [960,1920) of the group's 4 rows
//   P2: 64 lanes x 16 B, two instructions = 2 KB aligned (upper bound, 6.7 % more bytes)
//   P3: P1 with dwordx2 stores (4 instructions of 480 B)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int P>
__global__ __launch_bounds__(512) void k_store(uint8_t *obs, uint8_t *mask, int64_t n, int T, int nvb = 256) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane / 15, ch = lane % 15;
  uint4 v = make_uint4(lane, wave, 1, 0x01010101u);
  // nvb virtual blocks of 32 tables spread over gridDim.x real ones
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x)
  for (int s = 0; s < T; s++) {
    const int64_t table0 = (int64_t)((vb % 8) * (nvb / 8) + vb / 8) * 32;
    const int g = wave;
    int64_t row = (int64_t)s * n + table0 + 4 * g;
    uint8_t *base = obs + row * 480;
    if (P == 0) {
      if (r < 4) {
        uint4 *dst = reinterpret_cast<uint4 *>(base + r * 480 + ch * 32);
        dst[0] = v;
        dst[1] = v;
      }
    } else if (P == 1) {
      if (lane < 60) {
        *reinterpret_cast<uint4 *>(base + 16 * lane) = v;
        *reinterpret_cast<uint4 *>(base + 960 + 16 * lane) = v;
      }
    } else if (P == 2) {
      uint8_t *b2 = obs + ((int64_t)s * n / 4 + table0 / 4 + g) * 2048;
      *reinterpret_cast<uint4 *>(b2 + 16 * lane) = v;
      *reinterpret_cast<uint4 *>(b2 + 1024 + 16 * lane) = v;
    } else if (P == 3) {
      if (lane < 60) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<uint2 *>(base + 480 * q + 8 * lane) = make_uint2(v.x, v.y);
      }
    }
    if (lane < 38) reinterpret_cast<uint32_t *>(mask + row * 38)[lane] = v.w;
    v.x += 1;
  }
}
int main() {
  const int64_t n = 8192; const int T = 33;
  uint8_t *obs, *mask;
  hipMalloc(&obs, 4 * n * T * 512 + 4096); hipMalloc(&mask, 4 * n * T * 38 + 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int cfg = 0; cfg < 8; cfg++) {
    float best = 1e9;
    for (int it = 0; it < 30; it++) {
      hipEventRecord(a);
      if (cfg == 0) hipLaunchKernelGGL((k_store<0>), dim3(256), dim3(512), 0, 0, obs, mask, n, T);
      if (cfg == 1) hipLaunchKernelGGL((k_store<1>), dim3(256), dim3(512), 0, 0, obs, mask, n, T);
      if (cfg == 2) hipLaunchKernelGGL((k_store<2>), dim3(256), dim3(512), 0, 0, obs, mask, n, T);
      if (cfg == 3) hipLaunchKernelGGL((k_store<3>), dim3(256), dim3(512), 0, 0, obs, mask, n, T);
      if (cfg == 4) hipLaunchKernelGGL((k_store<1>), dim3(128), dim3(512), 0, 0, obs, mask, n, T, 256);
      if (cfg == 5) hipLaunchKernelGGL((k_store<1>), dim3(64), dim3(512), 0, 0, obs, mask, n, T, 256);
      if (cfg == 6) hipLaunchKernelGGL((k_store<1>), dim3(512), dim3(512), 0, 0, obs, mask, 2 * n, T, 512);
      if (cfg == 7) hipLaunchKernelGGL((k_store<1>), dim3(1024), dim3(512), 0, 0, obs, mask, 4 * n, T, 1024);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    double bytes = (double)n * T * 518 * (cfg == 6 ? 2 : (cfg == 7 ? 4 : 1));
    printf("cfg%d: %.1f us  %.0f GB/s (518 B/row basis)\n", cfg, best * 1e3, bytes / (best * 1e-3) / 1e9);
  }
  return 0;
}
