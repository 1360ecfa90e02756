// handle.hpp — the library's handle and the host-side launch helpers every translation unit with handle-taking entry points shares.
#pragma once
#include "abi_common.hpp"   // HIP_TRY, NEED, brl_fail
#include "wave_common.hpp"

struct brl_handle {
  int device;
  int4 *lut_keys;
  int4 *lut_values;
  uint4 *lut_hands;
  int64_t lut_len;
  float *neg_log_n;
  uint64_t seed;
  uint64_t env_offset;
  DevCtx *ctx_dev;  // device mirror of (LUT, seed, env_offset), read by the policy sub-step
  int tables_per_wave;
  int ws;  // 1: wave-specialised fused rollout k_rollout_ws<32,12,1> (default); 0: k_rollout_random<K> (BRL_ROLLOUT_WS=0)
  int fs;  // 1: flag-synchronised k_rollout_fs where it applies (default); 0: always k_rollout_ws (BRL_ROLLOUT_FS=0)
};
static inline int fail(int code, const char *fmt, const char *detail) { return brl_fail(code, fmt, detail); }

static inline Rng rng_of(const brl_handle *h) { return Rng{(uint32_t)h->seed, (uint32_t)(h->seed >> 32)}; }
static inline LutRef lut_of(const brl_handle *h) { return LutRef{h->lut_keys, h->lut_values, (uint32_t)h->lut_len, h->lut_hands}; }
static inline unsigned wave_grid(int64_t n, int K) {
  int64_t per_block = (int64_t)WAVES_PER_BLOCK * K;
  return (unsigned)((n + per_block - 1) / per_block);
}
static inline unsigned thread_grid(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

#define LAUNCH_K(h, kernel, n, stream, ...)                                                                  \
  do {                                                                                                       \
    hipStream_t _s = (hipStream_t)(stream);                                                                  \
    switch ((h)->tables_per_wave) {                                                                          \
      case 1: hipLaunchKernelGGL(kernel<1>, dim3(wave_grid(n, 1)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      case 2: hipLaunchKernelGGL(kernel<2>, dim3(wave_grid(n, 2)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      case 8: hipLaunchKernelGGL(kernel<8>, dim3(wave_grid(n, 8)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL(kernel<4>, dim3(wave_grid(n, 4)), dim3(BLOCK_THREADS), 0, _s, __VA_ARGS__); break; \
    }                                                                                                        \
    HIP_TRY(hipGetLastError());                                                                              \
  } while (0)

#define COMMON(h, n)                 \
  NEED((h) != nullptr, "handle");    \
  NEED((n) >= 0, "n");               \
  if ((n) == 0) return BRL_OK;       \
  HIP_TRY(hipSetDevice((h)->device))
