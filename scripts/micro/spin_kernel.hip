// spin_kernel.hip — a stand-in for an RCCL ring kernel in scripts/overlap_probe.py: `blocks` workgroups of 256 threads that hold their
// CUs for `usec` microseconds (constant 100 MHz clock), doing nothing.  What it models: the CUs a collective's channels occupy while a
// GEMM of the PPO step runs beside it; what it does not: the collective's HBM / xGMI traffic.  EXPERIMENT ONLY (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o scripts/micro/libspin.so scripts/micro/spin_kernel.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int spin_launch(int blocks, double usec, void *stream) {
  if (blocks <= 0 || usec <= 0) return 0;
  hipLaunchKernelGGL(k_spin, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)(usec * 100.0));
  return (int)hipGetLastError();
}
