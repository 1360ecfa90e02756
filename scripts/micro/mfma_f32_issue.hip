// What does ONE other instruction cost between two v_mfma_f32_16x16x4_f32 of the same wave (one wave per SIMD)?
// A 4-accumulator MFMA stream (the K loop of brl_amd/csrc/mlp_gemm.hpp) with a filler instruction behind every MFMA, every
// second, or every fourth; accumulators in VGPRs ("+v") or AGPRs ("+a").  Prints shader cycles per MFMA (ideal: 32).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/mfma_f32_issue scripts/micro/mfma_f32_issue.hip && ./scripts/micro/mfma_f32_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE, F_DSREAD128, F_DSWRITE128, F_GLOAD128, F_VMOV, F_DSREAD32, F_DSREAD64, F_SALU, F_BUFLOAD128, F_DSWRITE64, F_GLOAD_SADDR, F_VADD, F_WAITCNT, F_DSREAD128_OFF };

template <int FILL, int EVERY, bool AGPR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float *src, float *out, unsigned long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63, tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 64 * WAVES) lds[i] = src[i];
  __syncthreads();
  f32x4 acc[4];
  for (int i = 0; i < 4; i++) acc[i] = f32x4{0, 0, 0, 0};
  float a = src[lane], b = src[64 + lane];
  f32x4 r[4];
  for (int i = 0; i < 4; i++) r[i] = f32x4{1, 2, 3, 4};
  const float *gp = src + (size_t)(blockIdx.x * 64 * WAVES + tid) * 4;
  float *lp = lds + tid * 4;
  unsigned sacc = 0;
  unsigned vtmp[4] = {1u, 2u, 3u, 4u};
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, 1 << 24, 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
      __builtin_amdgcn_sched_barrier(0);
      if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
      __builtin_amdgcn_sched_barrier(0);
      if (m % EVERY == 0) {
        const int q = (m / EVERY) & 3;
        if (FILL == F_DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(r[q]) : "v"((unsigned)(size_t)(lp) & 0xFFFF) : "memory");
        if (FILL == F_DSREAD64) asm volatile("ds_read_b64 %0, %1" : "=v"(*(double *)&r[q]) : "v"((unsigned)(size_t)(lp) & 0xFFFF) : "memory");
        if (FILL == F_DSREAD32) asm volatile("ds_read_b32 %0, %1" : "=v"(r[q].x) : "v"((unsigned)(size_t)(lp) & 0xFFFF) : "memory");
        if (FILL == F_DSWRITE128) asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(size_t)(lp) & 0xFFFF), "v"(r[q]) : "memory");
        if (FILL == F_DSWRITE64) asm volatile("ds_write_b64 %0, %1" : : "v"((unsigned)(size_t)(lp) & 0xFFFF), "v"(*(double *)&r[q]) : "memory");
        if (FILL == F_GLOAD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(gp) : "memory");
        if (FILL == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r[q].x) : "v"(r[(q + 1) & 3].y));
        if (FILL == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (FILL == F_GLOAD_SADDR) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[q]) : "v"((unsigned)(tid * 16)), "s"(src) : "memory");
        if (FILL == F_BUFLOAD128) { i32x4 t = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, tid * 16, (it & 1023) * 1024, 0)); asm volatile("" : "+v"(t)); r[q] = __builtin_bit_cast(f32x4, t); }
        if (FILL == F_VADD) asm volatile("v_add_u32 %0, %1, %2" : "=v"(vtmp[q]) : "v"(vtmp[(q + 1) & 3]), "v"(lane));
        if (FILL == F_WAITCNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (FILL == F_DSREAD128_OFF) asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(r[q]) : "v"((unsigned)(size_t)(lp) & 0x3FFF) : "memory");
      }
    }
    if (FILL == F_DSREAD128 || FILL == F_DSREAD32 || FILL == F_DSREAD64 || FILL == F_DSWRITE128 || FILL == F_DSWRITE64)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (FILL == F_GLOAD128 || FILL == F_GLOAD_SADDR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + r[i].x + r[i].w;
  out[blockIdx.x * 64 * WAVES + tid] = s + sacc + vtmp[0] + vtmp[1] + vtmp[2] + vtmp[3];
  if (lane == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int FILL, int EVERY, bool AGPR, int WAVES>
static void run(const char *name, const float *src, float *out, unsigned long long *cyc) {
  const int iters = 400, blocks = 256;
  hipLaunchKernelGGL((k<FILL, EVERY, AGPR, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, out, cyc, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k<FILL, EVERY, AGPR, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, out, cyc, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 16);
  CK(hipMemcpy(h.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost));
  double m = 0, mx = 0;
  for (int b = 0; b < blocks; b++) for (int w = 0; w < WAVES; w++) { m += h[b * 16 + w]; if (h[b * 16 + w] > mx) mx = h[b * 16 + w]; }
  m /= blocks * WAVES;
  const double flop = 10.0 * blocks * WAVES * iters * 16.0 * 2048.0;
  printf("%-26s every %2d acc %s waves/CU %2d: %6.2f cycles per MFMA per wave (slowest wave %6.2f) => %5.1f extra per filler | wall %.1f us / launch, %.1f TFLOP/s\n", name,
         EVERY, AGPR ? "AGPR" : "VGPR", WAVES, m / (iters * 16.0), mx / (iters * 16.0), (m / (iters * 16.0) - 32.0) * EVERY, ms * 100.0, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float *src, *out;
  unsigned long long *cyc;
  CK(hipMalloc(&src, 1 << 24)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 256 * 16 * 8));
  std::vector<float> h(1 << 22);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 32768.0f - 1.0f;
  CK(hipMemcpy(src, h.data(), 1 << 24, hipMemcpyHostToDevice));
#define R(f, e, a, w) run<f, e, a, w>(#f, src, out, cyc)
  R(F_NONE, 1, false, 4); R(F_NONE, 1, false, 8); R(F_NONE, 1, false, 16); R(F_NONE, 1, false, 1);
  R(F_DSREAD128, 1, false, 4); R(F_DSREAD128_OFF, 1, false, 4); R(F_DSREAD128, 4, false, 4);
  R(F_DSWRITE128, 2, false, 4); R(F_DSWRITE128, 4, false, 4); R(F_DSWRITE64, 2, false, 4);
  R(F_GLOAD128, 4, false, 4); R(F_GLOAD_SADDR, 4, false, 4); R(F_BUFLOAD128, 4, false, 4); R(F_GLOAD_SADDR, 2, false, 4); R(F_BUFLOAD128, 2, false, 4);
  R(F_VMOV, 1, false, 4); R(F_VMOV, 4, false, 4); R(F_VADD, 1, false, 4); R(F_VADD, 4, false, 4); R(F_SALU, 1, false, 4); R(F_WAITCNT, 1, false, 4);
  // two waves per SIMD
  R(F_DSREAD128, 1, false, 8); R(F_DSWRITE128, 2, false, 8); R(F_GLOAD128, 4, false, 8); R(F_VADD, 1, false, 8);
  return 0;
}
