// How fast can a CU pull its two operand panels of a 1024^3 product (64 rows x K floats each, 524 KB per workgroup, every panel
// shared by 16 workgroups: an L2 / Infinity-Cache read stream, nothing from HBM after the first launch) into REGISTERS?
// One workgroup per 64 x 64 tile (mg::tile_of's XCD-aware mapping), no arithmetic but an XOR fold, by access shape:
//   shape 0 "full lines":  a wave instruction = 8 rows x 128 contiguous bytes (lane l: row l >> 3, 16 bytes at 16 (l & 7))
//   shape 1 "fragment":    a wave instruction = 32 rows x 2 x 16 bytes (lane (r, h): row r, 16 bytes at 16 h) — the MFMA operand order
//   shape 2 "fragment 64": a wave instruction = 32 rows x 2 x 16 bytes where the two pieces of a row are 64 bytes apart
// waves per workgroup 4 or 8 (K divided among the waves), DEPTH loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I brl_amd/csrc -o scripts/micro/l2_panel_bw scripts/micro/l2_panel_bw.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../../brl_amd/csrc/mlp_gemm.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int WAVES, int DEPTH>
__global__ __launch_bounds__(64 * WAVES) void k_panel(const float *A, const float *B, int K, int ld, unsigned *out) {
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  mg::tile_of((int)blockIdx.x, (int)gridDim.x, 16, 16, tm, tn);
  const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), (short)0, 0x7FFFFFFF, 0x00020000);
  // a "unit" = 128 bytes of K (32 floats) of all 64 rows of one operand = 8 KB = 8 wave instructions of 1 KB
  // units of an operand: K / 32; wave w takes units w, w + WAVES, ...
  unsigned fold = 0;
  const int units = K / 32;
  f32x4 r[DEPTH];
  int issued = 0, used = 0;
  // instruction i (0..7) of a unit, by shape -> byte offset of the lane
  auto voff = [&](int row0, int i) -> uint32_t {
    if (SHAPE == 0) return (uint32_t)((row0 + 8 * i + (lane >> 3)) * ld * 4 + 16 * (lane & 7));
    if (SHAPE == 1) return (uint32_t)((row0 + 32 * (i >> 2) + (lane & 31)) * ld * 4 + 32 * (i & 3) + 16 * (lane >> 5));
    return (uint32_t)((row0 + 32 * (i >> 2) + (lane & 31)) * ld * 4 + 16 * (i & 3) + 64 * (lane >> 5));
  };
  const int total = (units / WAVES) * 16;   // instructions of this wave: 8 of A + 8 of B per unit (a multiple of DEPTH: branch-free loop)
  // (no condition around a load: hipcc then branches around it and waits vmcnt(0) — cdna_hip_programming.md §5 trap (c))
  auto issue = [&](int d, int n) __attribute__((always_inline)) {
    const int nn = n < total ? n : total - 1;
    const int u = w + WAVES * (nn >> 4), i = nn & 15;
    const uint32_t vo = (i < 8) ? voff(tm * 64, i & 7) : voff(tn * 64, i & 7);
    const f32x4 va = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa, (int)vo, 128 * u, 0));
    (void)va;
    r[d] = va;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(d, d);
  for (int n = 0; n < total; n += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const f32x4 v = r[d];
      fold ^= __float_as_uint(v.x) ^ __float_as_uint(v.y) ^ __float_as_uint(v.z) ^ __float_as_uint(v.w);
      issue(d, n + d + DEPTH);
    }
  }
  (void)issued; (void)used;
  if (fold == 0x12345678u) out[blockIdx.x * 64 * WAVES + tid] = fold;
}

template <int SHAPE, int WAVES, int DEPTH>
static void run(const float *A, const float *B, unsigned *out, hipStream_t s, const char *name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_panel<SHAPE, WAVES, DEPTH>), dim3(256), dim3(64 * WAVES), 0, s, A, B, 1024, 1024, out);
  std::vector<double> t;
  for (int r = 0; r < 5; r++) {
    float ms;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((k_panel<SHAPE, WAVES, DEPTH>), dim3(256), dim3(64 * WAVES), 0, s, A, B, 1024, 1024, out);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms * 1e3 / 200);
  }
  std::sort(t.begin(), t.end());
  const double us = t[2];
  printf("%-14s waves %d depth %2d: %6.2f us per launch = %5.1f GB/s per CU (524 KB per workgroup), %5.2f TB/s chip-wide\n", name, WAVES, DEPTH, us,
         524288.0 / (us * 1e-6) / 1e9, 256 * 524288.0 / (us * 1e-6) / 1e12);
}

int main() {
  float *A, *B;
  unsigned *out;
  CK(hipMalloc(&A, 1024 * 1024 * 4)); CK(hipMalloc(&B, 1024 * 1024 * 4)); CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMemset(A, 1, 1024 * 1024 * 4)); CK(hipMemset(B, 2, 1024 * 1024 * 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  run<0, 4, 8>(A, B, out, s, "full lines");
  run<0, 4, 16>(A, B, out, s, "full lines");
  run<0, 4, 32>(A, B, out, s, "full lines");
  run<0, 8, 8>(A, B, out, s, "full lines");
  run<0, 8, 16>(A, B, out, s, "full lines");
  run<0, 8, 32>(A, B, out, s, "full lines");
  run<1, 4, 16>(A, B, out, s, "fragment");
  run<1, 4, 32>(A, B, out, s, "fragment");
  run<1, 8, 16>(A, B, out, s, "fragment");
  run<1, 8, 32>(A, B, out, s, "fragment");
  run<2, 4, 32>(A, B, out, s, "fragment 64");
  run<2, 8, 32>(A, B, out, s, "fragment 64");
  return 0;
}
