// brl_infer16.hip — translation unit of libbrl_hip.so: the opt-in 16-bit inference layer of the policy MLP (k_linear16,
// csrc/mlp_infer.hpp: y = act(x W^T + b) in bf16 / fp16, optionally with the heads' share in the epilogue) and the observation
// casts in front of it (include/brl_hip.h: brl_linear_act[_heads], brl_obs_cast[_rows]).  fp32 — the reference's precision — is the
// default everywhere; nothing here is on the default path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "handle.hpp"
#include "mlp_infer.hpp"   // k_linear16: one bf16 / fp16 layer of the policy MLP (inference)

// observation bytes (0/1) -> the network's input dtype: 16 bytes in, 16 elements out per thread
// (src/roll_out.py:75 `last_obs.astype(jnp.float32)`; torch's generic bool->bf16 copy takes 15 us for 3.9 MB)
template <int FMT>  // 0: f32, 1: bf16 (0x3F80), 2: f16 (0x3C00)
__device__ __forceinline__ void obs_cast16(const uint4 v, void *out, int64_t i) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  if (FMT == 0) {
    float4 *o = reinterpret_cast<float4 *>(out) + 4 * i;
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = make_float4((w[k] & 1u) ? 1.0f : 0.0f, (w[k] & 0x100u) ? 1.0f : 0.0f, (w[k] & 0x10000u) ? 1.0f : 0.0f,
                         (w[k] & 0x1000000u) ? 1.0f : 0.0f);
  } else {
    const uint32_t one = (FMT == 1) ? 0x3F80u : 0x3C00u;
    uint4 *o = reinterpret_cast<uint4 *>(out) + 2 * i;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // bytes (b0,b1,b2,b3) of a dword -> halves (b0,b1) and (b2,b3)
      h[2 * k] = ((w[k] & 1u) ? one : 0u) | ((w[k] & 0x100u) ? (one << 16) : 0u);
      h[2 * k + 1] = ((w[k] & 0x10000u) ? one : 0u) | ((w[k] & 0x1000000u) ? (one << 16) : 0u);
    }
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
  }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_obs_cast(const uint4 *in, void *out, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  obs_cast16<FMT>(in[i], out, i);
}

// the same for the rows rows[0..m) of `in` only (out row r = in row rows[r]): the forwards of an evaluator run on the boards
// that are still playing
template <int FMT>
__global__ __launch_bounds__(256) void k_obs_cast_rows(const uint4 *in, const int64_t *rows, void *out, int64_t m16) {
  constexpr int PER_ROW = BRL_OBS_SIZE / 16;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m16) return;
  const int64_t r = i / PER_ROW;
  obs_cast16<FMT>(in[rows[r] * PER_ROW + (i - r * PER_ROW)], out, i);
}

extern "C" int brl_obs_cast(brl_handle *h, const uint8_t *obs, int64_t n, void *out, int fmt, void *stream) {
  COMMON(h, n);
  NEED(obs && out, "NULL obs / out");
  NEED(fmt >= 0 && fmt <= 2, "fmt");
  const int64_t n16 = n * (BRL_OBS_SIZE / 16);
  const dim3 grid((unsigned)((n16 + 255) / 256)), block(256);
  if (fmt == 0) hipLaunchKernelGGL(k_obs_cast<0>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  else if (fmt == 1) hipLaunchKernelGGL(k_obs_cast<1>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  else hipLaunchKernelGGL(k_obs_cast<2>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, out, n16);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_obs_cast_rows(brl_handle *h, const uint8_t *obs, const int64_t *rows, int64_t m, void *out, int fmt,
                                 void *stream) {
  COMMON(h, m);
  NEED(obs && rows && out, "NULL obs / rows / out");
  NEED(fmt >= 0 && fmt <= 2, "fmt");
  const int64_t m16 = m * (BRL_OBS_SIZE / 16);
  const dim3 grid((unsigned)((m16 + 255) / 256)), block(256);
  if (fmt == 0) hipLaunchKernelGGL(k_obs_cast_rows<0>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  else if (fmt == 1) hipLaunchKernelGGL(k_obs_cast_rows<1>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  else hipLaunchKernelGGL(k_obs_cast_rows<2>, grid, block, 0, (hipStream_t)stream, (const uint4 *)obs, rows, out, m16);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
static int lin16_attr(int fmt) {
  static bool done[3] = {false, false, false};
  if (!done[fmt]) {   // 144 KB of dynamic LDS: above the default 64 KB limit
    if (fmt == 1) HIP_TRY(hipFuncSetAttribute((const void *)lin16::k_linear16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::LDS_BYTES));
    else HIP_TRY(hipFuncSetAttribute((const void *)lin16::k_linear16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::LDS_BYTES));
    done[fmt] = true;
  }
  return BRL_OK;
}

static int lin16_store_mode() {
  static int store_mode = -1;
  if (store_mode < 0) {
    // y leaves write-through by default: measured in the bf16 graph rollout, 8192 tables: 12.47 ms against 13.30 (plain
    // stores: the dirty lines are written back when the kernel ends) and 13.01 (non-temporal); BRL_LIN16_STORE=0/1/2 for A/B
    const char *e = getenv("BRL_LIN16_STORE");
    store_mode = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : 2;
  }
  return store_mode;
}

#ifdef LIN16_TIMING   // scripts/time_linear16.py --stamps: 4 shader-clock stamps per workgroup
static unsigned long long *g_lin16_dbg = nullptr;
extern "C" void brl_lin16_set_dbg(void *p) { g_lin16_dbg = (unsigned long long *)p; }
#endif
static int linear_act_impl(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                           int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w, int64_t ld_head_w,
                           int n_heads, float *head_part, int64_t head_part_ld, int64_t head_part_stride, void *stream) {
  COMMON(h, m);
  NEED(x && w && (y || head_part), "NULL x / w / y");
  if (head_part) {
    NEED(head_w && n_heads >= 1 && n_heads <= 48 && ld_head_w >= n_out && ld_head_w % 8 == 0 && (((uintptr_t)head_w) & 15) == 0,
         "head_w / n_heads (<= 48) / ld_head_w");
    NEED(head_part_ld >= ((n_heads + 3) & ~3) && head_part_ld % 4 == 0 && head_part_stride >= m * head_part_ld && head_part_stride % 4 == 0
         && (((uintptr_t)head_part) & 15) == 0, "head_part (16-byte aligned) / head_part_ld (% 4, >= n_heads rounded up to 4) / head_part_stride");
  }
  if (!y) ldy = n_out;
  NEED(fmt == 1 || fmt == 2, "fmt (1 = bf16, 2 = fp16)");
  NEED(n_out > 0 && n_out % lin16::BN == 0, "n_out % 128");
  NEED(k >= 8 && k % 8 == 0, "k % 8");
  NEED(ldx >= k && ldw >= k && ldy >= n_out && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "ldx / ldw / ldy");
  NEED((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "x / w / y not 16-byte aligned");
  NEED(m * ldx * 2 < ((int64_t)1 << 32) && (int64_t)n_out * ldw * 2 < ((int64_t)1 << 32), "operand larger than 4 GB");
  NEED(m <= (int64_t)1 << 30, "m");
  if (int rc = lin16_attr(fmt)) return rc;
  lin16::Args A;
  memset(&A, 0, sizeof(A));
  A.x = (const uint16_t *)x; A.ldx = ldx;
  A.w = (const uint16_t *)w; A.ldw = ldw;
  A.bias = bias;
  A.y = (uint16_t *)y; A.ldy = ldy;
  A.M = (int)m; A.N = n_out; A.K = k;
  A.relu = relu;
  A.head_w = (const uint16_t *)head_w; A.ld_head_w = ld_head_w; A.n_heads = n_heads;
  A.head_part = head_part; A.head_part_ld = head_part_ld; A.head_part_stride = head_part_stride;
  A.store_mode = lin16_store_mode();
#ifdef LIN16_TIMING
  A.dbg = g_lin16_dbg;
#endif
  const int tiles = (int)((m + lin16::BM - 1) / lin16::BM) * (n_out / lin16::BN);
  if (fmt == 1) hipLaunchKernelGGL(lin16::k_linear16<1>, dim3(tiles), dim3(lin16::THREADS), lin16::LDS_BYTES, (hipStream_t)stream, A);
  else hipLaunchKernelGGL(lin16::k_linear16<2>, dim3(tiles), dim3(lin16::THREADS), lin16::LDS_BYTES, (hipStream_t)stream, A);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}

extern "C" int brl_linear_act(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                            int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, void *stream) {
  NEED(y != nullptr, "NULL y");
  return linear_act_impl(h, x, ldx, w, ldw, bias, y, ldy, m, n_out, k, relu, fmt, nullptr, 0, 0, nullptr, 0, 0, stream);
}

extern "C" int brl_linear_act_heads(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias,
                                  void *y, int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w,
                                  int64_t ld_head_w, int n_heads, float *head_part, int64_t head_part_ld,
                                  int64_t head_part_stride, void *stream) {
  NEED(head_part != nullptr, "NULL head_part");
  return linear_act_impl(h, x, ldx, w, ldw, bias, y, ldy, m, n_out, k, relu, fmt, head_w, ld_head_w, n_heads, head_part,
                         head_part_ld, head_part_stride, stream);
}
