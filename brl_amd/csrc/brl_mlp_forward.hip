// brl_mlp_forward.hip — translation unit of libbrl_hip.so: the policy network's fp32 forward for selected rows with the library's
// own kernels end to end, behind ONE C-ABI call (brl_mlp_forward_rows, include/brl_hip.h): observation bytes of the selected
// boards -> float, the hidden layers through brl_mlp_gemm (bias + activation in the epilogue, nn.Linear's own weight layout), the
// 38 + 1 heads with the scatter back to the boards' rows.  Written for the small-batch iterations of the evaluators
// (src/evaluation.py:120-197: the last boards of a duplicate evaluation), which are bound by host launches, not by the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abi_common.hpp"

namespace {

constexpr int NHEADS = BRL_NUM_ACTIONS + 1;   // 38 logits + the value

// x[r] = float(obs[rows[r]]): one 128-thread workgroup per row, 4 observation bytes -> one 16-byte store
__global__ __launch_bounds__(128) void k_obs_rows_f32(const uint8_t *obs, const int64_t *rows, float *x) {
  const int64_t r = blockIdx.x, src = rows ? rows[r] : r;
  const int t = (int)threadIdx.x;
  if (t < BRL_OBS_SIZE / 4) {
    const uint32_t w = reinterpret_cast<const uint32_t *>(obs + src * BRL_OBS_SIZE)[t];
    reinterpret_cast<float4 *>(x + r * BRL_OBS_SIZE)[t] =
        make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24));
  }
}

// The heads: out[rows[r]][hd] = h[r] . w_hd + b_hd for hd = 0..38 (38 actor rows, then the critic row).  Workgroup (x, y) owns 4
// rows of h (in registers: lane l holds columns 4 l + 256 j .. + 3) and the 13 heads 13 y .. 13 y + 12; wave w takes its heads
// w, w + 4, w + 8 (, w + 12): ALL of their weights are requested before anything is used (the kernel is a chain of L2 round trips,
// not arithmetic), 16 partial dot products per lane, reduced across the wave by a halving butterfly (16 + 8 + 4 + 2 + 2 shuffles
// instead of 16 x 6).
constexpr int HR = 4, HG = 13, HPW = 4;   // rows per workgroup, heads per workgroup, heads per wave (at most)
__global__ __launch_bounds__(256) void k_heads_rows(const float *h, int64_t ldh, int hidden, const float *actor_w, const float *actor_b,
                                                    const float *critic_w, const float *critic_b, const int64_t *rows, int64_t m,
                                                    float *out, int64_t ldo) {
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * HR;
  const int h0 = (int)blockIdx.y * HG;
  float4 wv[HPW][4], hv[HR][4];
#pragma unroll
  for (int i = 0; i < HPW; i++) {
    const int l = w + 4 * i, hd = h0 + l;
    const bool ok = l < HG && hd < NHEADS;
    const float *wr = (hd < BRL_NUM_ACTIONS) ? actor_w + (int64_t)(ok ? hd : 0) * hidden : critic_w;
    // (unconditional loads from clamped addresses, zeros selected afterwards: a guard around a load makes hipcc branch around it
    //  and wait for each one — 32 memory round trips in a row instead of one, 12.6 us instead of 5)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = 4 * lane + 256 * j;
      wv[i][j] = *reinterpret_cast<const float4 *>(wr + ((k < hidden) ? k : 0));
    }
  }
#pragma unroll
  for (int r = 0; r < HR; r++) {
    const int64_t row = (r0 + r < m) ? r0 + r : m - 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = 4 * lane + 256 * j;
      hv[r][j] = *reinterpret_cast<const float4 *>(h + row * ldh + ((k < hidden) ? k : 0));
    }
  }
  // columns beyond `hidden` and heads this wave does not own contribute zeros (the weights are zeroed: one side is enough)
#pragma unroll
  for (int i = 0; i < HPW; i++) {
    const int l = w + 4 * i;
    const bool ok = l < HG && h0 + l < NHEADS;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (!ok || 4 * lane + 256 * j >= hidden) wv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float v[HPW * HR];   // [head i][row r]
#pragma unroll
  for (int i = 0; i < HPW; i++)
#pragma unroll
    for (int r = 0; r < HR; r++) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        s = fmaf(hv[r][j].x, wv[i][j].x, s);
        s = fmaf(hv[r][j].y, wv[i][j].y, s);
        s = fmaf(hv[r][j].z, wv[i][j].z, s);
        s = fmaf(hv[r][j].w, wv[i][j].w, s);
      }
      v[i * HR + r] = s;
    }
  // halving butterfly: after the step with lane bit `off`, a lane keeps the half of the values its bit selects; four steps leave one
  // value per lane (index = lane bits 5..2), two more add what lanes differing in bits 1..0 hold — a fixed order
#pragma unroll
  for (int step = 0; step < 4; step++) {
    const int off = 32 >> step, half = (HPW * HR / 2) >> step;
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
      const float send = hi ? v[i] : v[i + half], keep = hi ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  float s = v[0];
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  if ((lane & 3) == 0) {
    const int idx = lane >> 2, i = idx / HR, r = idx % HR, l = w + 4 * i, hd = h0 + l;
    if (l < HG && hd < NHEADS && r0 + r < m) {
      const int64_t row = r0 + r;
      out[(rows ? rows[row] : row) * ldo + hd] = s + ((hd < BRL_NUM_ACTIONS) ? actor_b[hd] : critic_b[0]);
    }
  }
}

}  // namespace

// (K divided over several workgroups per tile — tiles x slices = 512 workgroups, the last slice to arrive adding the partial tiles
//  in index order — was built and measured for the layers of m <= 512 rows: 11.3 us per layer against 12.5 at m = 256, slower at
//  m = 512: a layer this small is a chain of ~5 dependent memory round trips either way; profiles/r04/r04_experiments.txt section 11.)
extern "C" int brl_mlp_forward_rows(int device, const brl_mlp_ref *net, const uint8_t *obs, const int64_t *rows, int64_t m,
                                    float *scratch, int64_t scratch_len, float *out, int64_t ldo, void *stream) {
  NEED(net && obs && scratch && out && m > 0, "net / obs / scratch / out / m");
  NEED(net->nlayers >= 1 && net->nlayers <= 8, "nlayers (1..8)");
  NEED(net->in_features == BRL_OBS_SIZE, "in_features (480)");
  NEED(net->hidden > 0 && net->hidden % 4 == 0 && net->hidden <= 1024, "hidden (a multiple of 4, <= 1024)");
  NEED(net->act == 0 || net->act == 1, "act (0 ReLU, 1 tanh)");
  NEED(net->actor_w && net->actor_b && net->critic_w && net->critic_b, "NULL head arrays");
  for (int l = 0; l < net->nlayers; l++) NEED(net->w[l] && net->b[l], "NULL layer arrays");
  NEED(ldo >= NHEADS, "ldo (>= 39)");
  NEED(m < (1 << 24), "m below 2^24");
  const int64_t H = net->hidden;
  NEED(scratch_len >= m * (BRL_OBS_SIZE + 2 * H) && (((uintptr_t)scratch) & 15) == 0, "scratch: m * (480 + 2 * hidden) floats, 16-byte aligned");
  NEED((((uintptr_t)net->actor_w | (uintptr_t)net->critic_w) & 15) == 0, "head weights not 16-byte aligned");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  float *x = scratch, *act[2] = {scratch + m * BRL_OBS_SIZE, scratch + m * BRL_OBS_SIZE + m * H};
  hipLaunchKernelGGL(k_obs_rows_f32, dim3((unsigned)m), dim3(128), 0, s, obs, rows, x);
  const float *cur = x;
  int64_t k = BRL_OBS_SIZE;
  for (int l = 0; l < net->nlayers; l++) {
    float *dst = act[l & 1];
    const int rc = brl_mlp_gemm(device, BRL_GEMM_NT, BRL_GEMM_EPI_BIAS_ACT, cur, k, net->w[l], k, dst, H, m, H, k, net->act, net->b[l],
                                nullptr, 0, nullptr, nullptr, stream);
    if (rc) return rc;
    cur = dst;
    k = H;
  }
  hipLaunchKernelGGL(k_heads_rows, dim3((unsigned)((m + HR - 1) / HR), (NHEADS + HG - 1) / HG), dim3(256), 0, s, cur, H, (int)H,
                     net->actor_w, net->actor_b, net->critic_w, net->critic_b, rows, m, out, ldo);
  HIP_TRY(hipGetLastError());
  return BRL_OK;
}
