// heads_dw_role.hpp — the weight-gradient role of the 39-column head's backward pass (dW_h / db_h partials per batch split on the
// matrix cores + the extra workgroups that reduce the step's statistics / Gram partials), as a device function a kernel can give to
// its surplus workgroups.  It is NOT on the backward chain (nothing needs its outputs before the sums at the end of the step), so
// it rides with a launch that is: k_relu_bwd_tiles4_heads_dw (csrc/ppo_heads.hpp) or the dh GEMM (csrc/brl_mlp_gemm.hip).
// src/update.py:86-178 (the last Linear layers' weight gradients under autograd).
#pragma once
#include <stdint.h>

#ifndef BRL_NUM_ACTIONS
#define BRL_NUM_ACTIONS 38
#endif
constexpr int HD_NOUT = BRL_NUM_ACTIONS + 1;   // 39
constexpr int HD_GRAM = BRL_NUM_ACTIONS * BRL_NUM_ACTIONS;   // 1444
constexpr int HB_JT = 64;       // role A: columns of h per workgroup
typedef float hd_f32x4 __attribute__((ext_vector_type(4)));

struct HeadsBwdArgs {
  const float *dheads;   // [B, 39]
  const float *h;        // [B, H] the last hidden layer's output
  int64_t ldh;
  const float *Wh;       // [39, H]
  int64_t B;
  int H;                 // % 256 == 0
  int act;
  int nsplit;            // role A: batch splits
  int rows_per_split;    // ceil(B / nsplit) <= 64
  float *dWh_partials;   // [nsplit][39 * H]
  float *dbh_partials;   // [nsplit][39]
  float *dh;             // [B, H]: d(loss)/d(pre-activation of the last hidden layer)
  float *tile_sums;      // [ceil(B / 16)][H]: its column sums per 16-row tile (bias gradient)
  int blocks_a;          // (H / 64) * nsplit
  // optional (gram_sums != NULL): the step's statistics inputs reduced HERE, by extra workgroups of the dW launch, into row
  // *row_index of per-update buffers — brl_ppo_stats_rows turns all rows of an update into log rows with one launch at its end
  const float *loss_partials;   // [ngroups][8]   from k_heads_loss
  const float *gram_partials;   // [ngroups][1444]
  int ngroups;
  const int32_t *row_index;
  float *stat_sums;             // [rows][8]
  float *gram_sums;             // [rows][1444]
};
constexpr int HB_GRAM_BLOCKS = 91;  // 91 x 256 threads >= 16 x (1444 Gram entries + 8 statistics)

__device__ __forceinline__ void heads_bwd_dw_block(const HeadsBwdArgs &A, const int block) {
  __shared__ __attribute__((aligned(16))) float dh_s[64][HD_NOUT + 1];   // d(heads) rows of this workgroup (pad: 40 floats)
  __shared__ __attribute__((aligned(16))) float h_s[64][64];             // the split's tile of h
  const int tid = (int)threadIdx.x;
  if (block >= A.blocks_a) {
    // ---- extra workgroups: Gram matrix and statistics sums of this step.  Latency-bound (256 partial rows per entry at batch
    // 1024), so every entry is summed by SIXTEEN threads (a sixteenth of the rows each, all loads in flight), combined in a
    // fixed order: deterministic
    __shared__ float gs_part[256];
    const int64_t row = *A.row_index;
    const int gid = (block - A.blocks_a) * 256 + tid;
    const int e = gid >> 4, part = gid & 15;
    const int chunk = (A.ngroups + 15) / 16, i0 = part * chunk, i1 = (i0 + chunk < A.ngroups) ? i0 + chunk : A.ngroups;
    float s = 0.0f;
    if (e < HD_GRAM) {
      for (int i = i0; i < i1; i += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = A.gram_partials[(int64_t)((i + u < i1) ? i + u : i0) * HD_GRAM + e];
#pragma unroll
        for (int u = 0; u < 16; u++) s += (i + u < i1) ? v[u] : 0.0f;
      }
    } else if (e < HD_GRAM + 8) {
      const int k = e - HD_GRAM;
      for (int i = i0; i < i1; i++) s += A.loss_partials[(int64_t)i * 8 + k];
    }
    gs_part[tid] = s;
    __syncthreads();
    if (part == 0) {
      float tot = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; k++) tot += gs_part[tid + k];   // fixed order
      if (e < HD_GRAM) A.gram_sums[row * HD_GRAM + e] = tot;
      else if (e < HD_GRAM + 8) A.stat_sums[row * 8 + (e - HD_GRAM)] = tot;
    }
    return;
  }
  {
    // ---- role A: dW_h[n][j] = sum_b d(heads)[b][n] h[b][j] over this split's rows; thread = (column j, head group ng)
    const int jt = block % (A.H / HB_JT), sp = block / (A.H / HB_JT);
    const int64_t b0 = (int64_t)sp * A.rows_per_split;
    const int64_t left = A.B - b0;
    const int nb = (int)((left < A.rows_per_split) ? (left > 0 ? left : 0) : A.rows_per_split);
    // the split's 64 x 64 tile of h and its d(heads) rows go through LDS; ALL of the thread's global loads are issued
    // before anything waits (each byte is read once: the kernel is bound by load latency, not by bytes)
    float4 hv4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int rr = (tid >> 4) + 16 * u;   // 16 threads per row of 64 floats
      hv4[u] = *reinterpret_cast<const float4 *>(A.h + (b0 + ((rr < nb) ? rr : 0)) * A.ldh + jt * HB_JT + 4 * (tid & 15));
      if (rr >= nb) hv4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int DPT = (64 * HD_NOUT + 255) / 256;   // d(heads) elements per thread
    float dv[DPT];
#pragma unroll
    for (int k = 0; k < DPT; k++) {
      const int e = tid + 256 * k;
      dv[k] = (e < nb * HD_NOUT) ? A.dheads[b0 * HD_NOUT + e] : 0.0f;   // (the split's rows are contiguous)
    }
#pragma unroll
    for (int k = 0; k < DPT; k++) {
      const int e = tid + 256 * k;
      if (e < 64 * HD_NOUT) {
        const int rr = e / HD_NOUT, c = e - rr * HD_NOUT;
        dh_s[rr][c] = dv[k];   // rows past the split's end: zeros
      }
    }
    if (tid < 64) dh_s[tid][HD_NOUT] = 0.0f;   // the 40th "head" (group 3 has 9 real ones)
#pragma unroll
    for (int u = 0; u < 4; u++) *reinterpret_cast<float4 *>(&h_s[(tid >> 4) + 16 * u][4 * (tid & 15)]) = hv4[u];
    __syncthreads();
    // D[n][j] = sum_b d(heads)[b][n] h[b][j] on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32): wave w owns columns
    // 16 w .. 16 w + 15 of the tile and all three 16-head blocks; lane (c, kq): A row / B column c, K = 4 s + kq in step s.
    // (As plain FMAs with the operands broadcast from LDS this role was LDS-bound: 6 LDS reads per 10 FMAs.)
    const int lane = tid & 63, w = tid >> 6, c = lane & 15, kq = lane >> 4;
    hd_f32x4 acc[3];
#pragma unroll
    for (int nbk = 0; nbk < 3; nbk++) acc[nbk] = hd_f32x4{0.f, 0.f, 0.f, 0.f};
    int ncol[3];
#pragma unroll
    for (int nbk = 0; nbk < 3; nbk++) ncol[nbk] = (16 * nbk + c < HD_NOUT) ? 16 * nbk + c : HD_NOUT;   // (column 39 = the zero pad)
#pragma unroll 4
    for (int st = 0; st < 16; st++) {   // rows past the split's end are zero in both images
      const float bval = h_s[4 * st + kq][16 * w + c];
#pragma unroll
      for (int nbk = 0; nbk < 3; nbk++)
        acc[nbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh_s[4 * st + kq][ncol[nbk]], bval, acc[nbk], 0, 0, 0);
    }
    // accumulator register q of lane (c, rq): D[head 16 nbk + 4 rq + q][column 16 w + c]
#pragma unroll
    for (int nbk = 0; nbk < 3; nbk++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = 16 * nbk + 4 * kq + q;
        if (n < HD_NOUT) A.dWh_partials[(int64_t)sp * HD_NOUT * A.H + (int64_t)n * A.H + jt * HB_JT + 16 * w + c] = acc[nbk][q];
      }
    if (jt == 0 && tid < HD_NOUT) {   // db_h partial: column sums of d(heads) over the split, in row order
      float s = 0.0f;
      for (int rr = 0; rr < nb; rr++) s += dh_s[rr][tid];
      A.dbh_partials[sp * HD_NOUT + tid] = s;
    }
  }
}

