// policy_common.hpp — the categorical distribution over a table's 38 calls on the lanes that share the table (the policy
// sub-step of the rollouts and the evaluators' greedy step use the same device function) and the network-output conversions.
#pragma once
#include "wave_common.hpp"

// Masked categorical of one table on the 64 / K lanes that share it (lane l: table l % K, slot l / K): slot s holds the
// NI consecutive actions [s * NI, s * NI + NI).  Returns the chosen action and its log-probability on every lane of the
// table.  `cand`: the actions the distribution ranges over — the legal ones (masked policy, src/roll_out.py:27-29) or
// all 38 (unmasked / illegal-action-penalty policy, src/roll_out.py:33-39).
//   mode bit 0: 0 = pi.sample (inverse CDF in action order with the 24-bit uniform of `u32`), 1 = pi.mode (first max)
// network outputs as the GEMM wrote them: float (fmt 0), bf16 (1) or fp16 (2) -> float (exact conversions)
__device__ __forceinline__ float net_out(const void *base, int64_t idx, int fmt) {
  if (fmt == 0) return reinterpret_cast<const float *>(base)[idx];
  const uint16_t h = reinterpret_cast<const uint16_t *>(base)[idx];
  if (fmt == 1) return __uint_as_float((uint32_t)h << 16);
  _Float16 f16;
  __builtin_memcpy(&f16, &h, 2);
  return (float)f16;
}

__device__ __forceinline__ float net_cvt(uint32_t raw, int fmt) {  // what net_out makes of the bits it loaded
  if (fmt == 0) return __uint_as_float(raw);
  if (fmt == 1) return __uint_as_float(raw << 16);
  const uint16_t h = (uint16_t)raw;
  _Float16 f16;
  __builtin_memcpy(&f16, &h, 2);
  return (float)f16;
}

template <int K>
__device__ __forceinline__ int categorical(const void *logits, int64_t row_off, int fmt, bool valid, uint64_t cand, int mode,
                                           uint32_t u32, int lane, float &log_prob) {
  constexpr int LPT = 64 / K;
  constexpr int NI = (BRL_NUM_ACTIONS + LPT - 1) / LPT;
  const int tl = lane % K, slot = lane / K;
  float lg[NI], e[NI];
  bool ok[NI];
  float mx = -INFINITY;
  int amax = 64;
  // the lane's NI logits: unconditional loads (clamped index), the format decided ONCE around all of them — a select or a
  // format branch per element makes hipcc branch around every load and wait for each one (NI memory round trips)
  uint32_t raw[NI];
  const int64_t ro = valid ? row_off : 0;
  if (fmt == 0) {
#pragma unroll
    for (int i = 0; i < NI; i++) raw[i] = reinterpret_cast<const uint32_t *>(logits)[ro + min(slot * NI + i, BRL_NUM_ACTIONS - 1)];
  } else {
#pragma unroll
    for (int i = 0; i < NI; i++) raw[i] = reinterpret_cast<const uint16_t *>(logits)[ro + min(slot * NI + i, BRL_NUM_ACTIONS - 1)];
  }
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int a = slot * NI + i;
    const bool in = a < BRL_NUM_ACTIONS;
    lg[i] = (in && valid) ? net_cvt(raw[i], fmt) : 0.0f;
    ok[i] = in && ((cand >> (a & 63)) & 1ull);
    if (ok[i] && lg[i] > mx) {  // first maximum wins, like argmax
      mx = lg[i];
      amax = a;
    }
  }
#pragma unroll
  for (int off = K; off < 64; off <<= 1) {
    const float omx = __shfl_xor(mx, off, 64);
    const int oam = __shfl_xor(amax, off, 64);
    const bool take = (omx > mx) || (omx == mx && oam < amax);
    mx = take ? omx : mx;
    amax = take ? oam : amax;
  }
  amax = (amax >= BRL_NUM_ACTIONS) ? 0 : amax;  // (no finite candidate logit: NaN / -inf everywhere)
  float own = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; i++) {
    e[i] = ok[i] ? expf(lg[i] - mx) : 0.0f;
    own += e[i];
  }
  // inclusive scan of the slots' sums in action order
  float incl = own;
#pragma unroll
  for (int off = 1; off < LPT; off <<= 1) {
    const float v = __shfl_up(incl, off * K, 64);
    incl += (slot >= off) ? v : 0.0f;
  }
  const float total = __shfl(incl, (LPT - 1) * K + tl, 64);
  float excl = __shfl_up(incl, K, 64);
  excl = (slot == 0) ? 0.0f : excl;
  int act = amax;
  if (!(mode & 1)) {
    const float target = (float)(u32 >> 8) * (1.0f / 16777216.0f) * total;  // inverse CDF, u in [0,1)
    float cum = excl;
    int first = 64, last = -1;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      cum += e[i];
      if (ok[i]) {
        last = slot * NI + i;
        first = (first == 64 && cum > target) ? slot * NI + i : first;
      }
    }
#pragma unroll
    for (int off = K; off < 64; off <<= 1) {
      first = min(first, __shfl_xor(first, off, 64));
      last = max(last, __shfl_xor(last, off, 64));
    }
    act = (first < 64) ? first : max(last, 0);
  }
  // the chosen action's logit lives on slot act / NI
  const int ai = act % NI;
  float sel = lg[0];
#pragma unroll
  for (int i = 1; i < NI; i++) sel = (ai == i) ? lg[i] : sel;
  const float la = __shfl(sel, (act / NI) * K + tl, 64);
  log_prob = (la - mx) - logf(total);
  return act;
}
