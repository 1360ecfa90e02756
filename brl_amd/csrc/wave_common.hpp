// wave_common.hpp — what the per-step kernels of every translation unit share (K tables per 64-lane wave, 4 waves per 256-thread
// workgroup): the wave's LDS table images, row emission, the re-deal of finished tables, the optional per-step outputs, and the
// device-resident mirror of (LUT, RNG key, env_offset).  Device code only; the table logic itself is bridge_device.hpp.
#pragma once
#include "../../include/brl_hip.h"
#include "bridge_device.hpp"

using namespace brl;

// =====================================================================================
// wave-level helpers (K tables per 64-lane wave, 4 waves per 256-thread workgroup)
// =====================================================================================
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;

// Consecutive table groups go to the same XCD (blocks b and b+8 share one): neighbouring
// rows of the [n,480] / [n,38] outputs share 128-B lines, keep those in ONE L2.  Speed only.
__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nb) {
  return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
}

template <int K>
struct Wave {
  LaneConst c;
  uint8_t *wimg;   // this wave's K x 128 B LDS images
  int tl;          // local table of this lane's logic (lane % K)
  int64_t table0;  // first table of the wave
  int64_t table;   // table of this lane's logic
  bool valid;      // table < n
};

template <int K>
__device__ __forceinline__ Wave<K> wave_begin(uint8_t *lds, const uint64_t *state_in, int64_t n, Tbl &t) {
  Wave<K> w;
  w.c = make_lane_const();
  int wave = (int)(threadIdx.x >> 6);
  int64_t blk = xcd_block((int64_t)blockIdx.x, (int64_t)gridDim.x);
  w.table0 = (blk * WAVES_PER_BLOCK + wave) * K;
  w.wimg = lds + wave * K * TABLE_BYTES;
  uint64_t *wimg64 = reinterpret_cast<uint64_t *>(w.wimg);
#pragma unroll
  for (int i = w.c.lane; i < K * 16; i += 64) {
    int64_t tb = w.table0 + i / 16;
    wimg64[i] = (state_in != nullptr && tb < n) ? state_in[w.table0 * 16 + i] : 0ull;
  }
  wave_lds_fence();
  w.tl = w.c.lane % K;
  w.table = w.table0 + w.tl;
  w.valid = w.table < n;
  load_scalars(t, w.wimg + w.tl * TABLE_BYTES);
  return w;
}

template <int K>
__device__ __forceinline__ void wave_end(const Wave<K> &w, const Tbl &t, uint64_t *state_out, int64_t n) {
  if (w.c.lane < K) store_scalars(t, w.wimg + w.tl * TABLE_BYTES);
  wave_lds_fence();
  const uint64_t *wimg64 = reinterpret_cast<const uint64_t *>(w.wimg);
#pragma unroll
  for (int i = w.c.lane; i < K * 16; i += 64) {
    int64_t tb = w.table0 + i / 16;
    if (tb < n) state_out[w.table0 * 16 + i] = wimg64[i];
  }
}

template <int K>
__device__ __forceinline__ void wave_or_hist(const Wave<K> &w, int hist_bit) {
  if (w.c.lane < K && hist_bit >= 0) {
    uint32_t *p = reinterpret_cast<uint32_t *>(w.wimg + w.tl * TABLE_BYTES) + (hist_bit >> 5);
    atomicOr(p, 1u << (hist_bit & 31));  // ds_or_b32
  }
}

typedef uint32_t brl_u32x4 __attribute__((ext_vector_type(4)));

// Write-through 16-byte store (sc0 sc1: the bytes go to memory now and the line is not kept in L2).  For everything a
// fused rollout launch writes besides the observations: a plain store leaves a dirty line in the XCD's L2 and all of them — 15 MB of
// mask rows and scalar columns — are written back when the kernel ENDS, after the last wave: 2.6 us of 27.4.
__device__ __forceinline__ void store_wt16(void *p, brl_u32x4 v) {
  // (s_nop: the compiler does not know that the instruction still reads its data registers for two more cycles)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

struct LutRef {
  const int4 *keys;
  const int4 *values;
  uint32_t len;
  const uint4 *hands;  // per row: the four packed hand words (hand_obs[seat] << 4, image words 7..10), derived from keys
};

// A5 post-step half of auto_reset (src/utils.py:45-55) for every table of the wave that
// just terminated: deal board bctr+1 of that slot, keep (terminated, rewards).
template <int K>
__device__ __forceinline__ void wave_reset(const Wave<K> &w, Tbl &t, bool need, const Rng &g, uint64_t env_offset,
                                           const LutRef &lut, uint32_t next_ctr) {
  uint64_t needm = __ballot(need) & ((1ull << K) - 1ull);
  if (needm == 0ull) return;
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (need) {
    uint32_t keep = t.sc & ((1u << SC_TERM) | (1u << SC_ILLEGAL));
    fresh_scalars(t, g, env_offset + (uint64_t)w.table, next_ctr, lut.len, keep);
    int4 kv = lut.keys[t.lut];
    int4 vv = lut.values[t.lut];
    q0 = (uint32_t)kv.x; q1 = (uint32_t)kv.y; q2 = (uint32_t)kv.z; q3 = (uint32_t)kv.w;
    pack_tricks(t, (uint32_t)vv.x, (uint32_t)vv.y, (uint32_t)vv.z, (uint32_t)vv.w);
  }
  while (needm) {
    int j = __ffsll((unsigned long long)needm) - 1;
    needm &= needm - 1ull;
    deal_image(w.wimg + j * TABLE_BYTES, __builtin_amdgcn_readlane(q0, j), __builtin_amdgcn_readlane(q1, j),
               __builtin_amdgcn_readlane(q2, j), __builtin_amdgcn_readlane(q3, j), w.c);
  }
  wave_lds_fence();
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int j) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, j);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), j);
  return ((uint64_t)hi << 32) | lo;
}

// emit obs/mask rows of the wave's K tables; row of table0 is row0 (rows are consecutive)
template <int K>
__device__ __forceinline__ void wave_emit(const Wave<K> &w, int64_t n, int oseat, uint32_t vulnib, uint64_t legal,
                                          uint8_t *obs, uint8_t *mask, int64_t row0) {
  uint32_t pack = (uint32_t)oseat | (vulnib << 2);
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (w.table0 + j < n) {
      if (obs) {
        uint32_t p = __builtin_amdgcn_readlane(pack, j);
        emit_obs_row(w.wimg + j * TABLE_BYTES, (int)(p & 3u), p >> 2, obs + (row0 + j) * BRL_OBS_SIZE, w.c);
      }
      if (mask) emit_mask_row(readlane64(legal, j), mask + (row0 + j) * BRL_NUM_ACTIONS, w.c);
    }
  }
}

__device__ __forceinline__ float4 rewards_f32(const Tbl &t) {
  return make_float4((float)reward_of(t, 0), (float)reward_of(t, 1), (float)reward_of(t, 2), (float)reward_of(t, 3));
}

__device__ __forceinline__ int sanitize_action(int a, uint32_t &bad) {
  bad = ((uint32_t)a >= (uint32_t)BRL_NUM_ACTIONS) ? 1u : 0u;
  return bad ? 0 : a;
}

struct StepOut {
  uint8_t *obs;
  uint8_t *mask;
  float *rewards;
  uint8_t *terminated;
  int32_t *current_player;
};

template <int K>
__device__ __forceinline__ void wave_step_outputs(const Wave<K> &w, const Tbl &t, int64_t n, const StepOut &o) {
  int oseat = cur_seat(t);
  wave_emit<K>(w, n, oseat, vul_nibble(t, oseat), legal_mask(t), o.obs, o.mask, w.table0);
  if (w.c.lane < K && w.valid) {
    if (o.rewards) reinterpret_cast<float4 *>(o.rewards)[w.table] = rewards_f32(t);
    if (o.terminated) o.terminated[w.table] = (uint8_t)bits(t.sc, SC_TERM, 1);
    if (o.current_player) o.current_player[w.table] = cur_player(t);
  }
}

// What brl_set_rng / brl_set_lut change, mirrored in device memory: the policy sub-step reads it from there instead
// of taking it by value, so that a hipGraph replay of a captured launch follows a later re-seed or LUT rotation
// (ppo.py:525-549) instead of reading freed tables / a stale key.
struct DevCtx {
  LutRef lut;
  Rng g;
  uint64_t env_offset;
};
