// adam_role.hpp — clip_by_global_norm + Adam on a RANGE of the flat fp32 buffers (ppo.py:195-211: optax.chain(clip_by_global_norm,
// adam); torch.optim.Adam's arithmetic) as a device function: the whole buffer (single rank, k_adam_apply) or one slice of one
// bucket (k_shard_apply).  The sweep is HBM-bound (7 floats moved per parameter).  Every block re-adds the norm partials in the
// same fixed order: deterministic, no cross-block hand-off.
#pragma once
#include <stdint.h>

struct AdamRange {
  float *p;
  const float *g;
  float *m, *v;
  int64_t lo4, hi4;            // the float4s [lo4, hi4) of the flat buffers
  const float *partials;       // the step's square-sum partials (k_adam_norm / k_adam_norm_fin)
  int npartials;
  const float *step;           // Adam's step count t (already advanced by the norm launch)
  const float *lr_dev;         // device-resident learning rate, or NULL: lr
  float lr, b1, b2, eps, max_norm, gscale;
  float *norm_out;             // NULL, or: block 0 stores the gradient norm
};

__device__ __forceinline__ float adam_wave_sum(float v) {   // (sum over the 64 lanes, every lane: fixed order)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block `b` of `nb` blocks of 256 threads sweeps its share of the range.  red: 8 floats of LDS.
__device__ __forceinline__ void adam_range_block(const AdamRange &A, const int b, const int nb, float *red) {
  const int tid = (int)threadIdx.x;
  {
    float s = 0.0f;
    for (int i0 = tid; i0 < A.npartials; i0 += 8 * 256) {   // (eight loads in flight: one round trip for <= 2048 partials)
      float pv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) pv[u] = (i0 + u * 256 < A.npartials) ? A.partials[i0 + u * 256] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; u++) s += pv[u];
    }
    s = adam_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
      // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
      red[4] = (A.max_norm > 0.0f) ? fminf(A.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (b == 0 && A.norm_out) *A.norm_out = norm;
    }
    __syncthreads();
  }
  const float scale = red[4] * A.gscale;
  const float t = *A.step;
  const float lr = (A.lr_dev != nullptr) ? *A.lr_dev : A.lr;  // device-resident: a captured launch follows the lr schedule
  const float b1 = A.b1, b2 = A.b2, eps = A.eps;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  const int64_t n4 = A.hi4 - A.lo4;
  const int64_t chunk = (n4 + nb - 1) / nb;
  const int64_t lo = A.lo4 + (int64_t)b * chunk, hi = (lo + chunk < A.hi4) ? lo + chunk : A.hi4;
  auto upd = [&](float4 &p4, const float4 g4, float4 &m4, float4 &v4) __attribute__((always_inline)) {
    const float gs[4] = {g4.x * scale, g4.y * scale, g4.z * scale, g4.w * scale};
    float ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w}, ps[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      ms[k] = ms[k] + (gs[k] - ms[k]) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
      vs[k] = vs[k] * b2 + gs[k] * gs[k] * (1.0f - b2);          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      ps[k] -= step_size * (ms[k] / (sqrtf(vs[k]) / bc2_sqrt + eps));
    }
    m4 = make_float4(ms[0], ms[1], ms[2], ms[3]);
    v4 = make_float4(vs[0], vs[1], vs[2], vs[3]);
    p4 = make_float4(ps[0], ps[1], ps[2], ps[3]);
  };
  // two float4s of every array per trip: EIGHT 16-byte loads in flight per thread before anything waits (the sweep is bound by
  // how many bytes a CU keeps in flight, not by arithmetic)
  int64_t i = lo + tid;
  for (; i + 256 < hi; i += 512) {
    const int64_t j = i + 256;
    const float4 ga = reinterpret_cast<const float4 *>(A.g)[i], gb = reinterpret_cast<const float4 *>(A.g)[j];
    float4 ma = reinterpret_cast<float4 *>(A.m)[i], mb = reinterpret_cast<float4 *>(A.m)[j];
    float4 va = reinterpret_cast<float4 *>(A.v)[i], vb = reinterpret_cast<float4 *>(A.v)[j];
    float4 pa = reinterpret_cast<float4 *>(A.p)[i], pb = reinterpret_cast<float4 *>(A.p)[j];
    upd(pa, ga, ma, va);
    upd(pb, gb, mb, vb);
    reinterpret_cast<float4 *>(A.m)[i] = ma; reinterpret_cast<float4 *>(A.v)[i] = va; reinterpret_cast<float4 *>(A.p)[i] = pa;
    reinterpret_cast<float4 *>(A.m)[j] = mb; reinterpret_cast<float4 *>(A.v)[j] = vb; reinterpret_cast<float4 *>(A.p)[j] = pb;
  }
  for (; i < hi; i += 256) {
    const float4 g4 = reinterpret_cast<const float4 *>(A.g)[i];
    float4 m4 = reinterpret_cast<float4 *>(A.m)[i], v4 = reinterpret_cast<float4 *>(A.v)[i], p4 = reinterpret_cast<float4 *>(A.p)[i];
    upd(p4, g4, m4, v4);
    reinterpret_cast<float4 *>(A.m)[i] = m4; reinterpret_cast<float4 *>(A.v)[i] = v4; reinterpret_cast<float4 *>(A.p)[i] = p4;
  }
}
