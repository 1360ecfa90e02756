// imp.hpp — the IMP conversion of a duplicate board (src/duplicate.py:15-70), shared by brl_imp_reward and the evaluators' step.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float4 imp_vector(float a0, float b0) {
  const float th[24] = {20, 50, 90, 130, 170, 220, 270, 320, 370, 430, 500, 600,
                        750, 900, 1100, 1300, 1500, 1750, 2000, 2250, 2500, 3000, 3500, 4000};
  float d = a0 + b0;
  float win = (d >= 0.0f) ? 1.0f : -1.0f;  // src/duplicate.py:52-54
  float ad = fabsf(d);
  int imp = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) imp += (ad >= th[i]) ? 1 : 0;  // src/duplicate.py:46-69
  float v = (float)imp * win;
  return make_float4(v, v, -v, -v);
}
