// Gate arithmetic shared by the per-position GRU kernels (gru_kernels.hip) and the persistent window-chain kernels
// (gru_chain.hip).
#pragma once
#include "common.hpp"

namespace temp {

// Gate non-linearities on the hardware exp / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  libm's
// expf / tanhf cost ~150 VALU instructions per element, i.e. ~10 us per position of the window chain.
// tanh switches to its odd Taylor polynomial below |x| = 0.25, where 1 - 2/(1+e^2x) would cancel.
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gate_tanh(float x) {
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
  const float x2 = x * x;
  const float small = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
  return fabsf(x) < 0.25f ? small : big;
}

__device__ __forceinline__ float decay_factor(float dt, float lambda, const float* wb) {
  if (wb) return expf(-fmaxf(fmaf(wb[0], dt, wb[1]), 0.f));
  return expf(-dt * lambda);
}

}  // namespace temp
