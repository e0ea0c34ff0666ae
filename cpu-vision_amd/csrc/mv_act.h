// mv_act.h -- ReLU / clamp on gfx950's v_maximum3_f32 / v_minimum3_f32 (IEEE-754-2019 maximum / minimum): NaN propagates, like
// torch.relu / hardtanh (ATen clamp), in ONE instruction each.  The compare-select form `v < 0 ? 0 : v` costs v_cmp + 2 wait
// states (VALU write of VCC -> v_cndmask) + v_cndmask: 10 issue cycles per element against 4 -- the expansion phase of the fused
// InvertedResidual kernels spent as long clamping as multiplying (profiles/r03_trace_invres_wide_v1.log).
// maximum(-0, +0) = +0 where the select kept -0: equal by value, and nothing downstream looks at a zero's sign.
#pragma once

namespace mv {

__device__ __forceinline__ float relu_f32(float v) { return __builtin_elementwise_maximum(v, 0.f); }
__device__ __forceinline__ float clamp_f32(float v, float lo, float hi) {
  return __builtin_elementwise_minimum(__builtin_elementwise_maximum(v, lo), hi);
}

}  // namespace mv
