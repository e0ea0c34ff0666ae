// mv_epilogue.h -- the per-element epilogue shared by the conv kernels of convnorm.hip and deform_fused.hip:
// [+ bias] -> folded norm -> [+ residual] -> activation, in the oracle's operation order (oracle/oracle.c::orc_conv2d_affine_act_f32).
#pragma once
#include "mv_act.h"
#include "mv_common.h"

namespace mv {

// The epilogue is branch-free per element: the per-channel terms are loaded once per channel (`ChannelTerms`), the
// norm variants are computed side by side and selected, ReLU / ReLU6 are one compare-select pair against (lo, hi)
// (NaN passes through, as `v < 0 ? 0 : v` does); only Hardswish / SiLU take a (wave-uniform) branch.  A run-time
// switch per element cost more than the convolution itself.
struct ChannelTerms {
  float bias, alpha, beta;
};

__device__ inline ChannelTerms channel_terms(const Epilogue& e, int m) {
  ChannelTerms c = {0.f, 1.f, 0.f};
  if (e.bias) c.bias = e.bias[m];
  if (e.affine) c.alpha = e.alpha[m], c.beta = e.beta[m];
  return c;
}

__device__ inline float epi_norm(float acc, const ChannelTerms& c, const Epilogue& e) {
  const float v = e.bias ? acc + c.bias : acc;
  float two = v * c.alpha;      // FrozenBatchNorm2d: x * scale, then + bias
  two = two + c.beta;
  const float one = fmaf(v, c.alpha, c.beta);  // BatchNorm2d (eval)
  return e.affine == 2 ? one : (e.affine == 1 ? two : v);
}

// activation kinds: 0 = none / ReLU / ReLU6 as one compare-select pair against (lo, hi), 1 = Hardswish, 2 = SiLU
struct Clamp {
  float lo, hi;
};
__device__ inline Clamp make_clamp(int act) {
  Clamp c;
  c.lo = (act == 1 || act == 2) ? 0.f : -__builtin_inff();
  c.hi = (act == 2) ? 6.f : __builtin_inff();
  return c;
}
template <int ACTK>
__device__ inline float epi_act(float v, const Clamp& c) {
  if (ACTK == 1) {
    float t = v + 3.f;
    t = clamp_f32(t, 0.f, 6.f);
    return v * t / 6.f;
  }
  if (ACTK == 2) return v / (1.f + expf(-v));
  return clamp_f32(v, c.lo, c.hi);
}

// single-output form (depthwise / stem kernels): one channel's terms, one element
__device__ inline float epi_apply(float acc, const ChannelTerms& c, size_t out_index, const Epilogue& e) {
  float v = epi_norm(acc, c, e);
  if (e.res) v = e.res[out_index] + v;
  if (e.act == 3) return epi_act<1>(v, Clamp{});
  if (e.act == 4) return epi_act<2>(v, Clamp{});
  return epi_act<0>(v, make_clamp(e.act));
}

}  // namespace mv
