// mv_invres.h -- the fused InvertedResidual block (invres.hip)
#pragma once
#include "mv_common.h"

namespace mv {

// 1 if a fused kernel covers the shape (then *slices / *slice_len state the projection's summation order: `slices` chains over
// `slice_len` hidden channels each, every one ascending from +0, added in ascending order), else 0
int invres_plan(int64_t n, int cin, int hidden, int cout, int h, int w, int stride, int* slices, int* slice_len);
int64_t invres_workspace_bytes(int64_t n, int cin, int hidden, int cout, int h, int w, int stride);
int launch_invres(const float* x, const float* w1, const float* a1, const float* b1, const float* wd, const float* a2, const float* b2,
                  const float* w2, const float* a3, const float* b3, int residual, float* y, int64_t n, int cin, int hidden, int cout,
                  int h, int w, int stride, int affine, void* workspace, int64_t workspace_bytes, hipStream_t s);

}  // namespace mv
