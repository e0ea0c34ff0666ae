// mv_conv.h -- the general-cin 3x3 convolution with K slices across workgroups (conv3x3_gen.hip)
#pragma once
#include "mv_common.h"

namespace mv {

// slices == 1: one ascending (channel, ky, kx) chain per output; otherwise `slices` chains over `slice_channels` input channels
// each (the last may be shorter), added in ascending slice order, then bias, then ReLU
void conv3x3_gen_plan(int64_t n, int cin, int h, int wdt, int cout, int* slices, int* slice_channels);
int64_t conv3x3_gen_workspace_bytes(int64_t n, int cin, int h, int wdt, int cout);
// workspace == nullptr: the single chain (== launch_conv3x3_gen)
int launch_conv3x3_gen_ws(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt, int cout,
                          int relu, hipStream_t s, void* workspace, int64_t workspace_bytes);

// conv_igemm.hip: workgroups the implicit-GEMM conv would launch with its smallest tiles; below ~128 the columns form (smaller
// tiles) is faster and mv_conv2d_needs_workspace() answers 2 = optional
long long conv2d_implicit_min_workgroups(int64_t n, int mg, int oh, int ow);

}  // namespace mv
