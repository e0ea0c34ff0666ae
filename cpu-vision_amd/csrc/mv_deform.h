// mv_deform.h -- deform_conv2d's two forms: deform_fused.hip (no columns in HBM) and deform.hip (columns workspace + GEMM)
#pragma once
#include "mv_common.h"

namespace mv {

// true: launch_deform_fused takes this geometry (its tiles fit in LDS), and the caller needs no workspace
bool deform_fused_supported(int cin, int cout, int h, int wd, int kh, int kw, int sh, int sw, int dh, int dw, int groups,
                            int offset_groups);
// workgroups the fused launch would have (0: unsupported geometry)
int64_t deform_fused_workgroups(int64_t n, int cin, int cout, int h, int wd, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                int dw, int groups, int offset_groups);
constexpr int64_t kDeformFusedMinWorkgroups = 256;  // fewer: the columns workspace + GEMM form is faster when a workspace is there
int launch_deform_fused(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                        int64_t n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                        int groups, int offset_groups, int use_mask, hipStream_t s, int act);

}  // namespace mv
