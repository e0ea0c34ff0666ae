// deform.hip -- torchvision::deform_conv2d forward (csrc/ops/deform_conv2d.cpp:164-169; CPU kernel
// csrc/ops/cpu/deform_conv2d_kernel.cpp) on gfx950: SURVEY.md section 8f.4.
//
//   1. k_deform_im2col: columns[b][(c*kh + i)*kw + j][oy*ow + ox] = mask * bilinear(input[b, c], y, x) with
//      y = (oy*stride_h - pad_h) + i*dil_h + offset_h, x likewise (deform_conv2d_kernel.cpp:118-193, bilinear :80-116:
//      same float operations in the same order, no contraction) -- a gather: 4 corner loads per tap, 2-3 offset / mask
//      loads (coalesced along ox), one coalesced store.  Thread = one (b, c, oy, ox), loop over the kh*kw taps.
//   2. the GEMM out[b, m] = W[m, :] . columns[b] per weight group, + bias: k_conv1x1 of convnorm.hip (fp32 MFMA, one
//      accumulator per output in ascending k, bias in the epilogue), reading the group's channel slice of `columns`.
// `columns` lives in caller-provided workspace ([images][cin*kh*kw][oh*ow] floats); the batch is processed in passes of
// as many images as the workspace holds (the reference does the same with up to 32 "parallel images").
#include "mv_common.h"

namespace mv {

struct DeformArgs {
  const float* x;
  const float* offset;
  const float* mask;
  float* col;
  long long total;  // images * cin * oh * ow
  int cin, h, wd, kh, kw, sh, sw, ph, pw, dh, dw;
  int oh, ow, offset_groups, use_mask;
};

__device__ inline float deform_bilinear(const float* in, int height, int width, float h, float w) {
  if (h <= -1 || height <= h || w <= -1 || width <= w) return 0.f;
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = in[h_low * width + w_low];
  if (h_low >= 0 && w_high <= width - 1) v2 = in[h_low * width + w_high];
  if (h_high <= height - 1 && w_low >= 0) v3 = in[h_high * width + w_low];
  if (h_high <= height - 1 && w_high <= width - 1) v4 = in[h_high * width + w_high];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  float val = w1 * v1;
  val = val + w2 * v2;
  val = val + w3 * v3;
  val = val + w4 * v4;
  return val;
}

__global__ __launch_bounds__(256) void k_deform_im2col(const DeformArgs A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.total) return;
  const int ox = (int)(idx % A.ow);
  long long t = idx / A.ow;
  const int oy = (int)(t % A.oh);
  t /= A.oh;
  const int c = (int)(t % A.cin);
  const long long b = t / A.cin;
  const size_t ohw = (size_t)A.oh * A.ow;
  const size_t pix = (size_t)oy * A.ow + ox;
  const int taps = A.kh * A.kw;
  const int og = c / (A.cin / A.offset_groups);
  const float* xp = A.x + ((size_t)b * A.cin + c) * A.h * A.wd;
  const float* op = A.offset ? A.offset + ((size_t)b * A.offset_groups + og) * 2 * taps * ohw + pix : nullptr;
  const float* mp = A.use_mask ? A.mask + ((size_t)b * A.offset_groups + og) * taps * ohw + pix : nullptr;
  float* cp = A.col + ((size_t)b * A.cin + c) * taps * ohw + pix;
  const int y0 = oy * A.sh - A.ph, x0 = ox * A.sw - A.pw;
  for (int i = 0; i < A.kh; ++i)
    for (int j = 0; j < A.kw; ++j) {
      const int mi = i * A.kw + j;
      if (A.offset == nullptr) {  // plain im2col (mv_conv2d_bias_act_f32): zero padding outside the image
        const int iy = y0 + i * A.dh, ix = x0 + j * A.dw;
        cp[(size_t)mi * ohw] = (iy >= 0 && iy < A.h && ix >= 0 && ix < A.wd) ? xp[(size_t)iy * A.wd + ix] : 0.f;
        continue;
      }
      const float mv = A.use_mask ? mp[(size_t)mi * ohw] : 1.f;
      const float off_h = op[(size_t)(2 * mi) * ohw];
      const float off_w = op[(size_t)(2 * mi + 1) * ohw];
      const float y = (y0 + i * A.dh) + off_h;
      const float x = (x0 + j * A.dw) + off_w;
      cp[(size_t)mi * ohw] = mv * deform_bilinear(xp, A.h, A.wd, y, x);
    }
}

int64_t deform_workspace_bytes_per_image(int cin, int kh, int kw, int oh, int ow) {
  return (int64_t)sizeof(float) * cin * kh * kw * oh * ow;
}

// offset == nullptr: ordinary convolution (plain im2col); act: MV_ACT_* applied after the bias
int launch_deform_conv2d(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                         int64_t n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                         int groups, int offset_groups, int use_mask, void* workspace, int64_t workspace_bytes, hipStream_t s, int act) {
  const int oh = (h + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int ow = (wd + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  const int64_t per_image = deform_workspace_bytes_per_image(cin, kh, kw, oh, ow);
  if (workspace == nullptr || workspace_bytes < per_image)
    return set_error(MV_ERR_INVALID_ARGUMENT, "deform_conv2d: workspace of at least %lld bytes (one image's columns) needed, got %lld",
                     (long long)per_image, (long long)workspace_bytes);
  int64_t pass = workspace_bytes / per_image;
  if (pass > n) pass = n;
  if (pass > 65535) pass = 65535;
  const int taps = kh * kw, cg = cin / groups, mg = cout / groups;
  const int64_t ohw = (int64_t)oh * ow;
  const Epilogue none = {nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int64_t b0 = 0; b0 < n; b0 += pass) {
    const int64_t nb = (n - b0 < pass) ? n - b0 : pass;
    DeformArgs a = {};
    a.x = x + (size_t)b0 * cin * h * wd;
    a.offset = offset ? offset + (size_t)b0 * offset_groups * 2 * taps * ohw : nullptr;
    a.mask = use_mask ? mask + (size_t)b0 * offset_groups * taps * ohw : nullptr;
    a.col = static_cast<float*>(workspace);
    a.cin = cin, a.h = h, a.wd = wd, a.kh = kh, a.kw = kw, a.sh = sh, a.sw = sw, a.ph = ph, a.pw = pw, a.dh = dh, a.dw = dw;
    a.oh = oh, a.ow = ow, a.offset_groups = offset_groups, a.use_mask = use_mask;
    a.total = nb * cin * ohw;
    if (a.total > 256LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "deform_conv2d: pass too large for one launch");
    hipLaunchKernelGGL(k_deform_im2col, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
    if (int rc = check_launch("k_deform_im2col")) return rc;
    for (int g = 0; g < groups; ++g) {
      Epilogue e = none;
      e.bias = bias ? bias + (size_t)g * mg : nullptr;
      e.act = act;
      const int rc = launch_conv1x1(a.col + (size_t)g * cg * taps * ohw, weight + (size_t)g * mg * cg * taps,
                                    y + ((size_t)b0 * cout + (size_t)g * mg) * ohw, nb, cg * taps, ohw, mg, e, s,
                                    (int64_t)cin * taps * ohw, (int64_t)cout * ohw);
      if (rc) return rc;
    }
  }
  return MV_OK;
}

}  // namespace mv
