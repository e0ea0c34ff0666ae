// deform.hip -- torchvision::deform_conv2d forward (csrc/ops/deform_conv2d.cpp:164-169; CPU kernel
// csrc/ops/cpu/deform_conv2d_kernel.cpp) on gfx950: SURVEY.md section 8f.4.
//
//   1. k_deform_im2col: columns[b][(c*kh + i)*kw + j][oy*ow + ox] = mask * bilinear(input[b, c], y, x) with
//      y = (oy*stride_h - pad_h) + i*dil_h + offset_h, x likewise (deform_conv2d_kernel.cpp:118-193, bilinear :80-116:
//      same float operations in the same order, no contraction) -- a gather: per (tap, pixel) the sampling position and
//      the four corner weights are computed once, then 4 corner loads + one coalesced store per channel.
//   2. the GEMM out[b, m] = W[m, :] . columns[b] per weight group, + bias: k_conv1x1 of convnorm.hip (fp32 MFMA, one
//      accumulator per output in ascending k, bias in the epilogue), reading the group's channel slice of `columns`.
// `columns` lives in caller-provided workspace ([images][cin*kh*kw][oh*ow] floats); the batch is processed in passes of
// as many images as the workspace holds (the reference does the same with up to 32 "parallel images").
// This two-kernel form serves the plain im2col convolutions (mv_conv2d_bias_act_f32) and the deformable geometries whose
// tiles do not fit the fused kernel of deform_fused.hip (more than 40 taps, huge stride x dilation windows).
#include "mv_common.h"
#include "mv_conv.h"
#include "mv_deform.h"

namespace mv {

struct DeformArgs {
  const float* x;
  const float* offset;
  const float* mask;
  float* col;
  long long total;  // images * offset groups * channel chunks * taps * oh * ow
  int cin, h, wd, kh, kw, sh, sw, ph, pw, dh, dw;
  int oh, ow, offset_groups, use_mask;
  int cchunks, cper;  // channel chunks per offset group (more threads for small maps) and channels per chunk
};

// thread = one (image, offset group, tap, output pixel): the offsets, the mask and the bilinear corner weights depend on
// nothing else, so they are computed once and applied to every channel of the offset group (4 gathers + 1 coalesced
// store per channel).  A thread per (channel, pixel) re-read the offsets and recomputed the weights cin times.
__global__ __launch_bounds__(256) void k_deform_im2col(const DeformArgs A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.total) return;
  const int taps = A.kh * A.kw;
  const int ox = (int)(idx % A.ow);
  long long t = idx / A.ow;
  const int oy = (int)(t % A.oh);
  t /= A.oh;
  const int mi = (int)(t % taps);
  t /= taps;
  const int cc = (int)(t % A.cchunks);
  t /= A.cchunks;
  const int og = (int)(t % A.offset_groups);
  const long long b = t / A.offset_groups;
  const int i = mi / A.kw, j = mi - i * A.kw;
  const size_t ohw = (size_t)A.oh * A.ow;
  const size_t pix = (size_t)oy * A.ow + ox;
  const int cog = A.cin / A.offset_groups;
  const int H = A.h, W = A.wd;
  const int c_first = cc * A.cper, c_count = min(A.cper, cog - c_first);
  const float* xp = A.x + ((size_t)b * A.cin + (size_t)og * cog + c_first) * H * W;
  float* cp = A.col + (((size_t)b * A.cin + (size_t)og * cog + c_first) * taps + mi) * ohw + pix;
  const int y0 = oy * A.sh - A.ph + i * A.dh, x0 = ox * A.sw - A.pw + j * A.dw;
  if (A.offset == nullptr) {  // plain im2col (mv_conv2d_bias_act_f32): zero padding outside the image
    const bool ok = y0 >= 0 && y0 < H && x0 >= 0 && x0 < W;
    const size_t o = ok ? (size_t)y0 * W + x0 : 0;
    for (int c = 0; c < c_count; ++c) cp[(size_t)c * taps * ohw] = ok ? xp[(size_t)c * H * W + o] : 0.f;
    return;
  }
  const float* op = A.offset + (((size_t)b * A.offset_groups + og) * 2 * taps + 2 * mi) * ohw + pix;
  const float mv = A.use_mask ? A.mask[(((size_t)b * A.offset_groups + og) * taps + mi) * ohw + pix] : 1.f;
  const float h = y0 + op[0];
  const float w = x0 + op[ohw];
  // bilinear_interpolate (deform_conv2d_kernel.cpp:80-116) with the channel-independent part hoisted
  const bool outside = (h <= -1 || H <= h || w <= -1 || W <= w);
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  const bool ok1 = !outside && h_low >= 0 && w_low >= 0;
  const bool ok2 = !outside && h_low >= 0 && w_high <= W - 1;
  const bool ok3 = !outside && h_high <= H - 1 && w_low >= 0;
  const bool ok4 = !outside && h_high <= H - 1 && w_high <= W - 1;
  const int o1 = ok1 ? h_low * W + w_low : 0, o2 = ok2 ? h_low * W + w_high : 0;
  const int o3 = ok3 ? h_high * W + w_low : 0, o4 = ok4 ? h_high * W + w_high : 0;
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  // Both columns of the 2 x 2 neighbourhood inside the image (all but the samples on the left / right edge): the two corners
  // of a row are adjacent floats -> ONE 8-byte load per row (any 4-byte-aligned address works on gfx950) instead of two
  // 4-byte gathers; the gather instructions, not the arithmetic, bound this kernel.  Same values, same operations.
  if (!outside && w_low >= 0 && w_high <= W - 1) {
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    const bool top = h_low >= 0, bot = h_high <= H - 1;
    const int ot = top ? h_low * W + w_low : 0, ob = bot ? h_high * W + w_low : 0;
    for (int c = 0; c < c_count; ++c) {
      const float* in = xp + (size_t)c * H * W;
      const f32x2u zero2 = {0.f, 0.f};
      const f32x2u vt = top ? *reinterpret_cast<const f32x2u*>(in + ot) : zero2;
      const f32x2u vb = bot ? *reinterpret_cast<const f32x2u*>(in + ob) : zero2;
      float val = w1 * vt.x;
      val = val + w2 * vt.y;
      val = val + w3 * vb.x;
      val = val + w4 * vb.y;
      cp[(size_t)c * taps * ohw] = mv * val;
    }
    return;
  }
  for (int c = 0; c < c_count; ++c) {
    const float* in = xp + (size_t)c * H * W;
    const float v1 = ok1 ? in[o1] : 0.f, v2 = ok2 ? in[o2] : 0.f, v3 = ok3 ? in[o3] : 0.f, v4 = ok4 ? in[o4] : 0.f;
    float val = w1 * v1;
    val = val + w2 * v2;
    val = val + w3 * v3;
    val = val + w4 * v4;
    cp[(size_t)c * taps * ohw] = mv * (outside ? 0.f : val);
  }
}

// ---- the same columns, with the gathers served from LDS -------------------------------------------------------------------
// k_deform_im2col is bound by its gather INSTRUCTIONS: neighbouring lanes sample scattered addresses, so each of the 4
// (2 as pairs) corner loads per element costs the vector-memory address path ~64 cycles per wave, whatever the bytes
// (8 x 256 x 64 x 64, 3x3: 327 us for 75 M elements).  Here a workgroup owns a 32 x 8 tile of output pixels of one (image,
// offset group); per chunk of kDefCB channels it stages, with coalesced loads, the input window those pixels can sample
// from while |offset| <= kDefMargin (rows / columns outside the image are stored as zeros, which is exactly what the
// reference's bilinear_interpolate substitutes for corners outside the image), and every corner then comes from LDS
// (two ds_read2_b32 per element).  A sample whose 2 x 2 neighbourhood leaves the window (a larger offset) takes the
// global-memory path of k_deform_im2col for that (tap, pixel) -- same operations, same results, just slower.
// The arithmetic per element is unchanged: val = w1*v1; val += w2*v2; val += w3*v3; val += w4*v4; out = mask * val.
constexpr int kDefTW = 32, kDefTH = 8, kDefMargin = 6, kDefCB = 8;

struct DeformLdsArgs {
  const float* x;
  const float* offset;
  const float* mask;
  float* col;
  int cin, h, wd, kh, kw, sh, sw, ph, pw, dh, dw;
  int oh, ow, offset_groups, use_mask;
  int tiles_x, tiles_y;
  int csplit, cper;             // channel ranges per (image, offset group, tile) and channels per range (multiple of kDefCB)
  int win_h, win_w, win_pitch;  // staged window (floats), pitch odd
};

__global__ __launch_bounds__(256) void k_deform_im2col_lds(const DeformLdsArgs A) {
  extern __shared__ float win[];  // [kDefCB][win_h][win_pitch]
  const int tid = threadIdx.x;
  const int tx = tid % kDefTW, ty = tid / kDefTW;
  unsigned bid = blockIdx.x;
  const int cs = bid % A.csplit;
  bid /= A.csplit;
  const int tile_x = bid % A.tiles_x;
  bid /= A.tiles_x;
  const int tile_y = bid % A.tiles_y;
  bid /= A.tiles_y;
  const int og = bid % A.offset_groups;
  const long long b = bid / A.offset_groups;
  const int ox = tile_x * kDefTW + tx, oy = tile_y * kDefTH + ty;
  const bool live = ox < A.ow && oy < A.oh;
  const int taps = A.kh * A.kw;
  const int H = A.h, W = A.wd;
  const size_t ohw = (size_t)A.oh * A.ow;
  const size_t pix = (size_t)min(oy, A.oh - 1) * A.ow + min(ox, A.ow - 1);
  const int cog = A.cin / A.offset_groups;
  // window origin in image coordinates (may be negative: those rows / columns are zeros)
  const int wy0 = tile_y * kDefTH * A.sh - A.ph - kDefMargin, wx0 = tile_x * kDefTW * A.sw - A.pw - kDefMargin;
  const int wh = A.win_h, ww = A.win_w, wp = A.win_pitch;
  const int wsz = wh * wp;
  const float* xg = A.x + ((size_t)b * A.cin + (size_t)og * cog) * H * W;
  float* cg = A.col + ((size_t)b * A.cin + (size_t)og * cog) * taps * ohw + pix;
  const float* offp = A.offset + ((size_t)b * A.offset_groups + og) * 2 * taps * ohw + pix;
  const float* mskp = A.use_mask ? A.mask + ((size_t)b * A.offset_groups + og) * taps * ohw + pix : nullptr;

  const int c_end = min(cog, (cs + 1) * A.cper);
  for (int c0 = cs * A.cper; c0 < c_end; c0 += kDefCB) {
    const int cb = min(kDefCB, c_end - c0);
    __syncthreads();  // the previous chunk's window is no longer read
    for (int i = tid; i < cb * wh * ww; i += 256) {
      const int c = i / (wh * ww), r = i - c * (wh * ww);
      const int ly = r / ww, lx = r - ly * ww;
      const int gy = wy0 + ly, gx = wx0 + lx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xg[((size_t)(c0 + c) * H + gy) * W + gx];
      win[c * wsz + ly * wp + lx] = v;
    }
    __syncthreads();
    if (!live) continue;
    for (int mi = 0; mi < taps; ++mi) {
      const int i = mi / A.kw, j = mi - i * A.kw;
      const float mv = A.use_mask ? mskp[(size_t)mi * ohw] : 1.f;
      const float h = (oy * A.sh - A.ph + i * A.dh) + offp[(size_t)(2 * mi) * ohw];
      const float w = (ox * A.sw - A.pw + j * A.dw) + offp[(size_t)(2 * mi + 1) * ohw];
      const int h_low = (int)floorf(h), w_low = (int)floorf(w);
      const float lh = h - h_low, lw = w - w_low;
      const float hh = 1 - lh, hw = 1 - lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      float* cp = cg + ((size_t)c0 * taps + mi) * ohw;
      const int ly = h_low - wy0, lx = w_low - wx0;
      // The window serves samples INSIDE the image only: for h <= -1, H <= h, w <= -1 or W <= w the reference returns exactly 0
      // (deform_conv2d_kernel.cpp:88-90) where the window would compute 0 x (the border pixel) -- NaN if that pixel is not
      // finite.  Those samples, huge offsets and NaN positions (every comparison false) take the global path below, which
      // applies the reference's `outside` test.
      const bool inside = h > -1 && h < H && w > -1 && w < W;
      if (inside && ly >= 0 && ly + 1 < wh && lx >= 0 && lx + 1 < ww) {
        const float* wpnt = win + ly * wp + lx;
        for (int c = 0; c < cb; ++c) {
          const float* q = wpnt + c * wsz;
          const float v1 = q[0], v2 = q[1], v3 = q[wp], v4 = q[wp + 1];
          float val = w1 * v1;
          val = val + w2 * v2;
          val = val + w3 * v3;
          val = val + w4 * v4;
          cp[(size_t)c * taps * ohw] = mv * val;
        }
      } else {
        const bool outside = (h <= -1 || H <= h || w <= -1 || W <= w);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const bool ok1 = !outside && h_low >= 0 && w_low >= 0;
        const bool ok2 = !outside && h_low >= 0 && w_high <= W - 1;
        const bool ok3 = !outside && h_high <= H - 1 && w_low >= 0;
        const bool ok4 = !outside && h_high <= H - 1 && w_high <= W - 1;
        const int o1 = ok1 ? h_low * W + w_low : 0, o2 = ok2 ? h_low * W + w_high : 0;
        const int o3 = ok3 ? h_high * W + w_low : 0, o4 = ok4 ? h_high * W + w_high : 0;
        for (int c = 0; c < cb; ++c) {
          const float* in = xg + (size_t)(c0 + c) * H * W;
          const float v1 = ok1 ? in[o1] : 0.f, v2 = ok2 ? in[o2] : 0.f, v3 = ok3 ? in[o3] : 0.f, v4 = ok4 ? in[o4] : 0.f;
          float val = w1 * v1;
          val = val + w2 * v2;
          val = val + w3 * v3;
          val = val + w4 * v4;
          cp[(size_t)c * taps * ohw] = mv * (outside ? 0.f : val);
        }
      }
    }
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
  // deformable sampling: the fused kernel (deform_fused.hip) whenever its tiles fit -- every usual DCN layer; no workspace.
  // Launches too small to fill the chip with it use the two kernels below if the caller brought a workspace
  if (offset != nullptr && !tune_env("MV_DEFORM_UNFUSED")) {
    const int64_t wgs = deform_fused_workgroups(n, cin, cout, h, wd, kh, kw, sh, sw, ph, pw, dh, dw, groups, offset_groups);
    const bool have_ws = workspace != nullptr && workspace_bytes >= deform_workspace_bytes_per_image(cin, kh, kw, oh, ow);
    if (wgs > 0 && (wgs >= kDeformFusedMinWorkgroups || !have_ws))
      return launch_deform_fused(x, weight, offset, mask, bias, y, n, cin, h, wd, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                 offset_groups, use_mask, s, act);
  }
  // ordinary convolution (mv_conv2d_bias_act_f32): implicit GEMM -- the K chunk's columns are gathered from the input while the
  // GEMM stages them, so they never exist in HBM and the workspace is not used (the columns form below remains for sizes the
  // implicit kernel's index arithmetic does not cover, and for A/B in the tuning build: MV_CONV_COLUMNS)
  const bool small_launch_with_workspace = offset == nullptr && conv2d_implicit_min_workgroups(n, cout / groups, oh, ow) * groups < 128 &&
                                           workspace != nullptr && workspace_bytes >= deform_workspace_bytes_per_image(cin, kh, kw, oh, ow);
  if (offset == nullptr && conv2d_implicit_supported(cin / groups, kh, kw, oh, ow) && !small_launch_with_workspace) {
    const int cg = cin / groups, mg = cout / groups;
    for (int g = 0; g < groups; ++g) {
      Epilogue e = {bias ? bias + (size_t)g * mg : nullptr, nullptr, nullptr, nullptr, 0, act};
      const int rc = launch_conv2d_implicit(x + (size_t)g * cg * h * wd, weight + (size_t)g * mg * cg * kh * kw, y + (size_t)g * mg * oh * ow, n,
                                            cg, h, wd, mg, kh, kw, sh, sw, ph, pw, dh, dw, oh, ow, e, s, (int64_t)cin * h * wd,
                                            (int64_t)cout * oh * ow);
      if (rc) return rc;
    }
    return MV_OK;
  }
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
    if (!offset) a.offset_groups = 1;
    // channels per thread: as many as keeps >= ~500k threads in flight (the per-tap work is amortised over them)
    const int cog = cin / a.offset_groups;
    a.cchunks = 1;
    while (a.cchunks < cog && nb * a.offset_groups * a.cchunks * taps * ohw < 500000) a.cchunks *= 2;
    if (a.cchunks > cog) a.cchunks = cog;
    a.cper = (cog + a.cchunks - 1) / a.cchunks;
    a.cchunks = (cog + a.cper - 1) / a.cper;
    a.total = nb * a.offset_groups * a.cchunks * taps * ohw;
    if (a.total > 256LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "deform_conv2d: pass too large for one launch");
    // deformable sampling: the LDS-window kernel when its window fits (every usual DCN layer); plain im2col and exotic
    // geometries (huge stride x dilation) keep the direct gather
    DeformLdsArgs l = {};
    l.win_h = (kDefTH - 1) * sh + (kh - 1) * dh + 2 + 2 * kDefMargin;
    l.win_w = (kDefTW - 1) * sw + (kw - 1) * dw + 2 + 2 * kDefMargin;
    l.win_pitch = l.win_w | 1;
    const size_t lds = sizeof(float) * (size_t)kDefCB * l.win_h * l.win_pitch;
    const bool use_lds = offset != nullptr && lds <= 64 * 1024 && !tune_env("MV_DEFORM_DIRECT");
    if (use_lds) {
      l.x = a.x, l.offset = a.offset, l.mask = a.mask, l.col = a.col;
      l.cin = cin, l.h = h, l.wd = wd, l.kh = kh, l.kw = kw, l.sh = sh, l.sw = sw, l.ph = ph, l.pw = pw, l.dh = dh, l.dw = dw;
      l.oh = oh, l.ow = ow, l.offset_groups = offset_groups, l.use_mask = use_mask;
      l.tiles_x = (ow + kDefTW - 1) / kDefTW, l.tiles_y = (oh + kDefTH - 1) / kDefTH;
      // channel ranges: enough workgroups to fill the chip several times over (each re-reads the tile's offsets once per chunk)
      const int cog_l = cin / offset_groups;
      const long long tiles_all = (long long)nb * offset_groups * l.tiles_x * l.tiles_y;
      int chunks = (cog_l + kDefCB - 1) / kDefCB, per = chunks;  // chunks of kDefCB channels per workgroup
      while (per > 1 && tiles_all * ((chunks + per - 1) / per) < 4096) per = (per + 1) / 2;
      l.cper = per * kDefCB;
      l.csplit = (cog_l + l.cper - 1) / l.cper;
      const long long blocks = tiles_all * l.csplit;
      if (blocks > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "deform_conv2d: pass too large for one launch");
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_deform_im2col_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_deform_im2col_lds, dim3((unsigned)blocks), dim3(256), lds, s, l);
      if (int rc = check_launch("k_deform_im2col_lds")) return rc;
    } else {
      hipLaunchKernelGGL(k_deform_im2col, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
      if (int rc = check_launch("k_deform_im2col")) return rc;
    }
    for (int g = 0; g < groups; ++g) {
      Epilogue e = none;
      e.bias = bias ? bias + (size_t)g * mg : nullptr;
      e.act = act;
      const int rc = launch_conv1x1(a.col + (size_t)g * cg * taps * ohw, weight + (size_t)g * mg * cg * taps,
                                    y + ((size_t)b0 * cout + (size_t)g * mg) * ohw, nb, cg * taps, ohw, mg, e, s,
                                    (int64_t)cin * taps * ohw, (int64_t)cout * ohw, /*allow_k_slices=*/false);
      if (rc) return rc;
    }
  }
  return MV_OK;
}

}  // namespace mv
