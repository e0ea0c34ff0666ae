// abi.hip -- the extern "C" surface declared in include/mi355vision.h: argument validation (the same
// conditions the reference / ATen reject, reported as status codes instead of exceptions) and dispatch
// to the gfx950 kernels.  No allocation, no synchronisation, no environment lookups, no global state but the
// thread-local error / last-kernel strings.
#include <cstdlib>
#include <cstring>

#include "mv_common.h"
#include "mv_conv.h"
#include "mv_invres.h"
#include "mv_deform.h"

namespace mv {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static thread_local char g_kernel[160] = "";

int check_launch(const char* what) {
  snprintf(g_kernel, sizeof(g_kernel), "%s", what);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(MV_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MV_OK;
}

int check_launchf(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(MV_ERR_LAUNCH, "%s: %s", g_kernel, hipGetErrorString(e));
  return MV_OK;
}

// The frame table of the *_v call in progress on this thread: set for the duration of one dispatch, read by the launchers
// (mv_common.h: fill_frames), cleared before the entry point returns.  Call-scoped, never visible to another call.
static thread_local const FramePtrs* g_call_frames = nullptr;
const FramePtrs* call_frames() { return g_call_frames; }

// 3x3 single-output filters run on the LDS-halo-tile kernel (k_dwtile<3,3>): interleaved A/B on MI355X puts it
// 1.5-6 % ahead of the register-window kernel (k_dw3x3) and its HBM traffic is 1.000x algorithmic (vs 1.13x on
// the read side), profiles/r01_tune_dw3x3_5*.log.  MV_FORCE_REG3X3=1 selects the register kernel for A/B runs;
// the Sobel pair and adjust_sharpness always use k_dw3x3 (fused epilogues).
// Images up to 128 pixels wide (thumbnails) always take the register kernel: it packs 64 / lanes-per-row strips into a
// wave, where the 256-pixel tile would idle most of its lanes (4096x3x32x32: 101 -> 3x us).
static bool use_reg3x3(int wdt) {
  if (wdt <= 128) return true;
  const char* v = tune_env("MV_FORCE_REG3X3");
  return v && *v && *v != '0';
}

static int check_image(const void* x, const void* y, int64_t planes, int h, int w) {
  if (planes < 0 || h < 0 || w < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "negative size (planes=%lld h=%d w=%d)", (long long)planes, h, w);
  if (planes > 0 && h > 0 && w > 0 && (!x || !y)) return set_error(MV_ERR_INVALID_ARGUMENT, "null image pointer");
  if (x && x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return MV_OK;
}

static int check_kernel_size(int ky, int kx, int h, int w, int border) {
  if (ky <= 0 || kx <= 0 || (ky & 1) == 0 || (kx & 1) == 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "kernel size must be odd and positive, got (%d, %d)", ky, kx);
  if (border != MV_BORDER_VALID && border != MV_BORDER_REFLECT && border != MV_BORDER_ZERO)
    return set_error(MV_ERR_INVALID_ARGUMENT, "unknown border mode %d", border);
  if (border == MV_BORDER_REFLECT && (ky / 2 >= h || kx / 2 >= w))
    return set_error(MV_ERR_INVALID_ARGUMENT,
                     "reflect padding (%d, %d) must be smaller than the image (%d, %d)", ky / 2, kx / 2, h, w);
  if (border == MV_BORDER_VALID && (ky > h || kx > w))
    return set_error(MV_ERR_INVALID_ARGUMENT, "valid conv: kernel (%d, %d) larger than image (%d, %d)", ky, kx, h, w);
  return MV_OK;
}

static int check_taps1d(const float* k1d_x, int kx, const float* k1d_y, int ky) {
  if (!k1d_x || !k1d_y) return set_error(MV_ERR_INVALID_ARGUMENT, "null tap pointer");
  if (kx > kMaxTaps1D || ky > kMaxTaps1D)
    return set_error(MV_ERR_UNSUPPORTED, "1-D kernels up to %d taps are supported, got (%d, %d)", kMaxTaps1D, ky, kx);
  return MV_OK;
}

template <typename T>
static int depthwise(const T* x, T* y, const float* w, int w_on_device, int64_t planes, int h, int wdt, int ky, int kx,
                     int border, hipStream_t s) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, border)) return rc;
  if (!w) return set_error(MV_ERR_INVALID_ARGUMENT, "null tap pointer");
  if (!w_on_device && ky * kx > kMaxTaps2D)
    return set_error(MV_ERR_UNSUPPORTED, "%d host taps exceed MV_MAX_HOST_TAPS_2D=%d: pass a device pointer", ky * kx,
                     kMaxTaps2D);
  constexpr bool u8 = sizeof(T) == 1;
  // Host taps up to 11x11 of a size no specialised kernel is built for (7x5, 1x7, 3x9, ...) are zero-padded (centred) to
  // the next one that is: a zero tap is an exact no-op of the fma chain, and static fmas beat taps read from LDS one by one
  // by 3x (7x5 on 32 x 4K: 3.1 -> 1.5 ms).  Not for VALID borders (the output size follows the kernel size).
  float padded[121];
  if (!w_on_device && border != MV_BORDER_VALID && ky <= 11 && kx <= 11) {
    auto up = [](int k) { return k <= 3 ? 3 : (k <= 5 ? 5 : (k <= 7 ? 7 : (k <= 9 ? 9 : 11))); };
    int ty = up(ky), tx = up(kx);
    const bool pair_ok = (ty == tx) || (ty == 5 && tx == 3) || (ty == 3 && tx == 5) || (u8 && ty <= 7 && tx <= 7);
    if (!pair_ok) ty = tx = (ty > tx ? ty : tx);  // the tile kernels beyond 5x3 / 3x5 are square
    if ((ty != ky || tx != kx) && ty / 2 < h && tx / 2 < wdt) {
      for (int i = 0; i < ty * tx; ++i) padded[i] = 0.f;
      for (int j = 0; j < ky; ++j)
        for (int i = 0; i < kx; ++i) padded[((ty - ky) / 2 + j) * tx + (tx - kx) / 2 + i] = w[j * kx + i];
      w = padded, ky = ty, kx = tx;
    }
  }
  if constexpr (u8) {
    // the 16-pixel kernels narrow with a saturating pack; the reference's .to(uint8) of an out-of-range float is not
    // defined, so they only take averaging kernels (taps >= 0, sum <= 1), whose results lie in [0, 255] like a blur's
    bool averaging = !w_on_device;
    if (averaging) {
      double sum = 0.0;
      for (int i = 0; i < ky * kx; ++i) averaging = averaging && w[i] >= 0.f, sum += w[i];
      averaging = averaging && sum <= 1.0 + 1e-4;
    }
    if (averaging && ky == 3 && kx == 3 && border != MV_BORDER_VALID && dw3x3_u8x16_supported(x, y, h, wdt))
      return launch_dw3x3_u8x16(x, y, w, planes, h, wdt, border, 0, 0.0, s);
    if (averaging && dwk_u8x16_supported(x, y, h, wdt, ky, kx, border))
      return launch_dwk_u8x16(x, y, w, nullptr, nullptr, planes, h, wdt, ky, kx, border, s);
  }
  if (ky == 3 && kx == 3 && !w_on_device && border != MV_BORDER_VALID && use_reg3x3(wdt)) {
    if constexpr (u8)
      return launch_dw3x3_u8(x, y, w, planes, h, wdt, border, s);
    else
      return launch_dw3x3_f32(x, y, nullptr, w, nullptr, planes, h, wdt, border, s);
  }
  return launch_dwtile(x, y, u8, w_on_device ? nullptr : w, w_on_device ? w : nullptr, nullptr, nullptr, planes, h,
                       wdt, ky, kx, border, s);
}

template <typename T>
static int gaussian(const T* x, T* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx, const float* k1d_y,
                    int ky, hipStream_t s) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  constexpr bool u8 = sizeof(T) == 1;
  // Sides are zero-padded up to the next size the specialised kernels are built for (3 / 5 / 7 / 9 / 11; see the pairs
  // below): the padded outer product has exact zeros there, and a zero tap is an
  // exact no-op of the fma chain, so e.g. kernel_size = (1, 5) runs on the 3x5 kernel with the same bits.
  float pad_x[11], pad_y[11];
  if (kx <= 11 && ky <= 11) {
    auto up = [](int k) { return k <= 3 ? 3 : (k <= 5 ? 5 : (k <= 7 ? 7 : (k <= 9 ? 9 : 11))); };
    int tx = up(kx), ty = up(ky);
    // uint8 16-pixel kernels: any mix of 3 / 5 / 7; fp32 tile kernels: 3x3, 5x5, 5x3, 3x5, then squares 7 / 9 / 11
    const bool pair_ok = (tx == ty) || (tx <= 5 && ty <= 5) || (u8 && tx <= 7 && ty <= 7);
    if (!pair_ok) tx = ty = (tx > ty ? tx : ty);
    if ((tx != kx || ty != ky) && tx / 2 < wdt && ty / 2 < h) {
      for (int i = 0; i < 11; ++i) pad_x[i] = 0.f, pad_y[i] = 0.f;
      for (int i = 0; i < kx; ++i) pad_x[(tx - kx) / 2 + i] = k1d_x[i];
      for (int i = 0; i < ky; ++i) pad_y[(ty - ky) / 2 + i] = k1d_y[i];
      k1d_x = pad_x, k1d_y = pad_y, kx = tx, ky = ty;
    }
  }
  bool u8x16 = false;
  if constexpr (u8) u8x16 = (ky == 3 && kx == 3) && dw3x3_u8x16_supported(x, y, h, wdt);
  if (ky == 3 && kx == 3 && (use_reg3x3(wdt) || u8x16)) {
    float w9[9];  // kernel2d = k1d_y[:, None] * k1d_x  (_misc.py:97): one fp32 product per tap
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) w9[j * 3 + i] = k1d_y[j] * k1d_x[i];
    if constexpr (u8) {
      if (u8x16) return launch_dw3x3_u8x16(x, y, w9, planes, h, wdt, MV_BORDER_REFLECT, 0, 0.0, s);
    }
    if constexpr (u8)
      return launch_dw3x3_u8(x, y, w9, planes, h, wdt, MV_BORDER_REFLECT, s);
    else
      return launch_dw3x3_f32(x, y, nullptr, w9, nullptr, planes, h, wdt, MV_BORDER_REFLECT, s);
  }
  if constexpr (u8) {
    if (dwk_u8x16_supported(x, y, h, wdt, ky, kx, MV_BORDER_REFLECT))
      return launch_dwk_u8x16(x, y, nullptr, k1d_x, k1d_y, planes, h, wdt, ky, kx, MV_BORDER_REFLECT, s);
  }
  return launch_dwtile(x, y, u8, nullptr, nullptr, k1d_x, k1d_y, planes, h, wdt, ky, kx, MV_BORDER_REFLECT, s);
}

// ---- separately allocated frames, one launch per <= kMaxFrames frames --------------------------------------------------
// `dispatch(x0, y0, planes)` is the ordinary contiguous-batch dispatcher; it runs with the frame table in scope, so the
// kernels it launches take each plane's base from the table.  Frames whose pointers are not 16-byte aligned (the vector
// paths check the base pointer's alignment once per launch) go one launch per frame instead.
template <typename T, typename F>
static int for_frames(const T* const* xs, T* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt, F dispatch) {
  if (nframes < 0 || planes_per_frame < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "negative frame count / planes per frame");
  if (nframes == 0 || planes_per_frame == 0 || h == 0 || wdt == 0) return MV_OK;
  if (!xs || !ys) return set_error(MV_ERR_INVALID_ARGUMENT, "null frame pointer table");
  bool aligned = planes_per_frame <= 0x7fffffff;
  for (int i = 0; i < nframes; ++i) {
    if (!xs[i] || !ys[i]) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer for frame %d", i);
    if ((const void*)xs[i] == (const void*)ys[i]) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input (frame %d)", i);
    aligned = aligned && (uintptr_t)xs[i] % 16 == 0 && (uintptr_t)ys[i] % 16 == 0;
  }
  if (!aligned) {
    for (int i = 0; i < nframes; ++i)
      if (int rc = dispatch(xs[i], ys[i], planes_per_frame)) return rc;
    return MV_OK;
  }
  FramePtrs fp;
  fp.ppf = (int)planes_per_frame;
  for (int i0 = 0; i0 < nframes; i0 += kMaxFrames) {
    fp.n = nframes - i0 < kMaxFrames ? nframes - i0 : kMaxFrames;
    for (int i = 0; i < fp.n; ++i) fp.x[i] = xs[i0 + i], fp.y[i] = ys[i0 + i];
    g_call_frames = &fp;
    const int rc = dispatch(xs[i0], ys[i0], (int64_t)fp.n * planes_per_frame);
    g_call_frames = nullptr;
    if (rc) return rc;
  }
  return MV_OK;
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_abi_version(void) { return MV_ABI_VERSION; }

const char* mv_last_error(void) { return g_err; }

const char* mv_last_kernel(void) { return g_kernel; }

#ifndef MV_BUILD_ID
#define MV_BUILD_ID "unknown"
#endif
#ifdef MV_TUNING
const char* mv_build_id(void) { return MV_BUILD_ID "+tuning"; }
#else
const char* mv_build_id(void) { return MV_BUILD_ID; }
#endif

int mv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mv_depthwise_conv2d_f32(const float* x, float* y, const float* w, int w_on_device, int64_t planes, int h, int wdt,
                            int ky, int kx, int border, void* stream) {
  return depthwise<float>(x, y, w, w_on_device, planes, h, wdt, ky, kx, border, (hipStream_t)stream);
}

int mv_depthwise_conv2d_u8(const uint8_t* x, uint8_t* y, const float* w, int w_on_device, int64_t planes, int h,
                           int wdt, int ky, int kx, int border, void* stream) {
  return depthwise<uint8_t>(x, y, w, w_on_device, planes, h, wdt, ky, kx, border, (hipStream_t)stream);
}

int mv_gaussian_blur_f32(const float* x, float* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                         const float* k1d_y, int ky, void* stream) {
  return gaussian<float>(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

int mv_gaussian_blur_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                        const float* k1d_y, int ky, void* stream) {
  return gaussian<uint8_t>(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

// The same result as mv_gaussian_blur_u8 (the reference's single 2-D pass), bit for bit, at the cost of the separable pair:
// tiefix_u8.hip.  Taps must be what a Gaussian is -- non-negative, sum <= 1 -- for its error bound; anything else, and every
// size / width outside the separable kernels, takes the 2-D pass.
static bool taps_are_an_average(const float* k, int n) {
  double sum = 0.0;
  for (int i = 0; i < n; ++i) {
    if (!(k[i] >= 0.f)) return false;
    sum += k[i];
  }
  return sum <= 1.0 + 1e-5;
}

int64_t mv_gaussian_blur_u8_workspace_bytes(int64_t planes, int h, int wdt, int kx, int ky) {
  if (planes <= 0 || h <= 0 || wdt <= 0 || kx < 1 || ky < 1 || !(kx & 1) || !(ky & 1)) return 0;
  if (kx / 2 >= wdt || ky / 2 >= h || !gaussian_blur_u8_hybrid_supported(h, wdt, kx, ky)) return 0;
  const bool small = (kx <= 7 && ky <= 7) || (kx == 9 && (ky == 7 || ky == 9)) || (kx == 7 && ky == 9);
  return u8_tie_workspace_bytes(planes, h, wdt, small ? 16 : sepstream_u8_pixels_per_lane(kx, ky));
}

int mv_gaussian_blur_u8_ws(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                           const float* k1d_y, int ky, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  const bool fast = workspace != nullptr && workspace_bytes > 0 && gaussian_blur_u8_hybrid_supported(h, wdt, kx, ky) &&
                    taps_are_an_average(k1d_x, kx) && taps_are_an_average(k1d_y, ky);
  if (!fast) return gaussian<uint8_t>(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
  return launch_gaussian_blur_u8_hybrid(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, workspace, workspace_bytes, (hipStream_t)stream);
}

int mv_gaussian_blur_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                           const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream) {
  return for_frames<float>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const float* x, float* y, int64_t planes) {
    return gaussian<float>(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
  });
}

int mv_gaussian_blur_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                          const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream) {
  return for_frames<uint8_t>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const uint8_t* x, uint8_t* y, int64_t planes) {
    return gaussian<uint8_t>(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
  });
}

// float64 images: the reference computes them in float64 (taps, padding and conv2d all in the image dtype)
int mv_gaussian_blur_f64(const double* x, double* y, int64_t planes, int h, int wdt, const double* k1d_x, int kx,
                         const double* k1d_y, int ky, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (!k1d_x || !k1d_y) return set_error(MV_ERR_INVALID_ARGUMENT, "null tap pointer");
  if (kx > kMaxTaps1D || ky > kMaxTaps1D)
    return set_error(MV_ERR_UNSUPPORTED, "1-D kernels up to %d taps are supported, got (%d, %d): pass the 2-D kernel to "
                     "mv_depthwise_conv2d_f64 as a device array", kMaxTaps1D, ky, kx);
  return launch_dwf64(x, y, nullptr, k1d_x, k1d_y, planes, h, wdt, ky, kx, MV_BORDER_REFLECT, (hipStream_t)stream);
}

int mv_depthwise_conv2d_f64(const double* x, double* y, const double* w_dev, int64_t planes, int h, int wdt, int ky, int kx,
                            int border, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, border)) return rc;
  if (!w_dev) return set_error(MV_ERR_INVALID_ARGUMENT, "null tap pointer");
  if ((size_t)(64 + kx - 1) * (16 + ky - 1) * sizeof(double) > 160 * 1024)
    return set_error(MV_ERR_UNSUPPORTED, "fp64 filter: kernel (%d, %d) does not fit the LDS tile", ky, kx);
  return launch_dwf64(x, y, w_dev, nullptr, nullptr, planes, h, wdt, ky, kx, border, (hipStream_t)stream);
}

int mv_sharpness_f64(const double* x, double* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (!(sharpness_factor >= 0.0)) return set_error(MV_ERR_INVALID_ARGUMENT, "sharpness_factor (%g) is not non-negative.", sharpness_factor);
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (h <= 2 || wdt <= 2) {  // _color.py:240 returns the input unchanged
    if (hipMemcpyAsync(y, x, (size_t)planes * h * wdt * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return set_error(MV_ERR_LAUNCH, "sharpness: device copy failed");
    return MV_OK;
  }
  return launch_sharpness_f64(x, y, planes, h, wdt, sharpness_factor, v1, (hipStream_t)stream);
}

// fp16 / bf16 storage: the 2-D pass on the LDS tile kernel with fp32 arithmetic and one rounding on store.  Kernel sides up
// to 11 (zero-padded to the templated sizes like the fp32 path); larger ones return MV_ERR_UNSUPPORTED and the caller
// converts to fp32 around mv_separable_blur_f32.
static int gaussian_half(const void* x, void* y, int dtype, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                         const float* k1d_y, int ky, hipStream_t s) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  if (kx > 11 || ky > 11) return set_error(MV_ERR_UNSUPPORTED, "half-precision storage: kernel sides up to 11, got (%d, %d)", ky, kx);
  float pad_x[11], pad_y[11];
  auto up = [](int k) { return k <= 3 ? 3 : (k <= 5 ? 5 : (k <= 7 ? 7 : (k <= 9 ? 9 : 11))); };
  int tx = up(kx), ty = up(ky);
  if (!((tx == ty) || (tx <= 5 && ty <= 5))) tx = ty = (tx > ty ? tx : ty);
  if ((tx != kx || ty != ky) && tx / 2 < wdt && ty / 2 < h) {
    for (int i = 0; i < 11; ++i) pad_x[i] = 0.f, pad_y[i] = 0.f;
    for (int i = 0; i < kx; ++i) pad_x[(tx - kx) / 2 + i] = k1d_x[i];
    for (int i = 0; i < ky; ++i) pad_y[(ty - ky) / 2 + i] = k1d_y[i];
    k1d_x = pad_x, k1d_y = pad_y, kx = tx, ky = ty;
  }
  return launch_dwtile(x, y, dtype, nullptr, nullptr, k1d_x, k1d_y, planes, h, wdt, ky, kx, MV_BORDER_REFLECT, s);
}

int mv_gaussian_blur_f16(const void* x, void* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx, const float* k1d_y,
                         int ky, void* stream) {
  return gaussian_half(x, y, kDtF16, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

int mv_gaussian_blur_bf16(const void* x, void* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx, const float* k1d_y,
                          int ky, void* stream) {
  return gaussian_half(x, y, kDtBF16, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

// Unequal small kernels, e.g. GaussianBlur(kernel_size=(7, 3)): both 1-D kernels zero-padded (centred) to K = max(kx, ky, 3)
// in {3, 5, 7} run on the register-streaming k_sepfast<K>; a zero tap is an exact no-op of the fma chain, so the result
// equals the unpadded separable pair bit for bit (finite pixels).
struct PaddedTaps {
  float x[7], y[7];
  int k;
};
static bool pad_small_taps(const float* k1d_x, int kx, const float* k1d_y, int ky, PaddedTaps& p) {
  if (kx > 7 || ky > 7 || kx < 1 || ky < 1) return false;
  p.k = kx > ky ? kx : ky;
  if (p.k < 3) p.k = 3;
  for (int i = 0; i < 7; ++i) p.x[i] = 0.f, p.y[i] = 0.f;
  for (int i = 0; i < kx; ++i) p.x[(p.k - kx) / 2 + i] = k1d_x[i];
  for (int i = 0; i < ky; ++i) p.y[(p.k - ky) / 2 + i] = k1d_y[i];
  return true;
}

int mv_separable_blur_f32(const float* x, float* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                          const float* k1d_y, int ky, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  PaddedTaps pt;
  if (pad_small_taps(k1d_x, kx, k1d_y, ky, pt) && sepfast_supported(x, y, nullptr, h, wdt, pt.k, pt.k, false))
    return launch_sepfast(x, y, nullptr, nullptr, false, planes, h, wdt, pt.x, pt.y, pt.k, (hipStream_t)stream);
  if (sepstream_supported(x, y, false, h, wdt, kx, ky))
    return launch_sepstream(x, y, false, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
  return launch_separable(x, y, nullptr, nullptr, false, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

int mv_separable_blur_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                         const float* k1d_y, int ky, void* stream) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  const bool nine = (kx == 9 && (ky == 7 || ky == 9)) || (kx == 7 && ky == 9);
  if ((kx <= 7 && ky <= 7) || (nine && wdt >= 16)) {
    // small kernels: the 16-pixel-per-lane register kernel in its separable form; sides of 1 are zero-padded to 3 (an exact
    // no-op of both fma chains)
    float px[9], py[9];
    int tx = kx < 3 ? 3 : kx, ty = ky < 3 ? 3 : ky;
    for (int i = 0; i < 9; ++i) px[i] = 0.f, py[i] = 0.f;
    for (int i = 0; i < kx; ++i) px[(tx - kx) / 2 + i] = k1d_x[i];
    for (int i = 0; i < ky; ++i) py[(ty - ky) / 2 + i] = k1d_y[i];
    if (!sep_u8x16_supported(h, wdt, ty, tx))
      return set_error(MV_ERR_UNSUPPORTED, "separable uint8 blur with kernel sides <= 7 needs W >= 16 (got %d): use mv_gaussian_blur_u8", wdt);
    return launch_sep_u8x16(x, y, px, py, planes, h, wdt, ty, tx, (hipStream_t)stream);
  }
  if (!sepstream_supported(x, y, true, h, wdt, kx, ky))
    return set_error(MV_ERR_UNSUPPORTED, "separable uint8 blur with a kernel side above 7 needs K <= 63 and W >= 8");
  return launch_sepstream(x, y, true, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

static const float kSobelGX[9] = {-1.f, 0.f, 1.f, -2.f, 0.f, 2.f, -1.f, 0.f, 1.f};
static const float kSobelGY[9] = {-1.f, -2.f, -1.f, 0.f, 0.f, 0.f, 1.f, 2.f, 1.f};

int mv_sobel_f32(const float* x, float* gx, float* gy, int64_t planes, int h, int wdt, int border, void* stream) {
  if (int rc = check_image(x, gx, planes, h, wdt)) return rc;
  if (int rc = check_image(x, gy, planes, h, wdt)) return rc;
  if (gx && gx == gy) return set_error(MV_ERR_INVALID_ARGUMENT, "gx and gy must be distinct buffers");
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(3, 3, h, wdt, border)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (border == MV_BORDER_VALID) {
    if (int rc = launch_dwtile(x, gx, false, kSobelGX, nullptr, nullptr, nullptr, planes, h, wdt, 3, 3, border, s)) return rc;
    return launch_dwtile(x, gy, false, kSobelGY, nullptr, nullptr, nullptr, planes, h, wdt, 3, 3, border, s);
  }
  return launch_dw3x3_f32(x, gx, gy, kSobelGX, kSobelGY, planes, h, wdt, border, s);
}

int mv_gaussian_sobel_f32(const float* x, float* gx, float* gy, int64_t planes, int h, int wdt, const float* k1d_x,
                          int kx, const float* k1d_y, int ky, void* stream) {
  if (int rc = check_image(x, gx, planes, h, wdt)) return rc;
  if (int rc = check_image(x, gy, planes, h, wdt)) return rc;
  if (gx && gx == gy) return set_error(MV_ERR_INVALID_ARGUMENT, "gx and gy must be distinct buffers");
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (int rc = check_kernel_size(ky, kx, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_kernel_size(3, 3, h, wdt, MV_BORDER_REFLECT)) return rc;
  if (int rc = check_taps1d(k1d_x, kx, k1d_y, ky)) return rc;
  PaddedTaps pt;
  if (pad_small_taps(k1d_x, kx, k1d_y, ky, pt) && sepfast_supported(x, gx, gy, h, wdt, pt.k, pt.k, true))
    return launch_sepfast(x, nullptr, gx, gy, true, planes, h, wdt, pt.x, pt.y, pt.k, (hipStream_t)stream);
  return launch_separable(x, nullptr, gx, gy, true, planes, h, wdt, k1d_x, kx, k1d_y, ky, (hipStream_t)stream);
}

static int sharpness(const void* x, void* y, bool u8, int64_t planes, int h, int wdt, double f, int v1, float bound,
                     int round_blur, hipStream_t s) {
  if (int rc = check_image(x, y, planes, h, wdt)) return rc;
  if (!(f >= 0.0)) return set_error(MV_ERR_INVALID_ARGUMENT, "sharpness_factor (%g) is not non-negative.", f);
  if (planes == 0 || h == 0 || wdt == 0) return MV_OK;
  if (h <= 2 || wdt <= 2) {  // _color.py:240 returns the input unchanged
    const size_t bytes = (size_t)planes * h * wdt * (u8 ? 1 : 4);
    if (hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return set_error(MV_ERR_LAUNCH, "sharpness: device copy failed");
    return MV_OK;
  }
  if (u8 && dw3x3_u8x16_supported((const uint8_t*)x, (const uint8_t*)y, h, wdt))
    return launch_dw3x3_u8x16((const uint8_t*)x, (uint8_t*)y, nullptr, planes, h, wdt, MV_BORDER_ZERO, v1 ? 3 : 2, f, s);
  return launch_sharpness(x, y, u8, planes, h, wdt, f, v1, bound, round_blur, s);
}

int mv_sharpness_f32(const float* x, float* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1,
                     float bound, int integer_semantics, void* stream) {
  if (!(bound > 0.f)) return set_error(MV_ERR_INVALID_ARGUMENT, "sharpness: bound must be positive");
  return sharpness(x, y, false, planes, h, wdt, sharpness_factor, v1, bound, integer_semantics, (hipStream_t)stream);
}

int mv_sharpness_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1,
                    void* stream) {
  return sharpness(x, y, true, planes, h, wdt, sharpness_factor, v1, 255.f, 1, (hipStream_t)stream);
}

int mv_separable_blur_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                            const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream) {
  return for_frames<float>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const float* x, float* y, int64_t planes) {
    return mv_separable_blur_f32(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, stream);
  });
}

int mv_separable_blur_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                           const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream) {
  return for_frames<uint8_t>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const uint8_t* x, uint8_t* y, int64_t planes) {
    return mv_separable_blur_u8(x, y, planes, h, wdt, k1d_x, kx, k1d_y, ky, stream);
  });
}

int mv_sharpness_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                       double sharpness_factor, int v1, float bound, int integer_semantics, void* stream) {
  if (!(bound > 0.f)) return set_error(MV_ERR_INVALID_ARGUMENT, "sharpness: bound must be positive");
  if ((h <= 2 || wdt <= 2) && xs && ys) {  // the input is returned unchanged: one device copy per frame
    for (int i = 0; i < nframes; ++i)
      if (int rc = sharpness(xs[i], ys[i], false, planes_per_frame, h, wdt, sharpness_factor, v1, bound, integer_semantics, (hipStream_t)stream)) return rc;
    return MV_OK;
  }
  return for_frames<float>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const float* x, float* y, int64_t planes) {
    return sharpness(x, y, false, planes, h, wdt, sharpness_factor, v1, bound, integer_semantics, (hipStream_t)stream);
  });
}

int mv_sharpness_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                      double sharpness_factor, int v1, void* stream) {
  if ((h <= 2 || wdt <= 2) && xs && ys) {
    for (int i = 0; i < nframes; ++i)
      if (int rc = sharpness(xs[i], ys[i], true, planes_per_frame, h, wdt, sharpness_factor, v1, 255.f, 1, (hipStream_t)stream)) return rc;
    return MV_OK;
  }
  return for_frames<uint8_t>(xs, ys, nframes, planes_per_frame, h, wdt, [&](const uint8_t* x, uint8_t* y, int64_t planes) {
    return sharpness(x, y, true, planes, h, wdt, sharpness_factor, v1, 255.f, 1, (hipStream_t)stream);
  });
}

int mv_conv3x3_bias_relu_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h,
                             int wdt, int cout, int relu, void* stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || h < 0 || wdt < 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad conv shape n=%lld cin=%d cout=%d h=%d w=%d", (long long)n, cin, cout, h, wdt);
  if (n == 0 || h == 0 || wdt == 0) return MV_OK;
  if (!x || !w || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  if (conv3x3_c3_supported(x, y, cin, cout, h, wdt))
    return launch_conv3x3_c3(x, false, nullptr, nullptr, w, b, y, n, h, wdt, cout, relu, (hipStream_t)stream);
  if (conv3x3_gen_supported(cin, cout, h, wdt))
    return launch_conv3x3_gen(x, w, b, y, n, cin, h, wdt, cout, relu, (hipStream_t)stream);
  return launch_conv3x3(x, w, b, y, n, cin, h, wdt, cout, relu, (hipStream_t)stream);
}

int mv_conv3x3_k_slices(int64_t n, int cin, int h, int wdt, int cout, int* slice_channels) {
  int slices = 1, sc = cin;
  if (n > 0 && cin > 4 && cout > 0 && h > 0 && wdt > 0) conv3x3_gen_plan(n, cin, h, wdt, cout, &slices, &sc);
  if (slice_channels) *slice_channels = sc;
  return slices;
}

int64_t mv_conv3x3_workspace_bytes(int64_t n, int cin, int h, int wdt, int cout) {
  if (n <= 0 || cin <= 4 || cout <= 0 || h <= 0 || wdt <= 0) return 0;
  return conv3x3_gen_workspace_bytes(n, cin, h, wdt, cout);
}

int mv_conv3x3_bias_relu_ws_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt, int cout,
                                int relu, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || h < 0 || wdt < 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad conv shape n=%lld cin=%d cout=%d h=%d w=%d", (long long)n, cin, cout, h, wdt);
  if (n == 0 || h == 0 || wdt == 0) return MV_OK;
  if (!x || !w || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  const int64_t need = mv_conv3x3_workspace_bytes(n, cin, h, wdt, cout);
  if (need == 0 || conv3x3_c3_supported(x, y, cin, cout, h, wdt) || !conv3x3_gen_supported(cin, cout, h, wdt))
    return mv_conv3x3_bias_relu_f32(x, w, b, y, n, cin, h, wdt, cout, relu, stream);  // this shape runs its single chain
  if (workspace == nullptr || workspace_bytes < need)
    return set_error(MV_ERR_INVALID_ARGUMENT, "conv3x3: workspace of %lld bytes needed (mv_conv3x3_workspace_bytes), got %lld",
                     (long long)need, (long long)workspace_bytes);
  return launch_conv3x3_gen_ws(x, w, b, y, n, cin, h, wdt, cout, relu, (hipStream_t)stream, workspace, workspace_bytes);
}

int mv_conv3x3_bias_relu_u8norm_f32(const uint8_t* x, const float* mean3, const float* std3, const float* w, const float* b,
                                    float* y, int64_t n, int h, int wdt, int cout, int relu, void* stream) {
  if (n < 0 || cout <= 0 || h < 0 || wdt < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad conv shape");
  if (n == 0 || h == 0 || wdt == 0) return MV_OK;
  if (!x || !w || !y || !mean3 || !std3) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  for (int i = 0; i < 3; ++i)
    if (std3[i] == 0.f) return set_error(MV_ERR_INVALID_ARGUMENT, "std evaluated to zero, leading to division by zero.");
  if (cout > 64 || wdt % 4 != 0 || (uintptr_t)y % 16 != 0 || (size_t)cout * h * wdt * sizeof(float) >= (1ull << 32))
    return set_error(MV_ERR_UNSUPPORTED, "fused uint8 first layer needs cout <= 64, W %% 4 == 0 and a 16-byte aligned output");
  return launch_conv3x3_c3(x, true, mean3, std3, w, b, y, n, h, wdt, cout, relu, (hipStream_t)stream);
}

int mv_to_float_normalize_u8(const uint8_t* x, float* y, int64_t n, int c, int64_t hw, const float* mean, const float* stdv,
                             void* stream) {
  if (n < 0 || c <= 0 || hw < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad shape");
  if (n == 0 || hw == 0) return MV_OK;
  if (!x || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if ((mean == nullptr) != (stdv == nullptr)) return set_error(MV_ERR_INVALID_ARGUMENT, "mean and std go together");
  if (stdv)
    for (int i = 0; i < c && i < 16; ++i)
      if (stdv[i] == 0.f) return set_error(MV_ERR_INVALID_ARGUMENT, "std evaluated to zero, leading to division by zero.");
  return launch_to_float_normalize(x, y, true, n, c, hw, mean, stdv, (hipStream_t)stream);
}

int mv_normalize_f32(const float* x, float* y, int64_t n, int c, int64_t hw, const float* mean, const float* stdv,
                     void* stream) {
  if (n < 0 || c <= 0 || hw < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad shape");
  if (n == 0 || hw == 0) return MV_OK;
  if (!x || !y || !mean || !stdv) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  for (int i = 0; i < c && i < 16; ++i)
    if (stdv[i] == 0.f) return set_error(MV_ERR_INVALID_ARGUMENT, "std evaluated to zero, leading to division by zero.");
  return launch_to_float_normalize(x, y, false, n, c, hw, mean, stdv, (hipStream_t)stream);
}

int mv_maxpool2x2_f32(const float* x, float* y, int64_t planes, int h, int wdt, void* stream) {
  if (planes < 0 || h < 0 || wdt < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "negative size");
  if (planes == 0 || h < 2 || wdt < 2) return MV_OK;
  if (!x || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return launch_maxpool2x2(x, y, planes, h, wdt, (hipStream_t)stream);
}

int mv_linear_bias_relu_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                            void* stream) {
  if (n < 0 || k <= 0 || m <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad linear shape n=%lld k=%d m=%d", (long long)n, k, m);
  if (n == 0) return MV_OK;
  if (!x || !w || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return launch_linear(x, w, b, y, n, k, m, relu, (hipStream_t)stream);
}

int mv_linear_k_slices(int64_t n, int k, int m, int* slice_len) {
  int slices = 1, len = k;
  if (n >= 0 && k > 0 && m > 0) linear_plan(n, k, m, &slices, &len);
  if (slice_len) *slice_len = len;
  return slices;
}

int64_t mv_linear_workspace_bytes(int64_t n, int k, int m) {
  if (n <= 0 || k <= 0 || m <= 0) return 0;
  int slices, len;
  linear_plan(n, k, m, &slices, &len);
  return slices > 1 ? (int64_t)slices * n * m * (int64_t)sizeof(float) : 0;
}

int mv_linear_bias_relu_ws_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || k <= 0 || m <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad linear shape n=%lld k=%d m=%d", (long long)n, k, m);
  if (n == 0) return MV_OK;
  if (!x || !w || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  if (n > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "linear: problem too large for one launch");
  const int64_t need = mv_linear_workspace_bytes(n, k, m);
  if (need == 0) return launch_linear(x, w, b, y, n, k, m, relu, (hipStream_t)stream);
  if (!workspace || workspace_bytes < need)
    return set_error(MV_ERR_INVALID_ARGUMENT, "linear: workspace of %lld bytes needed (mv_linear_workspace_bytes), got %lld",
                     (long long)need, (long long)workspace_bytes);
  if ((uintptr_t)workspace % 4 != 0) return set_error(MV_ERR_INVALID_ARGUMENT, "linear: workspace must be 4-byte aligned");
  return launch_linear_sliced(x, w, b, y, n, k, m, relu, static_cast<float*>(workspace), (hipStream_t)stream);
}

int mv_adaptive_avgpool_f32(const float* x, float* y, int64_t planes, int h, int wdt, int oh, int ow, void* stream) {
  if (planes < 0 || h <= 0 || wdt <= 0 || oh <= 0 || ow <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad pooling shape");
  if (planes == 0) return MV_OK;
  if (!x || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return launch_adaptive_avgpool(x, y, planes, h, wdt, oh, ow, (hipStream_t)stream);
}

int mv_conv_norm_act_f32(int kind, const float* x, const float* w, const float* bias, const float* alpha, const float* beta,
                         const float* residual, float* y, int64_t n, int cin, int h, int wdt, int cout, int stride,
                         int affine, int act, void* stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || h <= 0 || wdt <= 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad conv shape n=%lld cin=%d cout=%d h=%d w=%d", (long long)n, cin, cout, h, wdt);
  if (stride != 1 && stride != 2) return set_error(MV_ERR_INVALID_ARGUMENT, "stride should be 1 or 2 instead of %d", stride);
  if (affine < MV_AFFINE_NONE || affine > MV_AFFINE_FMA || act < MV_ACT_NONE || act > MV_ACT_SILU)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad affine (%d) / activation (%d) code", affine, act);
  if (n == 0) return MV_OK;
  if (!x || !w || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (affine != MV_AFFINE_NONE && (!alpha || !beta)) return set_error(MV_ERR_INVALID_ARGUMENT, "affine without alpha / beta");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  Epilogue e = {bias, alpha, beta, residual, affine, act};
  switch (kind) {
    case MV_CONV_DENSE3X3:
      if (cin > 4) return set_error(MV_ERR_UNSUPPORTED, "dense 3x3 with norm: cin <= 4 (the stem); got %d", cin);
      return launch_conv3x3_smallcin(x, w, y, n, cin, h, wdt, cout, stride, e, (hipStream_t)stream);
    case MV_CONV_DW3X3:
      if (cin != cout) return set_error(MV_ERR_INVALID_ARGUMENT, "depthwise: cin (%d) != cout (%d)", cin, cout);
      return launch_dwpc3x3(x, w, y, n, cin, h, wdt, stride, e, (hipStream_t)stream);
    case MV_CONV_PW1X1:
      if (stride != 1) return set_error(MV_ERR_UNSUPPORTED, "pointwise conv: stride 1 only");
      return launch_conv1x1(x, w, y, n, cin, (int64_t)h * wdt, cout, e, (hipStream_t)stream);
  }
  return set_error(MV_ERR_INVALID_ARGUMENT, "unknown conv kind %d", kind);
}

int mv_conv1x1_k_slices(int64_t n, int cin, int h, int wdt, int cout, int* slice_len) {
  int slices = 1, len = cin;
  if (n > 0 && cin > 0 && cout > 0 && h > 0 && wdt > 0) conv1x1_plan(n, cin, (int64_t)h * wdt, cout, &slices, &len);
  if (slice_len) *slice_len = len;
  return slices;
}

int mv_inverted_residual_k_slices(int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride, int* slice_len) {
  int slices = 0, len = hidden;
  if (!invres_plan(n, cin, hidden, cout, h, wdt, stride, &slices, &len)) slices = 0, len = hidden;
  if (slice_len) *slice_len = len;
  return slices;
}

int64_t mv_inverted_residual_workspace_bytes(int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride) {
  return invres_workspace_bytes(n, cin, hidden, cout, h, wdt, stride);
}

int mv_inverted_residual_f32(const float* x, const float* w_expand, const float* a1, const float* b1, const float* w_dw,
                             const float* a2, const float* b2, const float* w_project, const float* a3, const float* b3,
                             int residual, float* y, int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride,
                             int affine, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || cin <= 0 || hidden <= 0 || cout <= 0 || h <= 0 || wdt <= 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad inverted_residual shape n=%lld %d -> %d -> %d on %d x %d", (long long)n, cin, hidden, cout, h, wdt);
  if (stride != 1 && stride != 2) return set_error(MV_ERR_INVALID_ARGUMENT, "stride should be 1 or 2 instead of %d", stride);
  if (affine != MV_AFFINE_MUL_ADD && affine != MV_AFFINE_FMA)
    return set_error(MV_ERR_INVALID_ARGUMENT, "inverted_residual: affine must be MV_AFFINE_MUL_ADD or MV_AFFINE_FMA (every conv of the block is followed by a norm)");
  if (n == 0) return MV_OK;
  // a block without expansion (expand_ratio 1: hidden == cin) has no w_expand / a1 / b1
  if (!x || !w_dw || !a2 || !b2 || !w_project || !a3 || !b3 || !y || (hidden != cin && (!w_expand || !a1 || !b1))) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return launch_invres(x, w_expand, a1, b1, w_dw, a2, b2, w_project, a3, b3, residual, y, n, cin, hidden, cout, h, wdt, stride, affine,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

void mv_fold_batchnorm(const float* weight, const float* bias, const float* mean, const float* var, double eps, int c,
                       float* alpha, float* beta) {
  for (int i = 0; i < c; ++i) {
    const float invstd = 1.0f / sqrtf(var[i] + (float)eps);
    const float a = invstd * (weight ? weight[i] : 1.0f);
    alpha[i] = a;
    beta[i] = fmaf(-mean[i], a, bias ? bias[i] : 0.0f);
  }
}

static int deform_out(int h, int wdt, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int* oh, int* ow) {
  if (h <= 0 || wdt <= 0 || kh <= 0 || kw <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad deform_conv2d shape (%d, %d) kernel (%d, %d)", h, wdt, kh, kw);
  if (sh <= 0 || sw <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "stride_h: %d stride_w: %d", sh, sw);  // deform_conv2d_kernel.cpp:1004-1006
  if (ph < 0 || pw < 0) return set_error(MV_ERR_INVALID_ARGUMENT, "pad_h: %d pad_w: %d", ph, pw);
  if (dh <= 0 || dw <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "dil_h: %d dil_w: %d", dh, dw);
  *oh = (h + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  *ow = (wdt + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (*oh <= 0 || *ow <= 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "Calculated output size too small - out_h: %d out_w: %d", *oh, *ow);  // :1040-1045
  return MV_OK;
}

int64_t mv_deform_conv2d_workspace_bytes(int64_t images, int cin, int h, int wdt, int kh, int kw, int stride_h, int stride_w,
                                         int pad_h, int pad_w, int dilation_h, int dilation_w) {
  int oh = 0, ow = 0;
  if (images <= 0 || cin <= 0 || deform_out(h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, &oh, &ow)) return 0;
  return images * deform_workspace_bytes_per_image(cin, kh, kw, oh, ow);
}

int mv_deform_conv2d_needs_workspace(int64_t images, int cin, int cout, int h, int wdt, int kh, int kw, int stride_h, int stride_w,
                                     int pad_h, int pad_w, int dilation_h, int dilation_w, int groups, int offset_groups) {
  int oh = 0, ow = 0;
  if (images <= 0 || cin <= 0 || cout <= 0 || groups <= 0 || offset_groups <= 0 || cin % groups || cout % groups || cin % offset_groups ||
      deform_out(h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, &oh, &ow))
    return 1;
  if (tune_env("MV_DEFORM_UNFUSED")) return 1;
  const int64_t wgs = deform_fused_workgroups(images, cin, cout, h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                              groups, offset_groups);
  return wgs == 0 ? 1 : (wgs >= kDeformFusedMinWorkgroups ? 0 : 2);
}

int mv_deform_conv2d_f32(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                         int64_t n, int cin, int h, int wdt, int cout, int kh, int kw, int stride_h, int stride_w, int pad_h,
                         int pad_w, int dilation_h, int dilation_w, int groups, int offset_groups, int use_mask, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  int oh = 0, ow = 0;
  if (n < 0 || cin <= 0 || cout <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad deform_conv2d shape n=%lld cin=%d cout=%d", (long long)n, cin, cout);
  if (int rc = deform_out(h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, &oh, &ow)) return rc;
  if (groups <= 0 || offset_groups <= 0 || cin % groups || cout % groups || cin % offset_groups)
    return set_error(MV_ERR_INVALID_ARGUMENT, "channels (%d -> %d) must divide into %d weight groups and %d offset groups", cin, cout,
                     groups, offset_groups);
  if (n == 0) return MV_OK;
  if (!x || !weight || !offset || !y || (use_mask && !mask)) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  return launch_deform_conv2d(x, weight, offset, mask, bias, y, n, cin, h, wdt, cout, kh, kw, stride_h, stride_w, pad_h, pad_w,
                              dilation_h, dilation_w, groups, offset_groups, use_mask, workspace, workspace_bytes, (hipStream_t)stream);
}

int mv_conv2d_needs_workspace(int64_t n, int cin, int cout, int h, int wdt, int kh, int kw, int stride_h, int stride_w, int pad_h,
                              int pad_w, int dilation_h, int dilation_w, int groups) {
  int oh = 0, ow = 0;
  if (groups <= 0 || cin <= 0 || cin % groups || cout <= 0 || cout % groups) return 1;
  if (deform_out(h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, &oh, &ow)) return 1;
  if (!conv2d_implicit_supported(cin / groups, kh, kw, oh, ow)) return 1;
  return conv2d_implicit_min_workgroups(n, cout / groups, oh, ow) * groups < 128 ? 2 : 0;  // 2: optional -- a handful of workgroups
}

int mv_conv2d_bias_act_f32(const float* x, const float* weight, const float* bias, float* y, int64_t n, int cin, int h, int wdt,
                           int cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                           int dilation_w, int groups, int act, void* workspace, int64_t workspace_bytes, void* stream) {
  int oh = 0, ow = 0;
  if (n < 0 || cin <= 0 || cout <= 0) return set_error(MV_ERR_INVALID_ARGUMENT, "bad conv2d shape n=%lld cin=%d cout=%d", (long long)n, cin, cout);
  if (int rc = deform_out(h, wdt, kh, kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, &oh, &ow)) return rc;
  if (groups <= 0 || cin % groups || cout % groups)
    return set_error(MV_ERR_INVALID_ARGUMENT, "channels (%d -> %d) must divide into %d groups", cin, cout, groups);
  if (act < MV_ACT_NONE || act > MV_ACT_SILU) return set_error(MV_ERR_INVALID_ARGUMENT, "bad activation code %d", act);
  if (n == 0) return MV_OK;
  if (!x || !weight || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  return launch_deform_conv2d(x, weight, nullptr, nullptr, bias, y, n, cin, h, wdt, cout, kh, kw, stride_h, stride_w, pad_h, pad_w,
                              dilation_h, dilation_w, groups, 1, 0, workspace, workspace_bytes, (hipStream_t)stream, act);
}

int mv_maxpool2d_f32(const float* x, float* y, int64_t planes, int h, int wdt, int k, int stride, void* stream) {
  if (planes < 0 || h <= 0 || wdt <= 0 || k <= 0 || stride <= 0 || k > h || k > wdt)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad pooling shape planes=%lld (%d, %d) k=%d stride=%d", (long long)planes, h, wdt, k, stride);
  if (planes == 0) return MV_OK;
  if (!x || !y) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return launch_maxpool2d(x, y, planes, h, wdt, k, stride, (hipStream_t)stream);
}

static int check_resize(const void* x, const void* y, int64_t planes, int h, int wdt, int oh, int ow, int ch, int cw) {
  if (planes < 0 || h <= 0 || wdt <= 0 || oh <= 0 || ow <= 0 || ch <= 0 || cw <= 0)
    return set_error(MV_ERR_INVALID_ARGUMENT, "bad resize shape planes=%lld (%d, %d) -> (%d, %d), window (%d, %d)",
                     (long long)planes, h, wdt, oh, ow, ch, cw);
  if (planes > 0 && (!x || !y)) return set_error(MV_ERR_INVALID_ARGUMENT, "null pointer");
  if (x == y && planes > 0) return set_error(MV_ERR_INVALID_ARGUMENT, "output must not alias input");
  return MV_OK;
}

int64_t mv_resize_workspace_bytes(int64_t planes, int h, int wdt, int oh, int ow, int crop_top, int crop_left, int crop_h,
                                  int crop_w) {
  if (planes <= 0 || h <= 0 || wdt <= 0 || oh <= 0 || ow <= 0 || crop_h <= 0 || crop_w <= 0) return 0;
  return resize_workspace_bytes(planes, h, wdt, oh, ow, crop_top, crop_left, crop_h, crop_w);
}

int mv_resize_bilinear_aa_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, int oh, int ow, int crop_top,
                             int crop_left, int crop_h, int crop_w, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_resize(x, y, planes, h, wdt, oh, ow, crop_h, crop_w)) return rc;
  if (planes == 0) return MV_OK;
  return launch_resize(x, y, true, planes, 1, h, wdt, oh, ow, crop_top, crop_left, crop_h, crop_w, 0, nullptr, nullptr,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

int mv_resize_bilinear_aa_f32(const float* x, float* y, int64_t planes, int h, int wdt, int oh, int ow, int crop_top,
                              int crop_left, int crop_h, int crop_w, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_resize(x, y, planes, h, wdt, oh, ow, crop_h, crop_w)) return rc;
  if (planes == 0) return MV_OK;
  return launch_resize(x, y, false, planes, 1, h, wdt, oh, ow, crop_top, crop_left, crop_h, crop_w, 0, nullptr, nullptr,
                       workspace, workspace_bytes, (hipStream_t)stream);
}

static int preset(const void* x, float* y, bool u8, int64_t n, int c, int h, int wdt, int oh, int ow, int ct, int cl, int ch,
                  int cw, const float* mean, const float* stdv, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || c <= 0 || c > 4) return set_error(MV_ERR_INVALID_ARGUMENT, "preset: n=%lld, c=%d (1..4 channels)", (long long)n, c);
  if (int rc = check_resize(x, y, n * c, h, wdt, oh, ow, ch, cw)) return rc;
  if (n == 0) return MV_OK;
  if (!mean || !stdv) return set_error(MV_ERR_INVALID_ARGUMENT, "preset: null mean / std");
  for (int i = 0; i < c; ++i)
    if (stdv[i] == 0.f) return set_error(MV_ERR_INVALID_ARGUMENT, "std evaluated to zero, leading to division by zero.");
  return launch_resize(x, y, u8, n * c, c, h, wdt, oh, ow, ct, cl, ch, cw, 1, mean, stdv, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

int mv_preset_classification_u8(const uint8_t* x, float* y, int64_t n, int c, int h, int wdt, int oh, int ow, int crop_top,
                                int crop_left, int crop_h, int crop_w, const float* mean, const float* stdv,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  return preset(x, y, true, n, c, h, wdt, oh, ow, crop_top, crop_left, crop_h, crop_w, mean, stdv, workspace,
                workspace_bytes, stream);
}

int mv_preset_classification_f32(const float* x, float* y, int64_t n, int c, int h, int wdt, int oh, int ow, int crop_top,
                                 int crop_left, int crop_h, int crop_w, const float* mean, const float* stdv,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  return preset(x, y, false, n, c, h, wdt, oh, ow, crop_top, crop_left, crop_h, crop_w, mean, stdv, workspace,
                workspace_bytes, stream);
}

}  // extern "C"
