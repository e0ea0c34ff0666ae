// dwf64.hip -- float64 images.  The reference computes a float64 image in float64: kernel taps built with dtype=float64,
// pad + conv2d in float64 (transforms/v2/functional/_misc.py:139-155; _color.py:246-275 for adjust_sharpness).  fp64 is a
// correctness row, not a bandwidth row (the transforms' documentation steers users to uint8 / float32), so one LDS-tiled
// kernel covers every odd kernel size up to 63 x 63 and all three borders:
//   k_dwf64      depthwise KY x KX, taps either the fp64 outer product k1d_y[j] * k1d_x[i] (one rounding per tap, as
//                `kernel1d_y.unsqueeze(-1) * kernel1d_x` does, _misc.py:97) or an explicit (ky, kx) device array; one
//                fma chain per output in row-major tap order from +0.0 -- bit-identical to oracle.c's orc_*_f64.
//   k_sharp_f64  adjust_sharpness on fp64 planes: valid 3x3, blend as one fma (v2) / two products (v1), clamp to [0, 1].
#include "mv_common.h"

namespace mv {

constexpr int kF64TileW = 64, kF64TileH = 16;

struct Taps1D64 {
  double x[kMaxTaps1D];
  double y[kMaxTaps1D];
};

struct DwF64Args {
  const double* x;
  double* y;
  const double* w2d;  // device (ky, kx) taps, or null: outer product of t.y / t.x
  int h, w, ky, kx, border;
  int tiles_x, tiles_y;
  Taps1D64 t;
};

__global__ __launch_bounds__(256) void k_dwf64(DwF64Args a) {
  extern __shared__ double tile[];
  const int rx = a.kx / 2, ry = a.ky / 2;
  const int lw = kF64TileW + a.kx - 1, lh = kF64TileH + a.ky - 1;
  const unsigned tiles = (unsigned)a.tiles_x * a.tiles_y;
  const size_t plane = blockIdx.x / tiles;
  const unsigned tt = blockIdx.x % tiles;
  const int x0 = (int)(tt % a.tiles_x) * kF64TileW, y0 = (int)(tt / a.tiles_x) * kF64TileH;
  const bool valid = a.border == MV_BORDER_VALID;
  // VALID: output pixel (oy, ox) reads input rows oy .. oy+ky-1; otherwise rows oy-ry .. oy+ry of the padded image
  const int iy0 = valid ? y0 : y0 - ry, ix0 = valid ? x0 : x0 - rx;
  const double* xp = a.x + plane * (size_t)a.h * a.w;
  for (int i = threadIdx.x; i < lw * lh; i += 256) {
    const int ly = i / lw, lx = i - ly * lw;
    int gy = iy0 + ly, gx = ix0 + lx;
    double v = 0.0;
    if (a.border == MV_BORDER_REFLECT) {
      gy = reflect_clamp(gy, a.h), gx = reflect_clamp(gx, a.w);
      v = xp[(size_t)gy * a.w + gx];
    } else if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) {
      v = xp[(size_t)gy * a.w + gx];
    }
    tile[i] = v;
  }
  __syncthreads();
  const int tx = threadIdx.x % kF64TileW, ty = threadIdx.x / kF64TileW;  // 64 x 4 threads, 4 rows each
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < a.ky; ++j) {
    const double wy = a.w2d ? 0.0 : a.t.y[j];  // device taps: kernel sides may exceed the by-value 1-D taps (kMaxTaps1D)
    for (int i = 0; i < a.kx; ++i) {
      const double wv = a.w2d ? a.w2d[j * a.kx + i] : wy * a.t.x[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fma(wv, tile[(ty + 4 * r + j) * lw + tx + i], acc[r]);
    }
  }
  const int oh = valid ? a.h - a.ky + 1 : a.h, ow = valid ? a.w - a.kx + 1 : a.w;
  double* yp = a.y + plane * (size_t)oh * ow;
  const int ox = x0 + tx;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = y0 + ty + 4 * r;
    if (oy < oh && ox < ow) yp[(size_t)oy * ow + ox] = acc[r];
  }
}

int launch_dwf64(const double* x, double* y, const double* w2d_dev, const double* k1d_x, const double* k1d_y, int64_t planes,
                 int h, int w, int ky, int kx, int border, hipStream_t s) {
  DwF64Args a;
  a.x = x, a.y = y, a.w2d = w2d_dev;
  a.h = h, a.w = w, a.ky = ky, a.kx = kx, a.border = border;
  const int oh = border == MV_BORDER_VALID ? h - ky + 1 : h, ow = border == MV_BORDER_VALID ? w - kx + 1 : w;
  a.tiles_x = (ow + kF64TileW - 1) / kF64TileW, a.tiles_y = (oh + kF64TileH - 1) / kF64TileH;
  for (int i = 0; i < kMaxTaps1D; ++i) a.t.x[i] = 0.0, a.t.y[i] = 0.0;
  if (!w2d_dev) {
    for (int i = 0; i < kx; ++i) a.t.x[i] = k1d_x[i];
    for (int i = 0; i < ky; ++i) a.t.y[i] = k1d_y[i];
  }
  const int64_t blocks = planes * a.tiles_x * a.tiles_y;
  if (blocks > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "fp64 filter: %lld tiles exceed one launch", (long long)blocks);
  const size_t lds = sizeof(double) * (size_t)(kF64TileW + kx - 1) * (kF64TileH + ky - 1);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_dwf64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_dwf64, dim3((unsigned)blocks), dim3(256), lds, s, a);
  return check_launchf("k_dwf64<%dx%d>", ky, kx);
}

// adjust_sharpness on fp64 planes (h, w > 2): one thread per pixel, neighbours straight from global memory (L1/L2-served)
struct SharpF64Args {
  const double* x;
  double* y;
  int h, w, v1;
  double factor, alpha;  // alpha = 1 - factor (Python double; the opmath type of a float64 tensor is double: no narrowing)
  int64_t total;
};

__global__ __launch_bounds__(256) void k_sharp_f64(SharpF64Args a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.total) return;
  const int64_t hw = (int64_t)a.h * a.w;
  const int64_t p = idx / hw;
  const int rem = (int)(idx - p * hw);
  const int oy = rem / a.w, ox = rem - oy * a.w;
  const double* xp = a.x + p * hw;
  const double xv = xp[rem];
  double out = xv;
  const double ka = 1.0 / 13.0, kb = 5.0 / 13.0;
  if (oy >= 1 && oy < a.h - 1 && ox >= 1 && ox < a.w - 1) {
    double acc = 0.0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        acc = fma((dy == 1 && dx == 1) ? kb : ka, xp[(size_t)(oy + dy - 1) * a.w + ox + dx - 1], acc);
    if (!a.v1) {
      out = fma(a.alpha, acc - xv, xv);  // view.add_(blurred.sub_(view), alpha = 1 - f): ATen's add is a fused multiply-add
    } else {
      out = a.factor * xv + a.alpha * acc;  // _blend: ratio * img1 + (1 - ratio) * img2
    }
  } else if (a.v1) {
    out = a.factor * xv + a.alpha * xv;
  }
  out = out < 0.0 ? 0.0 : (out > 1.0 ? 1.0 : out);
  a.y[idx] = out;
}

int launch_sharpness_f64(const double* x, double* y, int64_t planes, int h, int w, double factor, int v1, hipStream_t s) {
  SharpF64Args a = {x, y, h, w, v1, factor, 1.0 - factor, planes * (int64_t)h * w};
  const int64_t blocks = (a.total + 255) / 256;
  if (blocks > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "fp64 sharpness: problem too large for one launch");
  hipLaunchKernelGGL(k_sharp_f64, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return check_launch("k_sharp_f64");
}

}  // namespace mv
