// resize.hip -- F.resize(bilinear, antialias=True) [+ center_crop] [+ convert_image_dtype(float) + normalize]:
// the head of the ImageClassification preset (transforms/_presets.py:56-63) on gfx950.
//
// The tensor path of F.resize (transforms/_functional_tensor.py:441-474) is .to(float32) -> interpolate(bilinear,
// align_corners=False, antialias=True) -> torch.round + cast back for integer images.  interpolate is ATen's separable
// _upsample_bilinear2d_aa (width pass into an fp32 temporary, then height pass; per output index a window
// [xmin, xmin + xsize) of triangle weights normalised by their fp32 sum; out = src0*w0 then one fma per further tap).
// Both kernels below follow oracle/oracle.c's restatement of it operation for operation (same float / double
// intermediates in the index and weight arithmetic), so results are bit-identical to the oracle, which is pinned
// bit-exactly to the reference's own outputs (tests/golden/resize_preset.npz).
//
//   k_resize_w  thread = one (plane, needed input row, needed resized column): the width pass, restricted to the rows
//               the height pass will read and the columns of the crop window -> fp32 workspace
//   k_resize_h  thread = one output pixel of the crop window: height pass from the workspace, then the epilogue:
//               round_() + narrow (uint8 images), or the preset tail `/ 255 -> (v - mean) / std` in fp32
// Every thread recomputes the weights of its own window (2*scale+1 taps; a few dozen flops) instead of reading a
// table: no setup launch, no host->device copy, nothing to capture besides two kernel nodes.  The path is
// launch- and latency-bound per image (a 500x375 photo is 0.5 MB); batches of equal-sized images amortise it.
#include "mv_common.h"

namespace mv {

struct AxisAA {       // one interpolation axis
  int in, out;        // input / resized size
  float scale, support, invscale;
  int max_interp;
  int identity;       // out == in: ATen skips the pass
};

__host__ __device__ inline AxisAA make_axis(int in, int out) {
  AxisAA a;
  a.in = in, a.out = out;
  a.scale = (float)in / out;
  a.support = (a.scale >= 1.0f) ? a.scale : 1.0f;
  a.invscale = (a.scale >= 1.0f) ? (float)(1.0 / a.scale) : 1.0f;
  a.max_interp = (int)ceilf(a.support) * 2 + 1;
  a.identity = (in == out);
  return a;
}

// window of output index i: [xmin, xmin + xsize), and its centre
__host__ __device__ inline void aa_window(const AxisAA& a, int i, int& xmin, int& xsize, float& center) {
  center = (float)(a.scale * (i + 0.5));
  long long lo = (long long)((center - a.support) + 0.5);
  if (lo < 0) lo = 0;
  long long hi = (long long)((center + a.support) + 0.5);
  if (hi > a.in) hi = a.in;
  long long n = hi - lo;
  if (n < 0) n = 0;
  if (n > a.max_interp) n = a.max_interp;
  xmin = (int)lo, xsize = (int)n;
}

__device__ inline float aa_weight(const AxisAA& a, int j, int xmin, float center) {
  float x = (float)(((float)(j + xmin) - center + 0.5) * a.invscale);
  x = __builtin_fabsf(x);
  return x < 1.0f ? 1.0f - x : 0.0f;
}

struct ResizeArgs {
  const void* x;
  void* y;
  float* tmp;
  AxisAA ax, ay;          // width / height axes
  int ct, cl, ch, cw;     // crop window in resized coordinates (may reach outside the resized image: zero padding)
  int r0, nr;             // input rows the height pass reads: [r0, r0 + nr)
  int c0, nc;             // resized columns the width pass computes: [c0, c0 + nc)
  long long planes;
  int channels;
  int preset;             // 0: same dtype out; 1: float out = ((v [/ 255]) - mean) / std
  float mean[4], stdv[4];
};

template <typename T>
__global__ __launch_bounds__(256) void k_resize_w(const ResizeArgs A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per_plane = (long long)A.nr * A.nc;
  if (idx >= A.planes * per_plane) return;
  const long long plane = idx / per_plane;
  const int rem = (int)(idx - plane * per_plane);
  const int r = rem / A.nc, i = rem - r * A.nc;
  const T* src = static_cast<const T*>(A.x) + ((size_t)plane * A.ay.in + (A.r0 + r)) * A.ax.in;
  float t;
  if (A.ax.identity) {
    t = (float)src[A.c0 + i];
  } else {
    int xmin, xsize;
    float center;
    aa_window(A.ax, A.c0 + i, xmin, xsize, center);
    float total = 0.f;
    for (int j = 0; j < xsize; ++j) total += aa_weight(A.ax, j, xmin, center);
    const bool norm = total != 0.f;
    t = 0.f;
    for (int j = 0; j < xsize; ++j) {
      float wj = aa_weight(A.ax, j, xmin, center);
      if (norm) wj /= total;
      const float s = (float)src[xmin + j];
      t = (j == 0) ? s * wj : fmaf(s, wj, t);
    }
  }
  A.tmp[idx] = t;
}

// Width pass, tiled: a workgroup owns 64 resized columns x kWRows needed rows of one plane.  The 64 windows and their
// normalised weights are computed ONCE (by the first wave, the same arithmetic as aa_window / aa_weight) into LDS
// [tap][column] and reused for every row; the per-thread version above re-derives them -- two double-precision passes per
// tap -- for every output element, which made 4K inputs (31 taps) compute-bound at 4 % of HBM.
constexpr int kWRows = 64;
constexpr int kWMaxTaps = 96;

template <typename T>
__global__ __launch_bounds__(256) void k_resize_w_tiled(const ResizeArgs A) {
  __shared__ float wts[kWMaxTaps * 64];
  __shared__ int xmins[64], xsizes[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col_tiles = (A.nc + 63) / 64, row_tiles = (A.nr + kWRows - 1) / kWRows;
  const int ct = blockIdx.x % col_tiles;
  const int rt = (blockIdx.x / col_tiles) % row_tiles;
  const long long plane = blockIdx.x / ((long long)col_tiles * row_tiles);
  const int i = ct * 64 + lane;  // column within [0, nc)
  if (wave == 0) {
    int xmin = 0, xsize = 0;
    if (i < A.nc) {
      float center;
      aa_window(A.ax, A.c0 + i, xmin, xsize, center);
      float total = 0.f;
      for (int j = 0; j < xsize; ++j) total += aa_weight(A.ax, j, xmin, center);
      const bool norm = total != 0.f;
      for (int j = 0; j < xsize; ++j) {
        float wj = aa_weight(A.ax, j, xmin, center);
        if (norm) wj /= total;
        wts[j * 64 + lane] = wj;
      }
    }
    xmins[lane] = xmin, xsizes[lane] = xsize;
  }
  __syncthreads();
  if (i >= A.nc) return;
  const int xmin = xmins[lane], xsize = xsizes[lane];
  const int r_end = min((rt + 1) * kWRows, A.nr);
  for (int r = rt * kWRows + wave; r < r_end; r += 4) {
    const T* src = static_cast<const T*>(A.x) + ((size_t)plane * A.ay.in + (A.r0 + r)) * A.ax.in + xmin;
    float t = 0.f;
    for (int j = 0; j < xsize; ++j) {
      const float s = (float)src[j];
      const float wj = wts[j * 64 + lane];
      t = (j == 0) ? s * wj : fmaf(s, wj, t);
    }
    A.tmp[((size_t)plane * A.nr + r) * A.nc + i] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_resize_h(const ResizeArgs A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per_plane = (long long)A.ch * A.cw;
  if (idx >= A.planes * per_plane) return;
  const long long plane = idx / per_plane;
  const int rem = (int)(idx - plane * per_plane);
  const int oy = rem / A.cw, ox = rem - oy * A.cw;
  const int ry = oy + A.ct, rx = ox + A.cl;  // position in the resized image
  float t = 0.f;                             // center_crop pads with 0 (transforms/functional.py:587)
  const bool inside = ry >= 0 && ry < A.ay.out && rx >= 0 && rx < A.ax.out;
  if (inside) {
    const float* col = A.tmp + (size_t)plane * A.nr * A.nc + (rx - A.c0);
    if (A.ay.identity) {
      t = col[(size_t)(ry - A.r0) * A.nc];
    } else {
      int ymin, ysize;
      float center;
      aa_window(A.ay, ry, ymin, ysize, center);
      float total = 0.f;
      for (int j = 0; j < ysize; ++j) total += aa_weight(A.ay, j, ymin, center);
      const bool norm = total != 0.f;
      for (int j = 0; j < ysize; ++j) {
        float wj = aa_weight(A.ay, j, ymin, center);
        if (norm) wj /= total;
        const float s = col[(size_t)(ymin - A.r0 + j) * A.nc];
        t = (j == 0) ? s * wj : fmaf(s, wj, t);
      }
    }
  }
  constexpr bool u8 = sizeof(T) == 1;
  if (u8 && inside) {  // torch.round, then .to(uint8) (_functional_tensor.py:540-548)
    t = __builtin_rintf(t);
    t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
  }
  if (A.preset) {
    // convert_image_dtype(float): uint8 -> `.to(float32) / 255.0` (v1, _functional_tensor.py:93-99); float stays.
    // normalize: tensor.sub_(mean).div_(std) (:928)
    if (u8) t = t / 255.0f;
    const int c = (int)(plane % A.channels);
    static_cast<float*>(A.y)[idx] = (t - A.mean[c]) / A.stdv[c];
  } else {
    if (u8)
      static_cast<uint8_t*>(A.y)[idx] = (uint8_t)t;
    else
      static_cast<float*>(A.y)[idx] = t;
  }
}

// ---- both passes in ONE kernel: the fp32 temporary of the width pass never leaves the CU ------------------------------------
// The two kernels above write the width pass of every needed row to a workspace (57 MB for 64 photos of 375 x 500) and read it
// back: 0.15 ms for 61 MB of input + output, 5 % of HBM, two launches.  Here a workgroup owns TH x 64 pixels of one plane's crop
// window: wave 0 computes the 64 column windows + weights, wave 1 the TH row windows + weights (the arithmetic of aa_window /
// aa_weight, as above), then the width pass runs over exactly the input rows these TH output rows read -- into LDS -- and the
// height pass + epilogue run from LDS.  Same operations in the same order as k_resize_w_tiled + k_resize_h: the same bits.
// Rows shared by vertically neighbouring tiles (2 x support + 1 of them) get their width pass twice.
struct ResizeFusedArgs {
  ResizeArgs r;
  int th, tiles_x, tiles_y;
  int ntx, nty, nrmax;  // taps kept per column / row window; rows of the LDS temporary
};

// MAXT: compile-time bound of a column window's taps (8 or 16; 0 = run-time loop).  With the bound the width pass issues the
// RB x MAXT loads of RB rows back to back (tap index clamped into the window, bytes kept raw) and only then runs the chains; the
// run-time loop `for j < xsize: t = fma((float)src[j], w[j], t)` waits a memory round trip per tap (taps past xsize are skipped, so
// the chain is the same).  Which form runs: launch_resize.
template <typename T, int MAXT>
__global__ __launch_bounds__(256) void k_resize_fused(const ResizeFusedArgs F) {
  const ResizeArgs& A = F.r;
  extern __shared__ __attribute__((aligned(16))) float rl[];
  float* const wx = rl;                                  // [ntx][64]
  float* const wy = wx + F.ntx * 64;                     // [nty][th]
  float* const tmp = wy + F.nty * F.th;                  // [nrmax][64]
  int* const xmins = reinterpret_cast<int*>(tmp + F.nrmax * 64);
  int* const xsizes = xmins + 64;
  int* const ymins = xsizes + 64;
  int* const ysizes = ymins + F.th;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned bid = blockIdx.x;
  const int tx = bid % F.tiles_x;
  bid /= F.tiles_x;
  const int ty = bid % F.tiles_y;
  const long long plane = bid / F.tiles_y;
  const int ox0 = tx * 64, oy0 = ty * F.th;            // tile origin in the crop window
  const int rx = ox0 + lane + A.cl;                    // this lane's column in the resized image
  const bool col_in = ox0 + lane < A.cw && rx >= 0 && rx < A.ax.out;
  if (wave == 0) {  // column windows
    int xmin = 0, xsize = 0;
    if (col_in) {
      if (A.ax.identity) {
        xmin = rx, xsize = 1;
        wx[lane] = 1.f;
      } else {
        float center;
        aa_window(A.ax, rx, xmin, xsize, center);
        float total = 0.f;
        for (int j = 0; j < xsize; ++j) total += aa_weight(A.ax, j, xmin, center);
        const bool norm = total != 0.f;
        for (int j = 0; j < xsize; ++j) {
          float wj = aa_weight(A.ax, j, xmin, center);
          if (norm) wj /= total;
          wx[j * 64 + lane] = wj;
        }
      }
    }
    xmins[lane] = xmin, xsizes[lane] = xsize;
  } else if (wave == 1 && lane < F.th) {  // row windows
    const int ry = oy0 + lane + A.ct;
    int ymin = 0, ysize = 0;
    if (oy0 + lane < A.ch && ry >= 0 && ry < A.ay.out) {
      if (A.ay.identity) {
        ymin = ry, ysize = 1;
        wy[lane] = 1.f;
      } else {
        float center;
        aa_window(A.ay, ry, ymin, ysize, center);
        float total = 0.f;
        for (int j = 0; j < ysize; ++j) total += aa_weight(A.ay, j, ymin, center);
        const bool norm = total != 0.f;
        for (int j = 0; j < ysize; ++j) {
          float wj = aa_weight(A.ay, j, ymin, center);
          if (norm) wj /= total;
          wy[j * F.th + lane] = wj;
        }
      }
    }
    ymins[lane] = ymin, ysizes[lane] = ysize;
  }
  __syncthreads();
  // input rows this tile reads: from the first to the last row window that exists
  int r_lo = 0x7fffffff, r_hi = 0;
  for (int i = 0; i < F.th; ++i)
    if (ysizes[i] > 0) r_lo = min(r_lo, ymins[i]), r_hi = max(r_hi, ymins[i] + ysizes[i]);
  const int nr = r_hi > r_lo ? r_hi - r_lo : 0;  // <= nrmax by construction of th
  {  // width pass into LDS: lane = column, the waves share the rows.  (Staging the tile's input region in LDS with 4-element
     // loads first and running this pass from there was tried: 0.104 -> 0.121 ms on 64 photos of 375 x 500.)
    const int xmin = xmins[lane], xsize = xsizes[lane];
    if constexpr (MAXT > 0) {
      constexpr int RB = MAXT <= 8 ? 4 : 2;  // rows in flight per wave
      float wj[MAXT];
#pragma unroll
      for (int j = 0; j < MAXT; ++j) wj[j] = (j < xsize) ? wx[j * 64 + lane] : 0.f;
      const T* const col = static_cast<const T*>(A.x) + ((size_t)plane * A.ay.in + r_lo) * A.ax.in + xmin;
      for (int r = wave; r < nr; r += 4 * RB) {
        T raw[RB][MAXT];
        if (xsize > 0) {
#pragma unroll
          for (int k = 0; k < RB; ++k) {
            const T* src = col + (size_t)min(r + 4 * k, nr - 1) * A.ax.in;  // a row past the tile: a valid row, loaded and dropped
#pragma unroll
            for (int j = 0; j < MAXT; ++j) raw[k][j] = src[min(j, xsize - 1)];
          }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          if (r + 4 * k >= nr) break;
          float t = 0.f;
          if (xsize > 0) {
            t = (float)raw[k][0] * wj[0];  // identity axis: weight 1, the value itself
#pragma unroll
            for (int j = 1; j < MAXT; ++j)
              if (j < xsize) t = fmaf((float)raw[k][j], wj[j], t);
          }
          tmp[(r + 4 * k) * 64 + lane] = t;
        }
      }
    } else {
      for (int r = wave; r < nr; r += 4) {
        float t = 0.f;
        if (xsize > 0) {
          const T* src = static_cast<const T*>(A.x) + ((size_t)plane * A.ay.in + (r_lo + r)) * A.ax.in + xmin;
          if (A.ax.identity) {
            t = (float)src[0];
          } else {
            for (int j = 0; j < xsize; ++j) {
              const float sv = (float)src[j];
              const float wj = wx[j * 64 + lane];
              t = (j == 0) ? sv * wj : fmaf(sv, wj, t);
            }
          }
        }
        tmp[r * 64 + lane] = t;
      }
    }
  }
  __syncthreads();
  constexpr bool u8 = sizeof(T) == 1;
  for (int oyl = wave; oyl < F.th; oyl += 4) {  // height pass + epilogue: lane = column
    const int oy = oy0 + oyl, ox = ox0 + lane;
    if (oy >= A.ch || ox >= A.cw) continue;
    const int ysize = ysizes[oyl];
    const bool inside = ysize > 0 && xsizes[lane] > 0;  // else: center_crop's zero padding
    float t = 0.f;
    if (inside) {
      const float* col = tmp + (ymins[oyl] - r_lo) * 64 + lane;
      if (A.ay.identity) {
        t = col[0];
      } else {
        for (int j = 0; j < ysize; ++j) {
          const float sv = col[j * 64];
          const float wj = wy[j * F.th + oyl];
          t = (j == 0) ? sv * wj : fmaf(sv, wj, t);
        }
      }
    }
    if (u8 && inside) {
      t = __builtin_rintf(t);
      t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
    }
    const size_t idx = ((size_t)plane * A.ch + oy) * A.cw + ox;
    if (A.preset) {
      if (u8) t = t / 255.0f;
      const int c = (int)(plane % A.channels);
      static_cast<float*>(A.y)[idx] = (t - A.mean[c]) / A.stdv[c];
    } else {
      if (u8)
        static_cast<uint8_t*>(A.y)[idx] = (uint8_t)t;
      else
        static_cast<float*>(A.y)[idx] = t;
    }
  }
}

// tile height and LDS size of the fused kernel for a geometry; false: use the two kernels (scale factors above 7: the tile height
// shrinks and the redundant width passes of the halo rows weigh more -- 8 x 4K -> 256: 0.15 ms with the two kernels, 0.20 ms fused)
constexpr int kFusedMaxTaps = 15;
static bool fused_plan(const ResizeArgs& a, ResizeFusedArgs& f, size_t& lds) {
  if ((!a.ax.identity && a.ax.max_interp > kFusedMaxTaps) || (!a.ay.identity && a.ay.max_interp > kFusedMaxTaps)) return false;
  f.ntx = a.ax.identity ? 1 : a.ax.max_interp;
  f.nty = a.ay.identity ? 1 : a.ay.max_interp;
  // input rows a tile of th output rows can read: (th - 1) * scale + the window of one row (+ 2 for the roundings)
  const float sy = a.ay.identity ? 1.f : a.ay.scale;
  int th = 16;
  auto rows_of = [&](int t) { return (int)ceilf((t - 1) * sy) + f.nty + 2; };
  auto bytes_of = [&](int t) { return sizeof(float) * ((size_t)f.ntx * 64 + (size_t)f.nty * t + (size_t)rows_of(t) * 64 + 128 + 2 * (size_t)t); };
  while (th > 1 && bytes_of(th) > 40 * 1024) th /= 2;
  if (bytes_of(th) > 64 * 1024) return false;
  f.th = th, f.nrmax = rows_of(th);
  f.tiles_x = (a.cw + 63) / 64, f.tiles_y = (a.ch + th - 1) / th;
  lds = bytes_of(th);
  return true;
}

// ---------------------------------------------------------------------------------------------
static int fill_geometry(ResizeArgs& a, int64_t planes, int h, int w, int oh, int ow, int ct, int cl, int ch, int cw) {
  a.planes = planes;
  a.ax = make_axis(w, ow);
  a.ay = make_axis(h, oh);
  a.ct = ct, a.cl = cl, a.ch = ch, a.cw = cw;
  // resized rows / columns the crop window really touches
  const int y_lo = ct < 0 ? 0 : ct, y_hi = (ct + ch > oh) ? oh : ct + ch;
  const int x_lo = cl < 0 ? 0 : cl, x_hi = (cl + cw > ow) ? ow : cl + cw;
  if (y_lo >= y_hi || x_lo >= x_hi) {  // the window lies wholly in the padding
    a.r0 = 0, a.nr = 0, a.c0 = 0, a.nc = 0;
    return MV_OK;
  }
  a.c0 = x_lo, a.nc = x_hi - x_lo;
  if (a.ay.identity) {
    a.r0 = y_lo, a.nr = y_hi - y_lo;
  } else {
    int m0, s0, m1, s1;
    float c;
    aa_window(a.ay, y_lo, m0, s0, c);
    aa_window(a.ay, y_hi - 1, m1, s1, c);
    a.r0 = m0, a.nr = m1 + s1 - m0;
    if (a.nr < 1) a.nr = 1;
    if (a.r0 + a.nr > h) a.nr = h - a.r0;
  }
  return MV_OK;
}

int64_t resize_workspace_bytes(int64_t planes, int h, int w, int oh, int ow, int ct, int cl, int ch, int cw) {
  ResizeArgs a = {};
  fill_geometry(a, planes, h, w, oh, ow, ct, cl, ch, cw);
  ResizeFusedArgs f = {};
  size_t lds = 0;
  if (!tune_env("MV_RESIZE_TWO_KERNELS") && fused_plan(a, f, lds)) return 0;  // one kernel, the temporary stays in LDS
  return (int64_t)sizeof(float) * planes * a.nr * a.nc;
}

int launch_resize(const void* x, void* y, bool u8, int64_t planes, int channels, int h, int w, int oh, int ow, int ct,
                  int cl, int ch, int cw, int preset, const float* mean, const float* stdv, void* workspace,
                  int64_t workspace_bytes, hipStream_t s) {
  ResizeArgs a = {};
  fill_geometry(a, planes, h, w, oh, ow, ct, cl, ch, cw);
  a.x = x, a.y = y, a.tmp = static_cast<float*>(workspace);
  a.channels = channels > 0 ? channels : 1;
  a.preset = preset;
  for (int i = 0; i < 4; ++i) a.mean[i] = (preset && i < channels) ? mean[i] : 0.f, a.stdv[i] = (preset && i < channels) ? stdv[i] : 1.f;
  {
    ResizeFusedArgs f = {};
    size_t lds = 0;
    f.r = a;
    if (!tune_env("MV_RESIZE_TWO_KERNELS") && fused_plan(a, f, lds)) {
      const long long blocks = planes * (long long)f.tiles_x * f.tiles_y;
      if (blocks > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "resize: batch too large for one launch");
      if (blocks == 0) return MV_OK;
      // batched loads pay from 9 taps up (24 x 1080p -> 256 / 224: 0.191 -> 0.167 ms); with 5 taps (375 x 500 photos) eight workgroups
      // per CU already hide the per-tap round trips and the batched form's address arithmetic costs more than it saves (0.107 ->
      // 0.122 ms): the run-time loop stays there.  MV_RESIZE_TAPS (tuning build) forces 0 / 8 / 16.
      int maxt = f.ntx <= 8 ? 0 : 16;  // f.ntx <= kFusedMaxTaps = 15
      if (const char* e = tune_env("MV_RESIZE_TAPS")) maxt = atoi(e) >= f.ntx || atoi(e) == 0 ? atoi(e) : maxt;
      auto launch = [&](auto kern) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, f);
      };
      if (u8) {
        if (maxt == 8) launch(k_resize_fused<uint8_t, 8>);
        else if (maxt == 16) launch(k_resize_fused<uint8_t, 16>);
        else launch(k_resize_fused<uint8_t, 0>);
      } else {
        if (maxt == 8) launch(k_resize_fused<float, 8>);
        else if (maxt == 16) launch(k_resize_fused<float, 16>);
        else launch(k_resize_fused<float, 0>);
      }
      return check_launchf("k_resize_fused<%s,th%d>", u8 ? "u8" : "f32", f.th);
    }
  }
  const long long need = (long long)sizeof(float) * planes * a.nr * a.nc;
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return set_error(MV_ERR_INVALID_ARGUMENT, "resize: workspace of %lld bytes needed (mv_resize_workspace_bytes), got %lld", need,
                     (long long)workspace_bytes);
  const long long n1 = planes * (long long)a.nr * a.nc, n2 = planes * (long long)ch * cw;
  if (n1 > 256LL * 0x7fffffffLL || n2 > 256LL * 0x7fffffffLL)
    return set_error(MV_ERR_UNSUPPORTED, "resize: batch too large for one launch");
  if (n1 > 0) {
    const long long tiles = planes * ((a.nc + 63) / 64) * ((a.nr + kWRows - 1) / kWRows);
    if (!a.ax.identity && a.ax.max_interp <= kWMaxTaps && tiles <= 0x7fffffffLL) {
      if (u8)
        hipLaunchKernelGGL((k_resize_w_tiled<uint8_t>), dim3((unsigned)tiles), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((k_resize_w_tiled<float>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    } else {
      const unsigned nb = (unsigned)((n1 + 255) / 256);
      if (u8)
        hipLaunchKernelGGL((k_resize_w<uint8_t>), dim3(nb), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((k_resize_w<float>), dim3(nb), dim3(256), 0, s, a);
    }
    if (int rc = check_launch("k_resize_w")) return rc;
  }
  if (n2 > 0) {
    const unsigned nb = (unsigned)((n2 + 255) / 256);
    if (u8)
      hipLaunchKernelGGL((k_resize_h<uint8_t>), dim3(nb), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_resize_h<float>), dim3(nb), dim3(256), 0, s, a);
    return check_launch("k_resize_h");
  }
  return MV_OK;
}

}  // namespace mv
