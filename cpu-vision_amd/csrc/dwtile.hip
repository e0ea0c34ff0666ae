// dwtile.hip -- LDS-staged depthwise cross-correlation for any odd (ky, kx): the general form of the
// reference primitive pad(border) + conv2d(groups=C) (transforms/v2/functional/_misc.py:153-155,
// _color.py:260, transforms/_functional_tensor.py:759-761).
//
// A 256-thread workgroup produces a 256 x (4*RPT) output tile of one plane:
//   1. the input tile plus its (ky-1, kx-1) halo is staged ONCE in LDS with 16-byte loads and
//      16-byte ds_write_b128; the border rule (reflect-101 / zero) is applied while staging, so no
//      padded frame is ever materialised (the reference's reflection_pad2d is a full extra pass);
//   2. each lane owns 4 adjacent output columns and RPT output rows; it walks down the tile reading
//      every LDS row segment once (ds_read_b128, lanes 16 B apart: conflict-free) and feeding all
//      the output rows that tap it, accumulators in registers;
//   3. taps sit in LDS (filled from by-value kernel arguments, from a device pointer, or as the
//      outer product k1d_y[j]*k1d_x[i] -- bit-identical to _misc.py:97) and, for the templated
//      sizes, are hoisted to registers.
// The fma chain per output is the oracle's (dy outer, dx inner, from +0): bit-identical results.
#include <cstdlib>

#include "mv_common.h"

namespace mv {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
// under-aligned flavours: gfx950 global memory takes 4- and 16-byte accesses at any byte address (tools/micro/unaligned.hip),
// so widths with W % 4 != 0 -- whose rows start anywhere -- still move 4 pixels per instruction
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned char u8x4u __attribute__((ext_vector_type(4), aligned(1)));

constexpr int kTileW = 256;
#ifndef MV_TILE_RPT
#define MV_TILE_RPT 4  // output rows per lane for the templated sizes (tile height = 4 * RPT)
#endif
constexpr int kRptSized = MV_TILE_RPT;

enum { TAPS_BY_VALUE = 0, TAPS_DEVICE = 1, TAPS_OUTER = 2 };

struct TileArgs {
  const void* x;
  void* y;
  const float* w_dev;
  Taps2D w2;
  Taps1D w1;
  int taps_mode;
  int h, w, ky, kx, border;
  int tiles_x, tiles_y;
  unsigned nblocks;
  FramePtrs fp;  // mv_*_v: per-frame base pointers (n == 0: x / y are one contiguous batch)
};

// ---- storage types: fp32, uint8 (round_() + narrow on store), fp16 and bf16 (fp32 arithmetic, one round-to-nearest-even on
//      store -- what `.to(float32)` -> filter -> `.to(dtype)` computes, without the two extra passes over the image)
struct bf16_t {
  unsigned short bits;
};
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4u __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4u __attribute__((ext_vector_type(4), aligned(2)));

__device__ inline float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ inline unsigned short f32_to_bf16(float f) {  // round to nearest even; NaN -> quiet NaN (as ATen's c10::BFloat16)
  unsigned u = __builtin_bit_cast(unsigned, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <typename T>
struct Px;
template <>
struct Px<float> {
  static __device__ inline float ld(const float* p) { return *p; }
  static __device__ inline void st(float* p, float v) { *p = v; }
  template <bool ALIGNED>
  static __device__ inline f4 ld4(const float* p) {
    if (ALIGNED) return *reinterpret_cast<const f4*>(p);
    const f4u q = *reinterpret_cast<const f4u*>(p);
    return (f4){q.x, q.y, q.z, q.w};
  }
  template <bool ALIGNED>
  static __device__ inline void st4(float* p, const float (&v)[4]) {
    if (ALIGNED)
      __builtin_nontemporal_store((f4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(p));
    else
      *reinterpret_cast<f4u*>(p) = (f4u){v[0], v[1], v[2], v[3]};
  }
};
template <>
struct Px<uint8_t> {
  static __device__ inline float ld(const uint8_t* p) { return (float)*p; }
  static __device__ inline void st(uint8_t* p, float v) { *p = round_u8(v); }
  template <bool ALIGNED>
  static __device__ inline f4 ld4(const uint8_t* p) {
    u8x4 b;
    if (ALIGNED) {
      b = *reinterpret_cast<const u8x4*>(p);
    } else {
      const u8x4u q = *reinterpret_cast<const u8x4u*>(p);
      b = (u8x4){q.x, q.y, q.z, q.w};
    }
    return (f4){(float)b.x, (float)b.y, (float)b.z, (float)b.w};
  }
  template <bool ALIGNED>
  static __device__ inline void st4(uint8_t* p, const float (&v)[4]) {
    if (ALIGNED)
      *reinterpret_cast<u8x4*>(p) = (u8x4){round_u8(v[0]), round_u8(v[1]), round_u8(v[2]), round_u8(v[3])};
    else
      *reinterpret_cast<u8x4u*>(p) = (u8x4u){round_u8(v[0]), round_u8(v[1]), round_u8(v[2]), round_u8(v[3])};
  }
};
template <>
struct Px<_Float16> {
  static __device__ inline float ld(const _Float16* p) { return (float)*p; }
  static __device__ inline void st(_Float16* p, float v) { *p = (_Float16)v; }
  template <bool ALIGNED>
  static __device__ inline f4 ld4(const _Float16* p) {
    h4 b;
    if (ALIGNED) {
      b = *reinterpret_cast<const h4*>(p);
    } else {
      const h4u q = *reinterpret_cast<const h4u*>(p);
      b = (h4){q.x, q.y, q.z, q.w};
    }
    return (f4){(float)b.x, (float)b.y, (float)b.z, (float)b.w};
  }
  template <bool ALIGNED>
  static __device__ inline void st4(_Float16* p, const float (&v)[4]) {
    if (ALIGNED)
      *reinterpret_cast<h4*>(p) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    else
      *reinterpret_cast<h4u*>(p) = (h4u){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  }
};
template <>
struct Px<bf16_t> {
  static __device__ inline float ld(const bf16_t* p) { return bf16_to_f32(p->bits); }
  static __device__ inline void st(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
  template <bool ALIGNED>
  static __device__ inline f4 ld4(const bf16_t* p) {
    us4 b;
    if (ALIGNED) {
      b = *reinterpret_cast<const us4*>(p);
    } else {
      const us4u q = *reinterpret_cast<const us4u*>(p);
      b = (us4){q.x, q.y, q.z, q.w};
    }
    return (f4){bf16_to_f32(b.x), bf16_to_f32(b.y), bf16_to_f32(b.z), bf16_to_f32(b.w)};
  }
  template <bool ALIGNED>
  static __device__ inline void st4(bf16_t* p, const float (&v)[4]) {
    const us4 b = {f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
    if (ALIGNED)
      *reinterpret_cast<us4*>(p) = b;
    else
      *reinterpret_cast<us4u*>(p) = (us4u){b.x, b.y, b.z, b.w};
  }
};

// TW = tile width in pixels: TW / 4 lanes cover a tile row, the 256 threads form 1024 / TW row groups of RPT output rows each
// (TW = 256: a wave per row group, the 256 x (4 * RPT) tile of the header; TW = 128 / 64 / 32 for images at most that wide --
// thumbnails -- where a 256-pixel tile would idle most lanes: 128 x 32, 64 x 32 and 32 x 32 tiles).
template <typename T, int KY, int KX, int RPT, bool VEC, int TW = kTileW>
__global__ __launch_bounds__(256) void k_dwtile(const TileArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ky = KY ? KY : A.ky, kx = KX ? KX : A.kx;
  const int ry = ky >> 1, rx = kx >> 1;
  const int L = (rx + 3) & ~3;  // left/right halo rounded up to 4 floats: the body stays 16-B aligned
  const int pitch = L + TW + L;
  constexpr int LPR = TW / 4;     // lanes per tile row
  constexpr int TH = (256 / LPR) * RPT;
  const int rows = TH + ky - 1;
  float* wl = lds + rows * pitch;  // ky*kx taps behind the tile

  const int tid = threadIdx.x;
  const int lane = tid % LPR, wave = tid / LPR;  // column slot and row group (TW = 256: the lane and the wave)
  const unsigned wid = xcd_remap(blockIdx.x, A.nblocks);
  const int tx = wid % A.tiles_x;
  const unsigned t2 = wid / A.tiles_x;
  const int ty = t2 % A.tiles_y;
  const long long plane = t2 / A.tiles_y;
  const int h = A.h, w = A.w, border = A.border;
  const int x0 = tx * TW, y0 = ty * TH;
  const T* xp = frame_in<T>(A.fp, A.x, plane, (size_t)h * w);

  // ---- taps -> LDS
  for (int i = tid; i < ky * kx; i += 256) {
    float v;
    if (A.taps_mode == TAPS_BY_VALUE)
      v = A.w2.w[i];
    else if (A.taps_mode == TAPS_DEVICE)
      v = A.w_dev[i];
    else
      v = A.w1.y[i / kx] * A.w1.x[i % kx];
    wl[i] = v;
  }

  // ---- stage the tile (+halo), border rule applied here
  const int slots = pitch >> 2;
  for (int idx = tid; idx < rows * slots; idx += 256) {
    const int row = idx / slots, slot = idx - row * slots;
    const int gx0 = x0 - L + (slot << 2);
    const int gy = y0 - ry + row;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    int sy = gy;
    bool row_ok = (gy >= 0 && gy < h);
    if (border == MV_BORDER_REFLECT) {
      sy = reflect_clamp(gy, h);
      row_ok = true;
    }
    if (row_ok) {
      const T* rp = xp + (size_t)sy * w;
      if (gx0 >= 0 && gx0 + 3 < w) {
        v = Px<T>::template ld4<VEC>(rp + gx0);
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gx = gx0 + j;
          if (border == MV_BORDER_REFLECT)
            e[j] = Px<T>::ld(rp + reflect_clamp(gx, w));
          else
            e[j] = (gx >= 0 && gx < w) ? Px<T>::ld(rp + gx) : 0.f;
        }
        v.x = e[0], v.y = e[1], v.z = e[2], v.w = e[3];
      }
    }
    *reinterpret_cast<f4*>(lds + row * pitch + (slot << 2)) = v;
  }
  __syncthreads();

  // ---- compute: lane -> 4 columns, wave -> RPT rows
  float acc[RPT][4];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;

  const float* lbase = lds + (wave * RPT) * pitch + (lane << 2);  // column (c0 - L) of this lane

  if constexpr (KY != 0) {
    constexpr int RX = KX / 2;
    constexpr int LL = (RX + 3) & ~3;
    constexpr int NCH = 1 + 2 * (LL / 4);  // 16-byte chunks covering [c0-LL, c0+4+LL)
    float wr[KY * KX];
#pragma unroll
    for (int i = 0; i < KY * KX; ++i) wr[i] = wl[i];
#pragma unroll
    for (int j = 0; j < RPT + KY - 1; ++j) {
      float seg[NCH * 4];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f4 q = *reinterpret_cast<const f4*>(lbase + j * pitch + c * 4);
        seg[c * 4 + 0] = q.x, seg[c * 4 + 1] = q.y, seg[c * 4 + 2] = q.z, seg[c * 4 + 3] = q.w;
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int dy = j - r;
        if (dy >= 0 && dy < KY) {
#pragma unroll
          for (int dx = 0; dx < KX; ++dx)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[r][p] = fmaf(wr[dy * KX + dx], seg[LL - RX + p + dx], acc[r][p]);
        }
      }
    }
  } else {
    const float* lcol = lbase + (L - rx);  // column (c0 - rx)
    for (int j = 0; j < RPT + ky - 1; ++j) {
      const float* lrow = lcol + j * pitch;
      float s0 = lrow[0], s1 = lrow[1], s2 = lrow[2];
      for (int dx = 0; dx < kx; ++dx) {
        const float s3 = lrow[dx + 3];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const int dy = j - r;
          if (dy >= 0 && dy < ky) {
            const float wv = wl[dy * kx + dx];
            acc[r][0] = fmaf(wv, s0, acc[r][0]);
            acc[r][1] = fmaf(wv, s1, acc[r][1]);
            acc[r][2] = fmaf(wv, s2, acc[r][2]);
            acc[r][3] = fmaf(wv, s3, acc[r][3]);
          }
        }
        s0 = s1, s1 = s2, s2 = s3;
      }
    }
  }

  // ---- store
  const int ox = x0 + (lane << 2);
  if (border == MV_BORDER_VALID) {
    const int ow = w - 2 * rx, oh = h - 2 * ry;
    T* yp = frame_out<T>(A.fp, A.y, plane, (size_t)oh * ow);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int oy = y0 + wave * RPT + r;
      if (oy >= ry && oy < h - ry) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int gx = ox + p;
          if (gx >= rx && gx < w - rx) {
            Px<T>::st(yp + (size_t)(oy - ry) * ow + gx - rx, acc[r][p]);
          }
        }
      }
    }
    return;
  }
  T* yp = frame_out<T>(A.fp, A.y, plane, (size_t)h * w);
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int oy = y0 + wave * RPT + r;
    if (oy < h) {
      T* rp = yp + (size_t)oy * w;
      if (VEC) {
        if (ox < w) Px<T>::template st4<true>(rp + ox, acc[r]);
      } else if (ox + 3 < w) {
        Px<T>::template st4<false>(rp + ox, acc[r]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (ox + p < w) Px<T>::st(rp + ox + p, acc[r][p]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int KY, int KX, int RPT, int TW = kTileW>
static int launch_sized(const TileArgs& a, bool vec, size_t lds_bytes, hipStream_t s) {
  dim3 grid(a.nblocks), block(256);
  if (vec) {
    auto k = k_dwtile<T, KY, KX, RPT, true, TW>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, a);
  } else {
    auto k = k_dwtile<T, KY, KX, RPT, false, TW>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, a);
  }
  // KY x KX == 0 x 0: the run-time-size instantiation
  return check_launchf("k_dwtile<%s,%dx%d,rpt%d,%s,tw%d>", sizeof(T) == 4 ? "f32" : (sizeof(T) == 1 ? "u8" : "f16/bf16"), KY, KX,
                       RPT, vec ? "vec16" : "scalar", TW);
}

// TW / RPT: 256 / (4 sized, 8 run-time sizes) for images; 128 / 4, 64 / 2, 32 / 1 (all 32 rows high) for fp32 images at most
// that wide
template <typename T, int TW, int RPTS, int RPTG>
static int launch_tw(TileArgs& a, int64_t planes, bool vec, hipStream_t s) {
  const int ky = a.ky, kx = a.kx;
  const bool sized = (ky == 3 && kx == 3) || (ky == 5 && kx == 5) || (ky == 7 && kx == 7) || (ky == 5 && kx == 3) ||
                     (ky == 3 && kx == 5) || (ky == 9 && kx == 9) || (ky == 11 && kx == 11);
  const int rpt = sized ? RPTS : RPTG;
  const int th = (1024 / TW) * rpt;
  const int L = ((kx / 2) + 3) & ~3;
  const int pitch = L + TW + L;
  const size_t lds_bytes = ((size_t)(th + ky - 1) * pitch + (size_t)ky * kx) * sizeof(float);
  if (lds_bytes > 160 * 1024) return set_error(MV_ERR_UNSUPPORTED, "dwtile: %dx%d taps need %zu B of LDS", ky, kx, lds_bytes);
  a.tiles_x = (a.w + TW - 1) / TW;
  a.tiles_y = (a.h + th - 1) / th;
  const long long nb = (long long)planes * a.tiles_x * a.tiles_y;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "dwtile: batch too large for one launch");
  a.nblocks = (unsigned)nb;
  if (ky == 3 && kx == 3) return launch_sized<T, 3, 3, RPTS, TW>(a, vec, lds_bytes, s);
  if (ky == 5 && kx == 5) return launch_sized<T, 5, 5, RPTS, TW>(a, vec, lds_bytes, s);
  if (ky == 7 && kx == 7) return launch_sized<T, 7, 7, RPTS, TW>(a, vec, lds_bytes, s);
  if (ky == 5 && kx == 3) return launch_sized<T, 5, 3, RPTS, TW>(a, vec, lds_bytes, s);
  if (ky == 3 && kx == 5) return launch_sized<T, 3, 5, RPTS, TW>(a, vec, lds_bytes, s);
  // box / user filters: the run-time-size path reads every tap from LDS and manages 10 Tfma/s
  if (ky == 9 && kx == 9) return launch_sized<T, 9, 9, RPTS, TW>(a, vec, lds_bytes, s);
  if (ky == 11 && kx == 11) return launch_sized<T, 11, 11, RPTS, TW>(a, vec, lds_bytes, s);
  return launch_sized<T, 0, 0, RPTG, TW>(a, vec, lds_bytes, s);
}

template <typename T>
static int launch_typed(TileArgs& a, int64_t planes, bool vec, hipStream_t s) {
  if constexpr (sizeof(T) == 4) {  // fp32 thumbnails (uint8 ones have the 16-pixel kernels, which pack strips the same way)
    if (a.w <= 32) return launch_tw<T, 32, 1, 1>(a, planes, vec, s);
    if (a.w <= 64) return launch_tw<T, 64, 2, 2>(a, planes, vec, s);
    if (a.w <= 128) return launch_tw<T, 128, 4, 4>(a, planes, vec, s);
  }
  if constexpr (sizeof(T) == 4) {
    // tuning knob (tools/perf_single_frame.py): tile height of the templated sizes, 8 / 16 (default) / 32 rows
    if (const char* e = tune_env("MV_TILE_RPT_SIZED")) {
      if (atoi(e) == 2) return launch_tw<T, kTileW, 2, 8>(a, planes, vec, s);
      if (atoi(e) == 8) return launch_tw<T, kTileW, 8, 8>(a, planes, vec, s);
      if (atoi(e) == 4) return launch_tw<T, kTileW, 4, 8>(a, planes, vec, s);
    }
    // one or a few frames per launch: 8-row tiles double the workgroups in flight (single 720p frame 8.8 -> 7.3 us, single
    // 4K frame 42.1 -> 40.5 us, 1080p unchanged at 13.6; profiles/r01_perf_single_frame.log); batches keep 16 rows
    if (a.ky == 3 && a.kx == 3 && planes * ((a.w + kTileW - 1) / kTileW) * ((a.h + 15) / 16) < 8192)
      return launch_tw<T, kTileW, 2, 8>(a, planes, vec, s);
  }
  return launch_tw<T, kTileW, kRptSized, 8>(a, planes, vec, s);
}

int launch_dwtile(const void* x, void* y, int dtype, const float* w2d_host, const float* w_dev, const float* k1d_x,
                  const float* k1d_y, int64_t planes, int h, int w, int ky, int kx, int border, hipStream_t s) {
  TileArgs a = {};
  a.x = x, a.y = y, a.w_dev = w_dev;
  fill_frames(a.fp);
  a.h = h, a.w = w, a.ky = ky, a.kx = kx, a.border = border;
  if (w2d_host) {
    a.taps_mode = TAPS_BY_VALUE;
    for (int i = 0; i < ky * kx; ++i) a.w2.w[i] = w2d_host[i];
  } else if (w_dev) {
    a.taps_mode = TAPS_DEVICE;
  } else {
    a.taps_mode = TAPS_OUTER;
    for (int i = 0; i < kx; ++i) a.w1.x[i] = k1d_x[i];
    for (int i = 0; i < ky; ++i) a.w1.y[i] = k1d_y[i];
  }
  const size_t al = dtype == kDtU8 ? 4 : (dtype == kDtF32 ? 16 : 8);  // 4 pixels per access
  const bool vec = (w % 4 == 0) && ((uintptr_t)x % al == 0) && ((uintptr_t)y % al == 0);
  switch (dtype) {
    case kDtU8: return launch_typed<uint8_t>(a, planes, vec, s);
    case kDtF16: return launch_typed<_Float16>(a, planes, vec, s);
    case kDtBF16: return launch_typed<bf16_t>(a, planes, vec, s);
    default: return launch_typed<float>(a, planes, vec, s);
  }
}

}  // namespace mv
