// dw3x3.hip -- the register-window 3x3 depthwise family for gfx950: the engine of the fused 3x3 operators (Sobel
// pair, adjust_sharpness) and the A/B partner of the LDS-halo-tile kernel (dwtile.hip), which is the default for the
// plain 3x3 filter (same speed +-4 %, but 1.000x instead of 1.13x algorithmic read traffic; MV_FORCE_REG3X3=1 flips).
//
// Replaces the reference's pad(reflect)+conv2d(groups=C) pair for 3x3 kernels
// (transforms/v2/functional/_misc.py:153-155), the valid 3x3 smoothing + blend of
// adjust_sharpness_image (_color.py:253-275; v1 _functional_tensor.py:809-838) and the Sobel pair.
//
// Design (HBM-bound: 8 B of traffic per element, 9 fma):
//   * one wave owns a 256-pixel-wide column segment (64 lanes x 4 pixels: one 16-byte load and one
//     16-byte store per lane and row, 1 KiB per wave instruction) and walks down a strip of R rows;
//   * the 3-row window lives in registers; horizontal neighbours come from the adjacent lanes by
//     wave shuffles; only lane 0 / lane 63 fetch a halo pixel (one predicated dword load per row,
//     served by L1/L2: the neighbouring wave of the same workgroup streams that line anyway);
//   * the border (reflect / zero) is resolved in registers: no padded copy of the frame ever
//     exists (the reference's reflection_pad2d costs a full extra read+write of the frame);
//   * loads run G rows ahead of the arithmetic (explicit double buffer) so that every wave keeps
//     several KiB in flight; rows shared by vertically adjacent strips are re-read through L2,
//     and the block->work map hands each XCD a contiguous run of strips (mv::xcd_remap);
//   * taps arrive as kernel arguments (SGPRs); the fma chain order is the oracle's (row-major from
//     +0), so results are bit-identical to oracle/oracle.c.
#include "mv_common.h"

namespace mv {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));

enum { EPI_STORE = 0, EPI_SOBEL = 1, EPI_SHARP_V2 = 2, EPI_SHARP_V1 = 3 };

struct Dw3x3Args {
  const void* x;
  void* y0;
  void* y1;
  float wa[9];
  float wb[9];
  float alpha;  // sharpness: (float)(1 - f)
  float ratio;  // sharpness v1: (float)f
  float bound;  // clamp upper bound
  int round_blur;  // float storage carrying integer pixels: round the blurred value like the integer path
  int h, w;
  int rows;        // rows per strip (R)
  int strips;      // ceil(h / R)
  int col_segs;    // ceil(w / 256)
  unsigned nblocks;
  long long nitems;  // waves: ceil(planes * strips / strips-per-wave) * col_segs
  long long nitems_units;  // planes * strips
  int lpr;  // lanes per image row (power of two <= 64): images up to 128 pixels wide put 64 / lpr strips in a wave
  FramePtrs fp;  // mv_*_v: per-frame base pointers for x / y0 (n == 0: contiguous batch; never with a second output)
};

// Tuning knobs (compile-time; tools/tune_dw3x3.py builds variants with -D and A/Bs them in one process).
#ifndef MV_DW3X3_GROUP
#define MV_DW3X3_GROUP 4  // rows prefetched ahead of the arithmetic
#endif
#ifndef MV_DW3X3_NT_STORE
#define MV_DW3X3_NT_STORE 1  // outputs are never re-read: keep them out of L2 so halo rows stay
#endif
#ifndef MV_DW3X3_NT_LOAD
#define MV_DW3X3_NT_LOAD 0
#endif
#ifndef MV_DW3X3_XCD
#define MV_DW3X3_XCD 1  // XCD-contiguous work map
#endif
// Ablation switches for profiling builds only (results are WRONG when set): tools/tune_dw3x3.py.
#ifndef MV_ABLATE_HALO
#define MV_ABLATE_HALO 0
#endif
#ifndef MV_ABLATE_SHFL
#define MV_ABLATE_SHFL 0
#endif
#ifndef MV_ABLATE_MATH
#define MV_ABLATE_MATH 0
#endif
constexpr int kGroup = MV_DW3X3_GROUP;

struct Row {
  float l, a, b, c, d, r;  // columns x-1, x .. x+3, x+4
};
struct Raw {
  float a, b, c, d, h;  // the lane's 4 pixels and (lanes 0 / 63 only) its halo pixel
};

template <typename T>
__device__ inline float ldf(const T* p) {
  return (float)*p;
}

// One row of raw loads for this lane.  `rowp` = first pixel of the (already border-mapped) row,
// or nullptr for an all-zero row.  VEC: w % 4 == 0 and 16-byte (u8: 4-byte) aligned rows.
// under-aligned vector types: gfx950 global memory takes 4- and 16-byte accesses at any byte address
// (tools/micro/unaligned.hip), so rows of odd-width images still move 4 pixels per instruction
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned char u8x4u __attribute__((ext_vector_type(4), aligned(1)));

template <typename T, bool VEC>
__device__ inline Raw load_raw(const T* rowp, int xs, int w, int lane, int lpr = kWave) {  // lane = lane within its row
  Raw q = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (rowp == nullptr) return q;
  if (VEC) {
    if (xs < w) {
      if constexpr (sizeof(T) == 4) {
#if MV_DW3X3_NT_LOAD
        f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(rowp + xs));
#else
        f4 v = *reinterpret_cast<const f4*>(rowp + xs);
#endif
        q.a = v.x, q.b = v.y, q.c = v.z, q.d = v.w;
      } else {
        u8x4 v = *reinterpret_cast<const u8x4*>(rowp + xs);
        q.a = (float)v.x, q.b = (float)v.y, q.c = (float)v.z, q.d = (float)v.w;
      }
    }
  } else if (xs + 3 < w) {
    if constexpr (sizeof(T) == 4) {
      const f4u v = *reinterpret_cast<const f4u*>(rowp + xs);
      q.a = v.x, q.b = v.y, q.c = v.z, q.d = v.w;
    } else {
      const u8x4u v = *reinterpret_cast<const u8x4u*>(rowp + xs);
      q.a = (float)v.x, q.b = (float)v.y, q.c = (float)v.z, q.d = (float)v.w;
    }
  } else {
    if (xs + 0 < w) q.a = ldf(rowp + xs + 0);
    if (xs + 1 < w) q.b = ldf(rowp + xs + 1);
    if (xs + 2 < w) q.c = ldf(rowp + xs + 2);
  }
  int hx = (lane == 0) ? xs - 1 : xs + 4;
  bool hl = (lane == 0 && xs > 0) || (lane == lpr - 1 && xs + 4 < w);
#if !MV_ABLATE_HALO
  if (hl) q.h = ldf(rowp + hx);
#endif
  return q;
}

// Complete the 6-wide window: neighbours from adjacent lanes, halo / border at the segment ends.
template <int BORDER>
__device__ inline Row finalize(const Raw& q, int xs, int w, int lane, int lpr = kWave) {
  Row r;
  r.a = q.a, r.b = q.b, r.c = q.c, r.d = q.d;
#if MV_ABLATE_SHFL
  float up = q.d, dn = q.a;
#else
  float up = __shfl_up(q.d, 1);    // lane-1's x+3  -> my x-1
  float dn = __shfl_down(q.a, 1);  // lane+1's x    -> my x+4
#endif
  r.l = (lane == 0) ? q.h : up;
  r.r = (lane == lpr - 1) ? q.h : dn;
  if (BORDER == MV_BORDER_REFLECT) {
    if (xs == 0) r.l = r.b;  // column -1 -> column 1
    int rem = w - xs;        // column w -> column w-2
    if (rem == 4) r.r = r.c;
    if (rem == 3) r.d = r.b;
    if (rem == 2) r.c = r.a;
    if (rem == 1) r.b = r.l;
  } else {
    if (xs == 0) r.l = 0.f;
    if (w - xs == 4) r.r = 0.f;  // (elements past w were loaded as 0 already)
  }
  return r;
}

__device__ inline float tap9(const float (&w)[9], float p0, float p1, float p2, float p3, float p4, float p5,
                             float p6, float p7, float p8) {
  float acc = fmaf(w[0], p0, 0.f);
  acc = fmaf(w[1], p1, acc);
  acc = fmaf(w[2], p2, acc);
  acc = fmaf(w[3], p3, acc);
  acc = fmaf(w[4], p4, acc);
  acc = fmaf(w[5], p5, acc);
  acc = fmaf(w[6], p6, acc);
  acc = fmaf(w[7], p7, acc);
  acc = fmaf(w[8], p8, acc);
  return acc;
}

__device__ inline void conv_row(const float (&w)[9], const Row& t, const Row& m, const Row& b, float (&o)[4]) {
#if MV_ABLATE_MATH
  o[0] = m.a + t.l + b.r, o[1] = m.b, o[2] = m.c, o[3] = m.d + w[0];
  return;
#endif
  o[0] = tap9(w, t.l, t.a, t.b, m.l, m.a, m.b, b.l, b.a, b.b);
  o[1] = tap9(w, t.a, t.b, t.c, m.a, m.b, m.c, b.a, b.b, b.c);
  o[2] = tap9(w, t.b, t.c, t.d, m.b, m.c, m.d, b.b, b.c, b.d);
  o[3] = tap9(w, t.c, t.d, t.r, m.c, m.d, m.r, b.c, b.d, b.r);
}

__device__ inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T, bool VEC>
__device__ inline void store4(T* rowp, int xs, int w, const float (&o)[4]) {
  if (VEC) {
    if (xs < w) {
      if constexpr (sizeof(T) == 4) {
        f4 v = {o[0], o[1], o[2], o[3]};
#if MV_DW3X3_NT_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(rowp + xs));
#else
        *reinterpret_cast<f4*>(rowp + xs) = v;
#endif
      } else {
        u8x4 v = {(unsigned char)(int)o[0], (unsigned char)(int)o[1], (unsigned char)(int)o[2],
                  (unsigned char)(int)o[3]};
        *reinterpret_cast<u8x4*>(rowp + xs) = v;
      }
    }
  } else if (xs + 3 < w) {
    if constexpr (sizeof(T) == 4)
      *reinterpret_cast<f4u*>(rowp + xs) = (f4u){o[0], o[1], o[2], o[3]};
    else
      *reinterpret_cast<u8x4u*>(rowp + xs) =
          (u8x4u){(unsigned char)(int)o[0], (unsigned char)(int)o[1], (unsigned char)(int)o[2], (unsigned char)(int)o[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (xs + j < w) {
        if constexpr (sizeof(T) == 4)
          rowp[xs + j] = o[j];
        else
          rowp[xs + j] = (unsigned char)(int)o[j];
      }
  }
}

// MULTI: several strips per wave (images up to 128 pixels wide: lanes beyond the row would idle otherwise); when false the
// strip -- and every row address -- is wave-uniform
template <typename T, int BORDER, int EPI, bool VEC, bool MULTI>
__global__ __launch_bounds__(256) void k_dw3x3(const Dw3x3Args A) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
#if MV_DW3X3_XCD
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
#else
  const long long item = (long long)blockIdx.x * 4 + wave;
#endif
  if (item >= A.nitems) return;  // whole wave leaves: no barriers in this kernel
  const int seg = (int)(item % A.col_segs);
  const long long t = item / A.col_segs;
  const int lpr = MULTI ? A.lpr : kWave, lir = MULTI ? (lane & (A.lpr - 1)) : lane;  // lanes per row, lane in row
  // MULTI: the wave's 64 / lpr lane groups take consecutive (plane, strip) units, across plane boundaries -- a 32 x 32
  // thumbnail has 4 strips, and 8 groups want work
  const long long units = A.nitems_units;
  const long long unit_raw = MULTI ? t * (kWave / A.lpr) + lane / A.lpr : t;
  const bool strip_ok = MULTI ? unit_raw < units : true;
  const long long unit = strip_ok ? unit_raw : units - 1;
  const int strip = (int)(unit % A.strips);
  const long long plane = unit / A.strips;

  const int h = A.h, w = A.w;
  const int xs = strip_ok ? seg * 256 + lir * 4 : w;  // a group without a strip owns no pixels
  const int y_begin = strip * A.rows;
  const int y_end = min(y_begin + A.rows, h);  // exclusive
  const int y_loop_end = y_begin + A.rows;     // uniform trip count over the wave's groups; stores are guarded by y_end

  const size_t plane_off = (size_t)plane * h * w;
  const T* xp = frame_in<T>(A.fp, A.x, plane, (size_t)h * w);
  T* y0p = frame_out<T>(A.fp, A.y0, plane, (size_t)h * w);
  T* y1p = (EPI == EPI_SOBEL) ? static_cast<T*>(A.y1) + plane_off : nullptr;

  // border-mapped row pointer (nullptr = zero row); rows past the last one this strip needs are
  // never fetched
  auto row_ptr = [&](int y) -> const T* {
    if (y > y_end || !strip_ok) return nullptr;
    if (BORDER == MV_BORDER_REFLECT) return xp + (size_t)reflect_clamp(y, h) * w;
    return (y >= 0 && y < h) ? xp + (size_t)y * w : nullptr;
  };

  Row top = finalize<BORDER>(load_raw<T, VEC>(row_ptr(y_begin - 1), xs, w, lir, lpr), xs, w, lir, lpr);
  Row mid = finalize<BORDER>(load_raw<T, VEC>(row_ptr(y_begin), xs, w, lir, lpr), xs, w, lir, lpr);
  Raw nxt[kGroup];
#pragma unroll
  for (int g = 0; g < kGroup; ++g) nxt[g] = load_raw<T, VEC>(row_ptr(y_begin + 1 + g), xs, w, lir, lpr);

  for (int y = y_begin; y < y_loop_end; y += kGroup) {
    Raw cur[kGroup];
#pragma unroll
    for (int g = 0; g < kGroup; ++g) cur[g] = nxt[g];
    if (y + kGroup < y_loop_end) {
#pragma unroll
      for (int g = 0; g < kGroup; ++g) nxt[g] = load_raw<T, VEC>(row_ptr(y + kGroup + 1 + g), xs, w, lir, lpr);
    }
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      // shuffles run for every lane of the wave (uniform control flow), stores are predicated
      Row bot = finalize<BORDER>(cur[g], xs, w, lir, lpr);
      const int yy = y + g;
      if (yy < y_end) {
        float o[4];
        conv_row(A.wa, top, mid, bot, o);
        if (EPI == EPI_STORE) {
          if constexpr (sizeof(T) == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __builtin_rintf(o[j]);
          }
          store4<T, VEC>(y0p + (size_t)yy * w, xs, w, o);
        } else if (EPI == EPI_SOBEL) {
          store4<T, VEC>(y0p + (size_t)yy * w, xs, w, o);
          float o2[4];
          conv_row(A.wb, top, mid, bot, o2);
          store4<T, VEC>(y1p + (size_t)yy * w, xs, w, o2);
        } else {
          const float xc[4] = {mid.a, mid.b, mid.c, mid.d};
          const bool row_interior = (yy >= 1 && yy < h - 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gx = xs + j;
            const bool interior = row_interior && gx >= 1 && gx < w - 1;
            float blur = o[j];
            if (sizeof(T) == 1 || A.round_blur) blur = __builtin_rintf(blur);
            float res;
            if (EPI == EPI_SHARP_V2) {
              // _color.py:270  view.add_(blurred.sub_(view), alpha=1-f): ATen's add is one fma
              res = interior ? fmaf(A.alpha, blur - xc[j], xc[j]) : xc[j];
            } else {
              // _functional_tensor.py:258-261  ratio*img1 + (1-ratio)*img2, two products then a sum
              const float deg = interior ? blur : xc[j];
              const float t1 = A.ratio * xc[j];
              const float t2 = A.alpha * deg;
              res = t1 + t2;
            }
            o[j] = clampf(res, 0.f, A.bound);
          }
          store4<T, VEC>(y0p + (size_t)yy * w, xs, w, o);
        }
      }
      top = mid;
      mid = bot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* v = tune_env(name);
  return (v && *v) ? atoi(v) : dflt;
}

static void plan(Dw3x3Args& a, int64_t planes, int h, int w) {
  a.h = h, a.w = w;
  a.col_segs = (w + 255) / 256;
  // Strip height.  Measured on MI355X (tools/tune_dw3x3.py, 64 x 4K frames, interleaved rounds): 8 rows per
  // strip (two prefetch groups) is the fastest -- 6.15 TB/s vs 5.6 at 16 rows and 5.1 at 64 -- because many
  // short waves keep the concurrently running blocks on adjacent addresses (whole row bands), and the two halo
  // rows a strip shares with its neighbours are then served by L2 instead of HBM.
  int rows = env_int("MV_DW3X3_ROWS", 0);
  if (rows <= 0) rows = 2 * kGroup;
  if (rows > h) rows = h;
  rows = ((rows + kGroup - 1) / kGroup) * kGroup;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  a.lpr = kWave;
  while (a.lpr > 1 && (a.lpr / 2) * 4 >= w) a.lpr /= 2;
  const int groups = kWave / a.lpr;  // strips per wave
  a.nitems_units = (long long)planes * a.strips;
  a.nitems = ((a.nitems_units + groups - 1) / groups) * a.col_segs;  // groups > 1 only when col_segs == 1
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
}

template <typename T, int BORDER, int EPI>
static int launch_t(const Dw3x3Args& a, bool vec, hipStream_t s) {
  dim3 grid(a.nblocks), block(256);
  const bool multi = a.lpr < kWave;
  if (vec) {
    if (multi)
      hipLaunchKernelGGL((k_dw3x3<T, BORDER, EPI, true, true>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((k_dw3x3<T, BORDER, EPI, true, false>), grid, block, 0, s, a);
  } else {
    if (multi)
      hipLaunchKernelGGL((k_dw3x3<T, BORDER, EPI, false, true>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((k_dw3x3<T, BORDER, EPI, false, false>), grid, block, 0, s, a);
  }
  return check_launch("k_dw3x3");
}

static bool too_many_blocks(const Dw3x3Args& a) { return a.nitems > 4LL * 0x7fffffffLL; }

int launch_dw3x3_f32(const float* x, float* y0, float* y1, const float* w9a, const float* w9b, int64_t planes, int h,
                     int w, int border, hipStream_t s) {
  Dw3x3Args a = {};
  a.x = x, a.y0 = y0, a.y1 = y1;
  if (!y1) fill_frames(a.fp);
  for (int i = 0; i < 9; ++i) a.wa[i] = w9a[i], a.wb[i] = w9b ? w9b[i] : 0.f;
  plan(a, planes, h, w);
  if (too_many_blocks(a)) return set_error(MV_ERR_UNSUPPORTED, "dw3x3: batch too large for one launch");
  const bool vec = (w % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y0 % 16 == 0) &&
                   (y1 == nullptr || (uintptr_t)y1 % 16 == 0);
  const bool sob = (y1 != nullptr);
  if (border == MV_BORDER_REFLECT)
    return sob ? launch_t<float, MV_BORDER_REFLECT, EPI_SOBEL>(a, vec, s)
               : launch_t<float, MV_BORDER_REFLECT, EPI_STORE>(a, vec, s);
  if (border == MV_BORDER_ZERO)
    return sob ? launch_t<float, MV_BORDER_ZERO, EPI_SOBEL>(a, vec, s)
               : launch_t<float, MV_BORDER_ZERO, EPI_STORE>(a, vec, s);
  return set_error(MV_ERR_INVALID_ARGUMENT, "dw3x3: border %d not handled here", border);
}

int launch_dw3x3_u8(const uint8_t* x, uint8_t* y, const float* w9, int64_t planes, int h, int w, int border,
                    hipStream_t s) {
  Dw3x3Args a = {};
  a.x = x, a.y0 = y, a.y1 = nullptr;
  fill_frames(a.fp);
  for (int i = 0; i < 9; ++i) a.wa[i] = w9[i];
  plan(a, planes, h, w);
  if (too_many_blocks(a)) return set_error(MV_ERR_UNSUPPORTED, "dw3x3: batch too large for one launch");
  const bool vec = (w % 4 == 0) && ((uintptr_t)x % 4 == 0) && ((uintptr_t)y % 4 == 0);
  if (border == MV_BORDER_REFLECT) return launch_t<uint8_t, MV_BORDER_REFLECT, EPI_STORE>(a, vec, s);
  if (border == MV_BORDER_ZERO) return launch_t<uint8_t, MV_BORDER_ZERO, EPI_STORE>(a, vec, s);
  return set_error(MV_ERR_INVALID_ARGUMENT, "dw3x3: border %d not handled here", border);
}

int launch_sharpness(const void* x, void* y, bool u8, int64_t planes, int h, int w, double factor, int v1,
                     float bound, int round_blur, hipStream_t s) {
  Dw3x3Args a = {};
  a.x = x, a.y0 = y, a.y1 = nullptr;
  fill_frames(a.fp);
  const float ta = (float)(1.0 / 13.0), tb = (float)(5.0 / 13.0);  // _color.py:253-256
  for (int i = 0; i < 9; ++i) a.wa[i] = (i == 4) ? tb : ta;
  a.alpha = (float)(1.0 - factor);  // the Python double (1 - f), narrowed by ATen
  a.ratio = (float)factor;
  a.bound = u8 ? 255.f : bound;
  a.round_blur = round_blur;
  plan(a, planes, h, w);
  if (too_many_blocks(a)) return set_error(MV_ERR_UNSUPPORTED, "sharpness: batch too large for one launch");
  if (u8) {
    const bool vec = (w % 4 == 0) && ((uintptr_t)x % 4 == 0) && ((uintptr_t)y % 4 == 0);
    return v1 ? launch_t<uint8_t, MV_BORDER_ZERO, EPI_SHARP_V1>(a, vec, s)
              : launch_t<uint8_t, MV_BORDER_ZERO, EPI_SHARP_V2>(a, vec, s);
  }
  const bool vec = (w % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  return v1 ? launch_t<float, MV_BORDER_ZERO, EPI_SHARP_V1>(a, vec, s)
            : launch_t<float, MV_BORDER_ZERO, EPI_SHARP_V2>(a, vec, s);
}

}  // namespace mv
