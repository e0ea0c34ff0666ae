// dwk_u8.hip -- uint8 KY x KX depthwise filter (KY, KX in {3, 5, 7}, not both 3) with 16 pixels per lane (gfx950).
//
// gaussian_blur_image on uint8 with a 5x5 / 7x7 / (3,5) / ... kernel (transforms/v2/functional/_misc.py:139-161:
// .to(float32) -> pad(reflect) + conv2d(groups=C) with the 2-D outer-product kernel -> round_() -> .to(uint8)).
// The LDS-tile kernel (dwtile.hip) moves 4 pixels = 4 BYTES per lane and is instruction-issue-bound at fp32's pixel
// rate (20 % of HBM peak at 2 B/pixel).  Here, like dw3x3_u8.hip, a lane owns 16 pixels (one 16-byte load and one
// 16-byte store per row) and a wave streams a 1024-pixel segment down a strip of rows:
//   * each raw row is unpacked once into a (16 + KX - 1)-wide fp32 window (neighbour lanes' edge dwords by shuffle,
//     reflect-101 / zero border patched in the window);
//   * the KY kernel rows are applied to it at once: stage i of a systolic chain holds the partial sum of the output
//     row that has already received kernel rows 0..i-1, `acc[i] = chain(acc[i-1], kernel row i, window)`, updated in
//     place from the last stage down.  Every output therefore accumulates its KY*KX taps in row-major order from +0
//     -- oracle/oracle.c's order, bit for bit -- and no input row is unpacked twice or kept as fp32.
// Requires W % 16 == 0 and 16-byte aligned planes; anything else takes the 4-pixel tile kernel.
#include <cstdlib>
#include <utility>

#include "mv_common.h"

namespace mv {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct DwkU8Args {
  const uint8_t* x;
  uint8_t* y;
  float w[49];  // row-major KY x KX
  int h, wdt;
  int rows, strips, col_segs;  // col_segs = ceil(w / 1024)
  unsigned nblocks;
  long long nitems;
};

constexpr int kDwkPF = 4;  // raw rows in flight per wave

struct RawRow {
  u32x4 v;        // 16 pixels
  unsigned halo;  // lanes 0 / 63: the 4 bytes left of / right of the segment
};

__device__ inline RawRow dwk_load(const uint8_t* rowp, int xs, int w, int lane) {
  RawRow q;
  q.v = (u32x4){0u, 0u, 0u, 0u};
  q.halo = 0u;
  if (rowp == nullptr) return q;
  if (xs < w) q.v = *reinterpret_cast<const u32x4*>(rowp + xs);
  const int hx = (lane == 0) ? xs - 4 : xs + 16;
  const bool hl = (lane == 0 && xs > 0) || (lane == kWave - 1 && xs + 16 < w);
  if (hl) q.halo = *reinterpret_cast<const unsigned*>(rowp + hx);
  return q;
}

__device__ inline float dwk_ub(unsigned word, int byte) { return (float)((word >> (8 * byte)) & 0xffu); }

// fp32 window of columns xs-RX .. xs+15+RX
template <int RX, int BORDER>
__device__ inline void dwk_window(const RawRow& q, int xs, int w, int lane, float (&win)[16 + 2 * RX]) {
  const unsigned wd[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
#pragma unroll
  for (int i = 0; i < 16; ++i) win[RX + i] = dwk_ub(wd[i >> 2], i & 3);
  unsigned up = __shfl_up(wd[3], 1);    // lane-1's last dword: bytes 3, 2, 1 are columns xs-1, xs-2, xs-3
  unsigned dn = __shfl_down(wd[0], 1);  // lane+1's first dword: bytes 0, 1, 2 are columns xs+16, +17, +18
  if (lane == 0) up = q.halo;
  if (lane == kWave - 1) dn = q.halo;
#pragma unroll
  for (int i = 0; i < RX; ++i) {
    win[RX - 1 - i] = dwk_ub(up, 3 - i);
    win[RX + 16 + i] = dwk_ub(dn, i);
  }
  if (xs == 0) {  // columns -1-i: reflect-101 -> column 1+i
#pragma unroll
    for (int i = 0; i < RX; ++i) win[RX - 1 - i] = (BORDER == MV_BORDER_REFLECT) ? win[RX + 1 + i] : 0.f;
  }
  if (w - xs == 16) {  // columns w+i: reflect-101 -> column w-2-i
#pragma unroll
    for (int i = 0; i < RX; ++i) win[RX + 16 + i] = (BORDER == MV_BORDER_REFLECT) ? win[RX + 14 - i] : 0.f;
  }
}

template <typename F, int... I>
__device__ inline void dwk_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ inline void dwk_static_for(F&& f) {
  dwk_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int KY, int KX, int BORDER>
__global__ __launch_bounds__(256) void k_dwk_u8(const DwkU8Args A) {
  constexpr int RY = KY / 2, RX = KX / 2;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
  if (item >= A.nitems) return;
  const int seg = (int)(item % A.col_segs);
  const long long t2 = item / A.col_segs;
  const int strip = (int)(t2 % A.strips);
  const long long plane = t2 / A.strips;
  const int h = A.h, w = A.wdt;
  const int xs = seg * 1024 + lane * 16;
  const int y0 = strip * A.rows, y1 = min(y0 + A.rows, h);
  const size_t poff = (size_t)plane * h * w;
  const uint8_t* xp = A.x + poff;
  uint8_t* yp = A.y + poff;
  const int t_first = y0 - RY, t_last = y1 - 1 + RY;

  auto row_ptr = [&](int t) -> const uint8_t* {
    if (t > t_last) return nullptr;
    if (BORDER == MV_BORDER_REFLECT) return xp + (size_t)reflect_clamp(t, h) * w;
    return (t >= 0 && t < h) ? xp + (size_t)t * w : nullptr;
  };

  float acc[KY - 1][16];
#pragma unroll
  for (int i = 0; i < KY - 1; ++i)
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[i][p] = 0.f;

  RawRow ring[kDwkPF];
  dwk_static_for<kDwkPF>([&](auto r) { ring[decltype(r)::value] = dwk_load(row_ptr(t_first + decltype(r)::value), xs, w, lane); });

  auto row_step = [&](const int t, auto slot) {
    constexpr int sl = decltype(slot)::value;
    const RawRow raw = ring[sl];
    ring[sl] = dwk_load(row_ptr(t + kDwkPF), xs, w, lane);
    float win[16 + 2 * RX];
    dwk_window<RX, BORDER>(raw, xs, w, lane, win);  // shuffles run for every lane (uniform control flow)
    // last stage first: output row t - RY receives kernel row KY-1
    unsigned out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      float a = acc[KY - 2][p];
#pragma unroll
      for (int j = 0; j < KX; ++j) a = fmaf(A.w[(KY - 1) * KX + j], win[p + j], a);
      // round_() (half to even), then the narrowing cast; blur results lie in [0, 255]
      out[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(a), p & 3, out[p >> 2]);
    }
#pragma unroll
    for (int i = KY - 2; i >= 1; --i)
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        float a = acc[i - 1][p];
#pragma unroll
        for (int j = 0; j < KX; ++j) a = fmaf(A.w[i * KX + j], win[p + j], a);
        acc[i][p] = a;
      }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      float a = fmaf(A.w[0], win[p], 0.f);
#pragma unroll
      for (int j = 1; j < KX; ++j) a = fmaf(A.w[j], win[p + j], a);
      acc[0][p] = a;
    }
    if (t - t_first >= KY - 1 && xs < w) {
      u32x4 v = {out[0], out[1], out[2], out[3]};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(yp + (size_t)(t - RY) * w + xs));
    }
  };
  for (int t = t_first; t <= t_last; t += kDwkPF) {
    dwk_static_for<kDwkPF>([&](auto r) {
      if (t + decltype(r)::value <= t_last) row_step(t + decltype(r)::value, r);
    });
  }
}

// ---------------------------------------------------------------------------------------------
bool dwk_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w, int ky, int kx, int border) {
  const char* v = getenv("MV_FORCE_U8X4");
  if (v && *v && *v != '0') return false;
  const bool ks = (ky == 3 || ky == 5 || ky == 7) && (kx == 3 || kx == 5 || kx == 7) && !(ky == 3 && kx == 3);
  return ks && border != MV_BORDER_VALID && (w % 16 == 0) && w >= 16 && h >= 1 && ((uintptr_t)x % 16 == 0) &&
         ((uintptr_t)y % 16 == 0);
}

template <int KY, int KX>
static int dwk_launch(const DwkU8Args& a, int border, hipStream_t s) {
  if (border == MV_BORDER_REFLECT)
    hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_REFLECT>), dim3(a.nblocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_ZERO>), dim3(a.nblocks), dim3(256), 0, s, a);
  return check_launch("k_dwk_u8");
}

// w2d: KY*KX host taps (row-major), or nullptr with the two 1-D factors (kernel2d = k1d_y[:, None] * k1d_x, one fp32
// product per tap, _misc.py:97)
int launch_dwk_u8x16(const uint8_t* x, uint8_t* y, const float* w2d, const float* k1d_x, const float* k1d_y,
                     int64_t planes, int h, int w, int ky, int kx, int border, hipStream_t s) {
  DwkU8Args a = {};
  a.x = x, a.y = y, a.h = h, a.wdt = w;
  for (int j = 0; j < ky; ++j)
    for (int i = 0; i < kx; ++i) a.w[j * kx + i] = w2d ? w2d[j * kx + i] : k1d_y[j] * k1d_x[i];
  a.col_segs = (w + 1023) / 1024;
  int rows = 64;
  if (const char* e = getenv("MV_DWK_U8_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > h) rows = h;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  a.nitems = (long long)planes * a.strips * a.col_segs;
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "dwk_u8: batch too large for one launch");
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
  switch (ky * 10 + kx) {
    case 35: return dwk_launch<3, 5>(a, border, s);
    case 37: return dwk_launch<3, 7>(a, border, s);
    case 53: return dwk_launch<5, 3>(a, border, s);
    case 55: return dwk_launch<5, 5>(a, border, s);
    case 57: return dwk_launch<5, 7>(a, border, s);
    case 73: return dwk_launch<7, 3>(a, border, s);
    case 75: return dwk_launch<7, 5>(a, border, s);
    case 77: return dwk_launch<7, 7>(a, border, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "dwk_u8: kernel size (%d, %d)", ky, kx);
}

}  // namespace mv
