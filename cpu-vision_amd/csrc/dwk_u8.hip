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
// Any width >= 16 and any row alignment (16-byte accesses at any byte address: tools/micro/unaligned.hip): as in
// dw3x3_u8.hip the lane at a ragged right edge is anchored at w - 16 and recomputes the pixels it shares with its left
// neighbour, the neighbour pixels that cannot come by shuffle are fetched as border-mapped single bytes, and images up
// to 512 pixels wide pack 64 / lpr strips into one wave.
#include <cstdlib>
#include <utility>

#include "mv_common.h"

namespace mv {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32b __attribute__((aligned(1)));
#ifdef MV_DWK_ALIGNED_TYPES  // A/B only: aligned 16-byte types (valid for W % 16 == 0)
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
#else
typedef unsigned int u32x4b __attribute__((ext_vector_type(4), aligned(1)));  // 16-byte access at any byte address
#endif

struct DwkU8Args {
  const uint8_t* x;
  uint8_t* y;
  float w[49];  // row-major KY x KX
  int h, wdt;
  int rows, strips, col_segs;  // col_segs = ceil(w / 1024)
  int lpr;                     // lanes per image row (power of two <= 64)
  unsigned nblocks;
  long long nitems;  // waves
  long long units;   // planes * strips
};

constexpr int kDwkPF = 4;  // raw rows in flight per wave

struct RawRow {
  u32x4 v;         // 16 pixels
  unsigned halo;   // the 4 bytes left of (sides[0] == kLoad) or right of the lane's 16 pixels, for the lanes that load them
};

// How a lane gets the RX pixels left / right of its 16: from the neighbouring lane (shuffle), from one 4-byte load of its
// own (segment ends; the lane anchored at a ragged right edge and its left neighbour), from its own pixels (the image border:
// reflect-101 or zeros) or -- only when an image border falls inside those 4 bytes -- byte by byte at use time.
enum { kShuffle = 0, kLoad = 1, kBorder = 2, kBytes = 3 };

struct DwkRole {
  int xs;
  bool valid;
  int sides[2];  // [0] left, [1] right
};

__device__ inline DwkRole dwk_role(int seg, int lane_in_row, int lpr, int w) {
  DwkRole r;
  const int nom = seg * 1024 + lane_in_row * 16;
  r.valid = nom < w;
  const bool anchored = r.valid && nom + 16 > w;             // ragged right edge: anchor at w - 16
  const bool next_anchored = nom + 16 < w && nom + 32 > w;     // my right neighbour lane is the anchored one
  r.xs = anchored ? w - 16 : nom;
  r.sides[0] = r.sides[1] = kShuffle;
  if (!r.valid) return r;
  if (r.xs == 0) r.sides[0] = kBorder;
  else if (lane_in_row == 0 || anchored) r.sides[0] = r.xs >= 4 ? kLoad : kBytes;
  if (r.xs + 16 == w) r.sides[1] = kBorder;
  else if (lane_in_row == lpr - 1 || next_anchored) r.sides[1] = r.xs + 20 <= w ? kLoad : kBytes;
  if (r.sides[0] == kLoad && r.sides[1] == kLoad) r.sides[1] = kBytes;  // one prefetched dword per lane (2-lane rows only)
  return r;
}

template <int BORDER>
__device__ inline unsigned dwk_border_px(const uint8_t* rowp, int c, int w) {
  if (rowp == nullptr) return 0u;
  if (BORDER == MV_BORDER_REFLECT) return rowp[reflect_clamp(c, w)];
  return (c >= 0 && c < w) ? rowp[c] : 0u;
}

__device__ inline RawRow dwk_load(const uint8_t* rowp, const DwkRole& L) {
  RawRow q;
  q.v = (u32x4){0u, 0u, 0u, 0u};
  q.halo = 0u;
  if (rowp == nullptr || !L.valid) return q;
  const u32x4b t = *reinterpret_cast<const u32x4b*>(rowp + L.xs);
  q.v = (u32x4){t.x, t.y, t.z, t.w};
  if (L.sides[0] == kLoad || L.sides[1] == kLoad)
    q.halo = *reinterpret_cast<const u32b*>(rowp + (L.sides[0] == kLoad ? L.xs - 4 : L.xs + 16));
  return q;
}

__device__ inline float dwk_ub(unsigned word, int byte) { return (float)((word >> (8 * byte)) & 0xffu); }

// fp32 window of columns xs-RX .. xs+15+RX; rowp (the row `q` came from) is only touched on the kBytes path
template <int RX, int BORDER>
__device__ inline void dwk_window(const RawRow& q, const DwkRole& L, const uint8_t* rowp, int w, float (&win)[16 + 2 * RX]) {
  const unsigned wd[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
#pragma unroll
  for (int i = 0; i < 16; ++i) win[RX + i] = dwk_ub(wd[i >> 2], i & 3);
  const unsigned up = __shfl_up(wd[3], 1);    // lane-1's last dword: bytes 3, 2, 1 are columns xs-1, xs-2, xs-3
  const unsigned dn = __shfl_down(wd[0], 1);  // lane+1's first dword: bytes 0, 1, 2 are columns xs+16, +17, +18
  const unsigned lw = L.sides[0] == kLoad ? q.halo : up;
  const unsigned rw = L.sides[1] == kLoad ? q.halo : dn;
#pragma unroll
  for (int i = 0; i < RX; ++i) {
    // image border from the lane's own pixels: column -1-i -> 1+i, column w+i -> w-2-i (reflect-101), or zero
    const float lb = (BORDER == MV_BORDER_REFLECT) ? win[RX + 1 + i] : 0.f;
    const float rb = (BORDER == MV_BORDER_REFLECT) ? win[RX + 14 - i] : 0.f;
    win[RX - 1 - i] = L.sides[0] == kBorder ? lb : dwk_ub(lw, 3 - i);
    win[RX + 16 + i] = L.sides[1] == kBorder ? rb : dwk_ub(rw, i);
  }
  if (L.sides[0] == kBytes || L.sides[1] == kBytes) {  // an image border inside the 4 neighbouring bytes: W < 20, or the
                                                       // lane left of an anchored lane that keeps fewer than 4 own pixels
#pragma unroll
    for (int i = 0; i < RX; ++i) {
      if (L.sides[0] == kBytes) win[RX - 1 - i] = (float)dwk_border_px<BORDER>(rowp, L.xs - 1 - i, w);
      if (L.sides[1] == kBytes) win[RX + 16 + i] = (float)dwk_border_px<BORDER>(rowp, L.xs + 16 + i, w);
    }
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

// MULTI: several strips per wave (images up to 512 pixels wide); otherwise the strip -- and with it every row address -- is
// wave-uniform and stays in scalar registers
template <int KY, int KX, int BORDER, bool MULTI>
__global__ __launch_bounds__(256) void k_dwk_u8(const DwkU8Args A) {
  constexpr int RY = KY / 2, RX = KX / 2;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
  if (item >= A.nitems) return;
  const int seg = (int)(item % A.col_segs);
  const long long t2 = item / A.col_segs;
  // narrow images: the wave's 64 / lpr lane groups take consecutive (plane, strip) units, across plane boundaries
  const long long unit_raw = MULTI ? t2 * (kWave / A.lpr) + lane / A.lpr : t2;
  const bool unit_ok = MULTI ? unit_raw < A.units : true;
  const long long unit = unit_ok ? unit_raw : A.units - 1;
  const int strip = (int)(unit % A.strips);
  const long long plane = unit / A.strips;
  const int h = A.h, w = A.wdt;
  DwkRole L = dwk_role(seg, MULTI ? (lane & (A.lpr - 1)) : lane, MULTI ? A.lpr : kWave, w);
  if (!unit_ok) L.valid = false, L.sides[0] = L.sides[1] = kShuffle;
  const int xs = L.xs;
  const int y0 = strip * A.rows, y1 = min(y0 + A.rows, h);
  const size_t poff = (size_t)plane * h * w;
  const uint8_t* xp = A.x + poff;
  uint8_t* yp = A.y + poff;
  const int t_first = y0 - RY, t_last = y1 - 1 + RY;
  const int t_loop_last = y0 + A.rows - 1 + RY;  // uniform trip count over the wave's groups (the last strip may be short)

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
  dwk_static_for<kDwkPF>([&](auto r) { ring[decltype(r)::value] = dwk_load(row_ptr(t_first + decltype(r)::value), L); });

  auto row_step = [&](const int t, auto slot) {
    constexpr int sl = decltype(slot)::value;
    const RawRow raw = ring[sl];
    ring[sl] = dwk_load(row_ptr(t + kDwkPF), L);
    float win[16 + 2 * RX];
    dwk_window<RX, BORDER>(raw, L, row_ptr(t), w, win);  // shuffles run for every lane (uniform control flow)
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
    if (t - t_first >= KY - 1 && t <= t_last && L.valid)
      __builtin_nontemporal_store((u32x4b){out[0], out[1], out[2], out[3]}, reinterpret_cast<u32x4b*>(yp + (size_t)(t - RY) * w + xs));
  };
  for (int t = t_first; t <= t_loop_last; t += kDwkPF) {
    dwk_static_for<kDwkPF>([&](auto r) {
      if (t + decltype(r)::value <= t_loop_last) row_step(t + decltype(r)::value, r);
    });
  }
}

// ---------------------------------------------------------------------------------------------
bool dwk_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w, int ky, int kx, int border) {
  const char* v = tune_env("MV_FORCE_U8X4");
  if (v && *v && *v != '0') return false;
  const bool ks = (ky == 3 || ky == 5 || ky == 7) && (kx == 3 || kx == 5 || kx == 7) && !(ky == 3 && kx == 3);
  (void)x, (void)y;
  return ks && border != MV_BORDER_VALID && w >= 16 && h >= 1;
}

template <int KY, int KX>
static int dwk_launch(const DwkU8Args& a, int border, hipStream_t s) {
  const bool multi = a.lpr < kWave;
  if (border == MV_BORDER_REFLECT) {
    if (multi)
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_REFLECT, true>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_REFLECT, false>), dim3(a.nblocks), dim3(256), 0, s, a);
  } else {
    if (multi)
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_ZERO, true>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, MV_BORDER_ZERO, false>), dim3(a.nblocks), dim3(256), 0, s, a);
  }
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
  a.lpr = kWave;
  while (a.lpr > 1 && (a.lpr / 2) * 16 >= w) a.lpr /= 2;
  int rows = 64;  // measured best of 16..540 on 4K frames; shorter while the launch would have fewer than ~8k waves
  while (rows > 8 && planes * ((h + rows - 1) / rows) * a.col_segs / (kWave / a.lpr) < 8192) rows /= 2;
  if (const char* e = tune_env("MV_DWK_U8_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > h) rows = h;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  const int groups = kWave / a.lpr;  // strips per wave
  a.units = (long long)planes * a.strips;
  a.nitems = ((a.units + groups - 1) / groups) * a.col_segs;  // groups > 1 only when col_segs == 1
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
