// dw3x3_u8.hip -- uint8 3x3 depthwise family with 16 pixels per lane (gfx950).
//
// Same operators and bit-exact results as the uint8 instantiations of k_dw3x3 / k_dwtile:
//   * gaussian_blur_image on uint8: .to(float32) -> pad(reflect)+conv2d -> round_() -> .to(uint8)
//     (transforms/v2/functional/_misc.py:150-161);
//   * adjust_sharpness_image on uint8 (_color.py:253-275; v1 _functional_tensor.py:809-838, 258-261).
// Those kernels move 4 pixels = 4 BYTES per lane and instruction: at uint8's 2 B/pixel of traffic they are
// instruction-issue-bound at the same pixel rate as fp32 (20-28 % of HBM peak).  Here a lane owns 16 pixels
// (one 16-byte load and one 16-byte store per row), a wave a 1024-pixel segment of an 8-row strip; rows stay
// packed (4 VGPRs) while in flight and are unpacked to fp32 (v_cvt_f32_ubyteN) only inside the 3-row window.
// The arithmetic is the fp32 kernels' (9-tap fma chain in row-major order from +0, round-half-even, truncating
// narrow), so every result is identical to oracle/oracle.c.
// Any width >= 16 and any row alignment: gfx950 global memory takes 16-byte accesses at any byte address
// (tools/micro/unaligned.hip).  When the width is not a multiple of 16, the lane that would hold the ragged right edge
// is anchored at w - 16 instead: it overlaps its left neighbour by 16 - r pixels, recomputes them (same inputs, same
// arithmetic, same bytes) and stores a full 16 bytes -- no byte-wise tail anywhere.  The lanes at a segment end, the anchored
// lane and its left neighbour fetch the one neighbouring pixel they cannot get by shuffle as a single byte, already mapped
// by the border rule.
#include <cstdlib>

#include "mv_common.h"

namespace mv {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4b __attribute__((ext_vector_type(4), aligned(1)));  // 16-byte access at any byte address
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { U8_STORE = 0, U8_SHARP_V2 = 2, U8_SHARP_V1 = 3 };

struct Dw3x3U8Args {
  const uint8_t* x;
  uint8_t* y;
  float w[9];
  float alpha, ratio;
  int h, wdt;
  int rows, strips, col_segs;  // col_segs = ceil(w / 1024)
  int lpr;                     // lanes per image row (power of two <= 64): images up to 512 pixels wide put 64 / lpr strips in a wave
  unsigned nblocks;
  long long nitems;  // waves
  long long units;   // planes * strips
  FramePtrs fp;      // mv_*_v: per-frame base pointers (n == 0: contiguous batch)
};

#ifndef MV_U8_GROUP
#define MV_U8_GROUP 3  // 3 rows: 91 VGPRs = 5 waves per SIMD (4: 112 VGPRs); A/B in tools/ab_libs.py: 2-4 % faster
#endif
constexpr int kU8Group = MV_U8_GROUP;  // raw rows in flight per wave

struct RawU8 {
  u32x4 v;                // 16 pixels
  unsigned hl, hr;        // the pixel left of / right of the lane's 16 (border rule applied), for the lanes that need it
};

// Which lanes cannot take a neighbour pixel from the adjacent lane by shuffle
struct LaneRole {
  int xs;                 // first column of the lane's 16 pixels
  bool valid;             // the lane has pixels
  bool need_l, need_r;    // fetch the left / right neighbour pixel itself
};

__device__ inline LaneRole u8_role(int seg, int lane_in_row, int lpr, int w) {
  LaneRole r;
  const int nom = seg * 1024 + lane_in_row * 16;
  r.valid = nom < w;
  const bool anchored = r.valid && nom + 16 > w;          // ragged right edge: anchor at w - 16
  const bool next_anchored = nom + 16 < w && nom + 32 > w;  // my right neighbour lane is the anchored one
  r.xs = anchored ? w - 16 : nom;
  r.need_l = r.valid && (lane_in_row == 0 || anchored);
  r.need_r = r.valid && (lane_in_row == lpr - 1 || next_anchored || r.xs + 16 >= w);
  return r;
}

template <int BORDER>
__device__ inline unsigned u8_border_px(const uint8_t* rowp, int c, int w) {
  if (BORDER == MV_BORDER_REFLECT) return rowp[reflect_clamp(c, w)];
  return (c >= 0 && c < w) ? rowp[c] : 0u;
}

template <int BORDER>
__device__ inline RawU8 u8_load(const uint8_t* rowp, const LaneRole& L, int w) {
  RawU8 q;
  q.v = (u32x4){0u, 0u, 0u, 0u};
  q.hl = q.hr = 0u;
  if (rowp == nullptr || !L.valid) return q;
  const u32x4b t = *reinterpret_cast<const u32x4b*>(rowp + L.xs);
  q.v = (u32x4){t.x, t.y, t.z, t.w};
  if (L.need_l) q.hl = u8_border_px<BORDER>(rowp, L.xs - 1, w);
  if (L.need_r) q.hr = u8_border_px<BORDER>(rowp, L.xs + 16, w);
  return q;
}

__device__ inline float ub(unsigned word, int byte) { return (float)((word >> (8 * byte)) & 0xffu); }

// 18-wide fp32 window, columns xs-1 .. xs+16, held as register PAIRS so that the taps run on v_pk_fma_f32 (two fp32 fmas per
// instruction and lane, each rounded exactly like v_fma_f32).  The instruction takes each source from ONE aligned register
// pair, so output pixel j is paired with pixel j + 8: H[k] = (win[k], win[k + 8]), and tap dx of the pair (j, j + 8) is
// H[j + dx] whatever dx -- one copy of the window (20 registers for 18 elements), no shuffling moves.
struct WinRow {
  f32x2 H[10];
};

__device__ inline void u8_window(const RawU8& q, const LaneRole& L, WinRow& R) {
  const unsigned wd[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
  float win[18];
#pragma unroll
  for (int i = 0; i < 16; ++i) win[1 + i] = ub(wd[i >> 2], i & 3);
  const unsigned up = __shfl_up(wd[3], 1);    // lane-1's last dword: its top byte is my column xs-1
  const unsigned dn = __shfl_down(wd[0], 1);  // lane+1's first dword: its low byte is my column xs+16
  win[0] = L.need_l ? (float)q.hl : ub(up, 3);
  win[17] = L.need_r ? (float)q.hr : ub(dn, 0);
#pragma unroll
  for (int k = 0; k < 10; ++k) R.H[k] = (f32x2){win[k], win[k + 8]};
}

__device__ inline f32x2 pk_fma(float w, f32x2 x, f32x2 acc) { return __builtin_elementwise_fma((f32x2){w, w}, x, acc); }


// MULTI: several strips per wave (images up to 512 pixels wide); otherwise the strip -- and with it every row address -- is
// wave-uniform and stays in scalar registers
template <int BORDER, int EPI, bool MULTI>
__global__ __launch_bounds__(256) void k_dw3x3_u8(const Dw3x3U8Args A) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
  if (item >= A.nitems) return;
  const int seg = (int)(item % A.col_segs);
  const long long t = item / A.col_segs;
  // narrow images: the wave's 64 lanes are 64 / lpr groups, each covering the full width of a different strip
  // (consecutive (plane, strip) units, across plane boundaries: a 32 x 32 thumbnail has few strips and 32 groups want work)
  const long long unit_raw = MULTI ? t * (kWave / A.lpr) + lane / A.lpr : t;
  const bool unit_ok = MULTI ? unit_raw < A.units : true;
  const long long unit = unit_ok ? unit_raw : A.units - 1;
  const int strip = (int)(unit % A.strips);
  const long long plane = unit / A.strips;
  const int h = A.h, w = A.wdt;
  LaneRole L = u8_role(seg, MULTI ? (lane & (A.lpr - 1)) : lane, MULTI ? A.lpr : kWave, w);
  if (!unit_ok) L.valid = false, L.need_l = false, L.need_r = false;
  const int xs = L.xs;
  const int y_begin = strip * A.rows;
  const int y_end = min(y_begin + A.rows, h);
  const int y_loop_end = y_begin + A.rows;  // uniform trip count over the wave's groups; stores are guarded by y_end
  const uint8_t* xp = frame_in<uint8_t>(A.fp, A.x, plane, (size_t)h * w);
  uint8_t* yp = frame_out<uint8_t>(A.fp, A.y, plane, (size_t)h * w);

  auto row_ptr = [&](int y) -> const uint8_t* {
    if (y > y_end) return nullptr;
    if (BORDER == MV_BORDER_REFLECT) return xp + (size_t)reflect_clamp(y, h) * w;
    return (y >= 0 && y < h) ? xp + (size_t)y * w : nullptr;
  };

  WinRow top, mid;
  u8_window(u8_load<BORDER>(row_ptr(y_begin - 1), L, w), L, top);
  u8_window(u8_load<BORDER>(row_ptr(y_begin), L, w), L, mid);
  RawU8 nxt[kU8Group];
#pragma unroll
  for (int g = 0; g < kU8Group; ++g) nxt[g] = u8_load<BORDER>(row_ptr(y_begin + 1 + g), L, w);

  for (int y = y_begin; y < y_loop_end; y += kU8Group) {
    RawU8 cur[kU8Group];
#pragma unroll
    for (int g = 0; g < kU8Group; ++g) cur[g] = nxt[g];
    if (y + kU8Group < y_loop_end) {
#pragma unroll
      for (int g = 0; g < kU8Group; ++g) nxt[g] = u8_load<BORDER>(row_ptr(y + kU8Group + 1 + g), L, w);
    }
#pragma unroll
    for (int g = 0; g < kU8Group; ++g) {
      WinRow bot;
      u8_window(cur[g], L, bot);  // shuffles run for every lane (uniform control flow)
      const int yy = y + g;
      if (yy < y_end) {
        unsigned out[4] = {0u, 0u, 0u, 0u};
        const bool row_interior = (yy >= 1 && yy < h - 1);
        // Border pixels keep the input (_color.py:262-270 blends the interior view only).  Column 0 is always a lane's pixel 0
        // and column w-1 always a lane's pixel 15 (exact fit, or the lane anchored at w-16), so the blend factor is wave-uniform
        // for pixels 1..14 and a per-lane value for the two ends: fma(0, blur - x, x) == x, no per-pixel select.
        const float alpha_mid = row_interior ? A.alpha : 0.f;
        const float alpha_p0 = (xs == 0) ? 0.f : alpha_mid;
        const float alpha_p15 = (xs + 15 == w - 1) ? 0.f : alpha_mid;
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // pixels j and j + 8: the oracle's 9-tap chain in row-major order from +0, two at a time
          f32x2 acc = pk_fma(A.w[0], top.H[j], (f32x2){0.f, 0.f});
          acc = pk_fma(A.w[1], top.H[j + 1], acc);
          acc = pk_fma(A.w[2], top.H[j + 2], acc);
          acc = pk_fma(A.w[3], mid.H[j], acc);
          acc = pk_fma(A.w[4], mid.H[j + 1], acc);
          acc = pk_fma(A.w[5], mid.H[j + 2], acc);
          acc = pk_fma(A.w[6], bot.H[j], acc);
          acc = pk_fma(A.w[7], bot.H[j + 1], acc);
          acc = pk_fma(A.w[8], bot.H[j + 2], acc);
          if (EPI == U8_STORE) {
            // plain filter: round_() + narrow is ONE instruction -- v_cvt_pk_u8_f32 rounds to nearest even itself and saturates
            // (tools/micro/cvt_pk_u8.hip); the v_rndne_f32 + v_trunc_f32 in front of it were 32 wasted VALU instructions per row
            out[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(acc.x, j & 3, out[j >> 2]);
            out[2 + (j >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(acc.y, j & 3, out[2 + (j >> 2)]);
            continue;
          }
          f32x2 r = {__builtin_rintf(acc.x), __builtin_rintf(acc.y)};  // round_(): half to even (the blend needs the rounded value)
          if (EPI == U8_SHARP_V2) {
            const f32x2 xc = mid.H[j + 1];
            const f32x2 al = {j == 0 ? alpha_p0 : alpha_mid, j == 7 ? alpha_p15 : alpha_mid};
            r = __builtin_elementwise_fma(al, r - xc, xc);  // _color.py:270 (ATen's add_ is one fma)
            // .clamp_(0, 255) is done by the pack below: v_cvt_pk_u8_f32 saturates, and trunc-then-saturate equals clamp-then-trunc
            // for every finite value ((-1, 0) -> -0 -> 0; (255, 256) -> 255; beyond -> 0 / 255): one v_med3_f32 per pixel less
          } else if (EPI == U8_SHARP_V1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int p = j + 8 * e;
              const float xc = mid.H[j + 1][e];
              const bool interior = row_interior && (p != 0 || xs != 0) && (p != 15 || xs + 15 != w - 1);
              const float deg = interior ? r[e] : xc;              // _functional_tensor.py:258-261
              const float t1 = A.ratio * xc;
              const float t2 = A.alpha * deg;
              r[e] = t1 + t2;  // clamp: the saturating pack below
            }
          }
          // .clamp(0, 255).to(uint8): truncation first, exactly like the reference's cast, then the pack's saturation
          out[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(r.x), j & 3, out[j >> 2]);
          out[2 + (j >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(r.y), j & 3, out[2 + (j >> 2)]);
        }
        if (L.valid)
          __builtin_nontemporal_store((u32x4b){out[0], out[1], out[2], out[3]}, reinterpret_cast<u32x4b*>(yp + (size_t)yy * w + xs));
      }
      top = mid, mid = bot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
bool dw3x3_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w) {
  const char* v = tune_env("MV_FORCE_U8X4");
  if (v && *v && *v != '0') return false;
  (void)x, (void)y;
  return w >= 16 && h >= 1;
}

template <int BORDER, int EPI>
static int u8_launch(const Dw3x3U8Args& a, hipStream_t s) {
  if (a.lpr < kWave)
    hipLaunchKernelGGL((k_dw3x3_u8<BORDER, EPI, true>), dim3(a.nblocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_dw3x3_u8<BORDER, EPI, false>), dim3(a.nblocks), dim3(256), 0, s, a);
  return check_launch("k_dw3x3_u8");
}

// epi: 0 = plain filter (round + narrow), 2 = sharpness v2, 3 = sharpness v1
int launch_dw3x3_u8x16(const uint8_t* x, uint8_t* y, const float* w9, int64_t planes, int h, int w, int border, int epi,
                       double factor, hipStream_t s) {
  Dw3x3U8Args a = {};
  a.x = x, a.y = y, a.h = h, a.wdt = w;
  fill_frames(a.fp);
  if (epi == U8_STORE) {
    for (int i = 0; i < 9; ++i) a.w[i] = w9[i];
  } else {
    const float ta = (float)(1.0 / 13.0), tb = (float)(5.0 / 13.0);  // _color.py:253-256
    for (int i = 0; i < 9; ++i) a.w[i] = (i == 4) ? tb : ta;
    a.alpha = (float)(1.0 - factor);
    a.ratio = (float)factor;
  }
  a.col_segs = (w + 1023) / 1024;
  a.lpr = kWave;
  while (a.lpr > 1 && (a.lpr / 2) * 16 >= w) a.lpr /= 2;
  // strip height: 32 rows amortise the 2 halo rows and the first loads' latency (measured on 32 x 4K: 8 rows 0.42 ms,
  // 16-48 rows 0.33 ms, 128 rows 0.37 ms); shorter while the launch would have fewer than ~8k waves
  int rows = 32;
  while (rows > 2 * kU8Group && planes * ((h + rows - 1) / rows) * a.col_segs / (kWave / a.lpr) < 8192) rows /= 2;
  if (const char* e = tune_env("MV_DW3X3_U8_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > h) rows = h;
  rows = ((rows + kU8Group - 1) / kU8Group) * kU8Group;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  const int groups = kWave / a.lpr;  // strips per wave
  a.units = (long long)planes * a.strips;
  a.nitems = ((a.units + groups - 1) / groups) * a.col_segs;  // groups > 1 only when col_segs == 1
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "dw3x3_u8: batch too large for one launch");
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
  if (epi == U8_SHARP_V2) return u8_launch<MV_BORDER_ZERO, U8_SHARP_V2>(a, s);
  if (epi == U8_SHARP_V1) return u8_launch<MV_BORDER_ZERO, U8_SHARP_V1>(a, s);
  if (border == MV_BORDER_REFLECT) return u8_launch<MV_BORDER_REFLECT, U8_STORE>(a, s);
  if (border == MV_BORDER_ZERO) return u8_launch<MV_BORDER_ZERO, U8_STORE>(a, s);
  return set_error(MV_ERR_INVALID_ARGUMENT, "dw3x3_u8: border %d not handled here", border);
}

}  // namespace mv
