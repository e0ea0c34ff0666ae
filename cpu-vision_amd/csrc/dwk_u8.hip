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
// SEP = true is the SEPARABLE form of the same blur on the same lane layout (mv_separable_blur_u8 for kernel sides <= 7, 9x9, 9x7, 7x9):
// the row pass (1 x KX taps, ascending from +0) of each unpacked row gives 16 fp32 values, which enter a systolic COLUMN
// chain `acc[i] = fma(k1d_y[i], tmp, acc[i-1])` -- KX + KY fmas per pixel instead of KY * KX (5x5: 10 instead of 25, 7x7: 14
// instead of 49), which takes the uint8 blur off the VALU wall (32 x 4K uint8, 5x5: 0.54-0.62 ms as one 2-D chain).  It is
// the integer recipe of gaussian_blur_image (.to(float32) -> filter -> round_() -> .to(uint8)) around the separable
// factorisation of its kernel: bit-exact against oracle.c's orc_separable_blur_u8, and equal to the single 2-D sum except
// at exact rounding ties (<= 1 LSB, the reference's own tolerance for this op).
// Any width >= 16 and any row alignment (16-byte accesses at any byte address: tools/micro/unaligned.hip): as in
// dw3x3_u8.hip the lane at a ragged right edge is anchored at w - 16 and recomputes the pixels it shares with its left
// neighbour, the neighbour pixels that cannot come by shuffle are fetched as border-mapped single bytes, and images up
// to 512 pixels wide pack 64 / lpr strips into one wave.
#include <cstdlib>
#include <utility>

#include "mv_common.h"

namespace mv {

typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32b __attribute__((aligned(1)));
#ifdef MV_DWK_ALIGNED_TYPES  // A/B only: aligned 16-byte types (valid for W % 16 == 0)
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
#else
typedef unsigned int u32x4b __attribute__((ext_vector_type(4), aligned(1)));  // 16-byte access at any byte address
#endif

constexpr int kSepY = 9;  // where the column taps of the separable form start in DwkU8Args::w (sides up to 9)

struct DwkU8Args {
  const uint8_t* x;
  uint8_t* y;
  float w[49];  // row-major KY x KX; separable form: w[0 .. KX) = k1d_x, w[kSepY .. kSepY + KY) = k1d_y
  int h, wdt;
  int rows, strips, col_segs;  // col_segs = ceil(w / 1024)
  int lpr;                     // lanes per image row (power of two <= 64)
  unsigned nblocks;
  long long nitems;  // waves
  long long units;   // planes * strips
  FramePtrs fp;      // mv_*_v: per-frame base pointers (n == 0: contiguous batch)
  TieList* ties;     // separable form with the tie check (TIES instantiation): lane-rows within tie_thresh of a rounding tie
  float tie_thresh;
};

#ifndef MV_DWK_PF
#define MV_DWK_PF 3
#endif
#ifndef MV_DWK_MINWAVES
#define MV_DWK_MINWAVES 1
#endif
#ifndef MV_DWK_MINWAVES_SEP7
#define MV_DWK_MINWAVES_SEP7 3  // 7x7 separable: 166 VGPRs instead of 178 = 3 waves per SIMD, no spills (0.424 -> 0.413 ms)
#endif
constexpr int kDwkPF = MV_DWK_PF;  // raw rows in flight per wave

struct RawRow {
  u32x4 v;         // 16 pixels
  unsigned halo;   // the 4 bytes left of (sides[0] == kLoad) or right of the lane's 16 pixels, for the lanes that load them
};

// How a lane gets the RX pixels left / right of its 16: from the neighbouring lane (shuffle), from one 4-byte load of its
// own (segment ends; the lane anchored at a ragged right edge and its left neighbour), from its own pixels (the image border:
// reflect-101 or zeros) or -- only when an image border falls inside those 4 bytes -- byte by byte at use time.
enum { kShuffle = 0, kLoad = 1, kBorder = 2, kBytes = 3 };

struct DwkRole {
  int xs;        // first column of the lane's 16 pixels; lanes without pixels get a valid column too (they load, never store)
  int hofs;      // column of the 4 halo bytes this lane loads (== xs for the lanes that need none): always inside the row
  bool valid;
  int sides[2];  // [0] left, [1] right
};

__host__ __device__ inline DwkRole dwk_role(int seg, int lane_in_row, int lpr, int w) {
  DwkRole r;
  const int nom = seg * 1024 + lane_in_row * 16;
  r.valid = nom < w;
  const bool anchored = r.valid && nom + 16 > w;             // ragged right edge: anchor at w - 16
  const bool next_anchored = nom + 16 < w && nom + 32 > w;     // my right neighbour lane is the anchored one
  r.xs = (anchored || !r.valid) ? w - 16 : nom;
  r.hofs = r.xs;
  r.sides[0] = r.sides[1] = kShuffle;
  if (!r.valid) return r;
  if (r.xs == 0) r.sides[0] = kBorder;
  else if (lane_in_row == 0 || anchored) r.sides[0] = r.xs >= 4 ? kLoad : kBytes;
  if (r.xs + 16 == w) r.sides[1] = kBorder;
  else if (lane_in_row == lpr - 1 || next_anchored) r.sides[1] = r.xs + 20 <= w ? kLoad : kBytes;
  if (r.sides[0] == kLoad && r.sides[1] == kLoad) r.sides[1] = kBytes;  // one prefetched dword per lane (2-lane rows only)
  if (r.sides[0] == kLoad) r.hofs = r.xs - 4;
  if (r.sides[1] == kLoad) r.hofs = r.xs + 16;
  return r;
}

// Does any lane of an image of this width take the byte-by-byte path (kBytes)?  Only then is that code compiled in: its
// conditional loads would otherwise force every wait in the row loop to vmcnt(0).
static bool dwk_needs_bytes(int w, int lpr, int col_segs) {
  for (int seg = 0; seg < col_segs; ++seg)
    for (int l = 0; l < lpr; ++l) {
      const DwkRole r = dwk_role(seg, l, lpr, w);
      if (r.valid && (r.sides[0] == kBytes || r.sides[1] == kBytes)) return true;
    }
  return false;
}

template <int BORDER>
__device__ inline unsigned dwk_border_px(const uint8_t* rowp, int c, int w, bool zero_row) {
  if (zero_row) return 0u;
  if (BORDER == MV_BORDER_REFLECT) return rowp[reflect_clamp(c, w)];
  return (c >= 0 && c < w) ? rowp[c] : 0u;
}

// Two UNCONDITIONAL loads per row and lane (every lane holds valid addresses): the row loop is straight-line code, so the
// compiler can count the loads in flight and wait with vmcnt(N > 0) -- with a load under a condition it must wait for
// vmcnt(0), i.e. drain the whole prefetch ring and the last store before every row.  `zero` (a row of the zero border):
// the loaded bytes are replaced by zeros.
__device__ inline RawRow dwk_load(const uint8_t* rowp, const DwkRole& L, bool zero) {
  RawRow q;
  const u32x4b t = *reinterpret_cast<const u32x4b*>(rowp + L.xs);
#if MV_DWK_ABLATE == 3
  const unsigned hv = 0u;
#else
  const unsigned hv = *reinterpret_cast<const u32b*>(rowp + L.hofs);
#endif
  q.v = zero ? (u32x4){0u, 0u, 0u, 0u} : (u32x4){t.x, t.y, t.z, t.w};
  q.halo = zero ? 0u : hv;
  return q;
}

__device__ inline float dwk_ub(unsigned word, int byte) { return (float)((word >> (8 * byte)) & 0xffu); }

// fp32 window of columns xs-RX .. xs+15+RX; rowp (the row `q` came from) is only touched on the kBytes path
template <int RX, int BORDER, bool BYTES>
__device__ inline void dwk_window(const RawRow& q, const DwkRole& L, const uint8_t* rowp, bool zero_row, int w, float (&win)[16 + 2 * RX]) {
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
  if constexpr (BYTES) {
    if (L.sides[0] == kBytes || L.sides[1] == kBytes) {  // an image border inside the 4 neighbouring bytes: W < 20, or the
                                                         // lane left of an anchored lane that keeps fewer than 4 own pixels
#pragma unroll
      for (int i = 0; i < RX; ++i) {
        if (L.sides[0] == kBytes) win[RX - 1 - i] = (float)dwk_border_px<BORDER>(rowp, L.xs - 1 - i, w, zero_row);
        if (L.sides[1] == kBytes) win[RX + 16 + i] = (float)dwk_border_px<BORDER>(rowp, L.xs + 16 + i, w, zero_row);
      }
    }
  }
}

#ifndef MV_DWK_NT
#define MV_DWK_NT 1
#endif
#ifndef MV_DWK_TIE_ABLATE
#define MV_DWK_TIE_ABLATE 0
#endif
#ifndef MV_DWK_ABLATE
#define MV_DWK_ABLATE 0  // profiling builds only (wrong results): 1 = no stores, 2 = no arithmetic (copy with the same
#endif                   // memory pattern), 3 = no halo load
__device__ inline void dwk_store(u32x4b v, u32x4b* p) {
#if MV_DWK_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
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
template <int KY, int KX, int BORDER, bool MULTI, bool SEP, bool BYTES, bool TIES = false>
__global__ __launch_bounds__(256, (SEP && KY == 7 && MV_DWK_MINWAVES < MV_DWK_MINWAVES_SEP7) ? MV_DWK_MINWAVES_SEP7 : ((SEP && KY == 9 && MV_DWK_MINWAVES < 2) ? 2 : MV_DWK_MINWAVES)) void k_dwk_u8(const DwkU8Args A) {
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
  const uint8_t* xp = frame_in<uint8_t>(A.fp, A.x, plane, (size_t)h * w);
  uint8_t* yp = frame_out<uint8_t>(A.fp, A.y, plane, (size_t)h * w);
  const int t_first = y0 - RY, t_last = y1 - 1 + RY;
  const int t_loop_last = y0 + A.rows - 1 + RY;  // uniform trip count over the wave's groups (the last strip may be short)

  // every row index maps to a row of the plane (reflect_clamp is total); a row of the ZERO border is flagged instead
  auto row_ptr = [&](int t) -> const uint8_t* { return xp + (size_t)reflect_clamp(t, h) * w; };
  auto row_zero = [&](int t) -> bool { return BORDER != MV_BORDER_REFLECT && (t < 0 || t >= h); };

  float acc[KY - 1][16];
#pragma unroll
  for (int i = 0; i < KY - 1; ++i)
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[i][p] = 0.f;

  RawRow ring[kDwkPF];
  dwk_static_for<kDwkPF>([&](auto r) {
    ring[decltype(r)::value] = dwk_load(row_ptr(t_first + decltype(r)::value), L, row_zero(t_first + decltype(r)::value));
  });
  constexpr int kTieCap = 512;  // per wave and strip: 64 lanes x <= 128 rows x the 1-3 % the bound flags, with room to spare
  __shared__ unsigned long long tie_lds[TIES ? 4 * kTieCap : 1];
  TieWave tw = {tie_lds + (TIES ? wave * kTieCap : 0), 0, kTieCap - 64, 0};  // the last 64 entries: tie_push's dump slots

  auto row_step = [&](const int t, auto slot) {
    constexpr int sl = decltype(slot)::value;
    const RawRow raw = ring[sl];
    ring[sl] = dwk_load(row_ptr(t + kDwkPF), L, row_zero(t + kDwkPF));
    float win[16 + 2 * RX];
    dwk_window<RX, BORDER, BYTES>(raw, L, row_ptr(t), row_zero(t), w, win);  // shuffles run for every lane (uniform control flow)
    // last stage first: output row t - RY receives kernel row KY-1
    unsigned out[4] = {0u, 0u, 0u, 0u};
    if constexpr (SEP) {
#if MV_DWK_ABLATE == 2
      out[0] = raw.v.x, out[1] = raw.v.y, out[2] = raw.v.z, out[3] = raw.v.w ^ raw.halo;
      if (t - t_first >= KY - 1 && t <= t_last && L.valid)
        dwk_store((u32x4b){out[0], out[1], out[2], out[3]}, reinterpret_cast<u32x4b*>(yp + (size_t)(t - RY) * w + xs));
      return;
#endif
      float tie_far = 0.f, tie_even = 0.f;  // TIES: the largest |blur - rint(blur)| of the lane's 16 pixels (0.5 = exactly on a tie)
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        float tmp = fmaf(A.w[0], win[p], 0.f);  // row pass: 1 x KX, ascending taps from +0
#pragma unroll
        for (int j = 1; j < KX; ++j) tmp = fmaf(A.w[j], win[p + j], tmp);
        // column pass: this row is tap KY-1 of output row t - RY, ..., tap 0 of output row t + RY
        if constexpr (KY == 1) {
          out[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(A.w[kSepY], tmp, 0.f), p & 3, out[p >> 2]);
        } else {
          const float blur = fmaf(A.w[kSepY + KY - 1], tmp, acc[KY - 2][p]);
          out[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(blur, p & 3, out[p >> 2]);  // round_() + narrow in one instruction (see below)
          if constexpr (TIES) {
            // |blur - rint(blur)| for two pixels at a time on the packed adder: rint(v) = (v + 1.5 * 2^23) - 1.5 * 2^23 for
            // 0 <= v < 2^22 (round to nearest even, exactly v_rndne_f32's result), three v_pk_add_f32 + one v_max3_f32 per
            // pair instead of rndne + sub + max per pixel (48 -> 32 VALU instructions per row of 16 pixels)
            if ((p & 1) == 0) {
              tie_even = blur;
            } else {
              const f32x2 v = {tie_even, blur}, big = {12582912.f, 12582912.f};
              const f32x2 sh = v + big;
              const f32x2 r = sh - big;
              const f32x2 d = v - r;
              tie_far = fmaxf(fmaxf(tie_far, fabsf(d.x)), fabsf(d.y));
            }
          }
#pragma unroll
          for (int i = KY - 2; i >= 1; --i) acc[i][p] = fmaf(A.w[kSepY + i], tmp, acc[i - 1][p]);
          acc[0][p] = fmaf(A.w[kSepY], tmp, 0.f);
        }
      }
#if MV_DWK_ABLATE == 1
      if (t - t_first >= KY - 1 && t <= t_last && L.valid && out[0] == 0x12345678u && out[3] == 0x9abcdef0u)
#else
      if (t - t_first >= KY - 1 && t <= t_last && L.valid)
#endif
        dwk_store((u32x4b){out[0], out[1], out[2], out[3]}, reinterpret_cast<u32x4b*>(yp + (size_t)(t - RY) * w + xs));
      if constexpr (TIES) {
        // a lane-row that holds a value within the separable-vs-2-D error bound of a rounding tie: k_u8_tie_fixup recomputes its
        // 16 pixels with the reference's 2-D chain (mv_common.h: TieList).  Rows outside the strip are never flagged.
        const bool stored = t - t_first >= KY - 1 && t <= t_last && L.valid;
#if MV_DWK_TIE_ABLATE == 1  // profiling builds only (wrong results): the flag is computed and dropped
        if (stored && tie_far > A.tie_thresh && xs == 0x7fffffff) tie_push(tw, true, 0ull, lane);
#else
        tie_push(tw, stored && tie_far > A.tie_thresh, ((unsigned long long)plane * h + (unsigned)(t - RY)) * w + xs, lane);
#endif
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      float a = acc[KY - 2][p];
#pragma unroll
      for (int j = 0; j < KX; ++j) a = fmaf(A.w[(KY - 1) * KX + j], win[p + j], a);
      // round_() (half to even) and the narrowing cast are ONE instruction: v_cvt_pk_u8_f32 rounds to nearest even itself and
      // saturates (tools/micro/cvt_pk_u8.hip: identical to rint-then-pack on 2^20 values incl. every x.5 tie), so the
      // v_rndne_f32 that used to sit in front of it was 16 wasted VALU instructions per row
      out[p >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(a, p & 3, out[p >> 2]);
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
      dwk_store((u32x4b){out[0], out[1], out[2], out[3]}, reinterpret_cast<u32x4b*>(yp + (size_t)(t - RY) * w + xs));
  };
  // whole rounds of kDwkPF steps, no per-step guard (the steps past t_loop_last load clamped rows and store nothing): with a
  // guard the number of loads in flight is unknown at every wait and the compiler falls back to vmcnt(0)
  for (int t = t_first; t <= t_loop_last; t += kDwkPF) {
    dwk_static_for<kDwkPF>([&](auto r) { row_step(t + decltype(r)::value, r); });
  }
  if constexpr (TIES) tie_flush(A.ties, tw, lane, blockIdx.x * 4u + (unsigned)wave);
}

// ---------------------------------------------------------------------------------------------
bool dwk_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w, int ky, int kx, int border) {
  const char* v = tune_env("MV_FORCE_U8X4");
  if (v && *v && *v != '0') return false;
  const bool ks = (ky == 3 || ky == 5 || ky == 7) && (kx == 3 || kx == 5 || kx == 7) && !(ky == 3 && kx == 3);
  (void)x, (void)y;
  return ks && border != MV_BORDER_VALID && w >= 16 && h >= 1;
}

template <int KY, int KX, int BORDER, bool SEP>
static void dwk_launch_b(const DwkU8Args& a, hipStream_t s) {
  const bool multi = a.lpr < kWave;
  if constexpr (SEP && BORDER == MV_BORDER_REFLECT && KY > 1) {
    if (a.ties) {  // the caller checked sep_u8x16_ties_supported(): no byte path; narrow images (several strips per wave) included
      if (multi)
        hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, true, true, false, true>), dim3(a.nblocks), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, false, true, false, true>), dim3(a.nblocks), dim3(256), 0, s, a);
      return;
    }
  }
  // the byte-by-byte neighbour path exists only in the (narrow / ragged-by-1..3) MULTI-or-not instantiations that need it
  if (dwk_needs_bytes(a.wdt, a.lpr, a.col_segs)) {
    if (multi)
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, true, SEP, true>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, false, SEP, true>), dim3(a.nblocks), dim3(256), 0, s, a);
  } else {
    if (multi)
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, true, SEP, false>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_dwk_u8<KY, KX, BORDER, false, SEP, false>), dim3(a.nblocks), dim3(256), 0, s, a);
  }
}

template <int KY, int KX, bool SEP>
static int dwk_launch(const DwkU8Args& a, int border, hipStream_t s) {
  if (border == MV_BORDER_REFLECT) {
    dwk_launch_b<KY, KX, MV_BORDER_REFLECT, SEP>(a, s);
  } else if constexpr (!SEP) {
    dwk_launch_b<KY, KX, MV_BORDER_ZERO, false>(a, s);
  } else {
    return set_error(MV_ERR_INVALID_ARGUMENT, "separable uint8 blur: reflect border only");
  }
  return check_launchf(SEP ? "k_dwk_u8<%dx%d,separable>" : "k_dwk_u8<%dx%d,2d>", KY, KX);
}

static void dwk_plan(DwkU8Args& a, int64_t planes, int h, int w, int rows) {
  a.col_segs = (w + 1023) / 1024;
  a.lpr = kWave;
  while (a.lpr > 1 && (a.lpr / 2) * 16 >= w) a.lpr /= 2;
  // strip height: shorter while the launch would have fewer than ~8k waves
  while (rows > 8 && planes * ((h + rows - 1) / rows) * a.col_segs / (kWave / a.lpr) < 8192) rows /= 2;
  if (const char* e = tune_env("MV_DWK_U8_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > h) rows = h;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  const int groups = kWave / a.lpr;  // strips per wave
  a.units = (long long)planes * a.strips;
  a.nitems = ((a.units + groups - 1) / groups) * a.col_segs;  // groups > 1 only when col_segs == 1
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
}

// w2d: KY*KX host taps (row-major), or nullptr with the two 1-D factors (kernel2d = k1d_y[:, None] * k1d_x, one fp32
// product per tap, _misc.py:97)
int launch_dwk_u8x16(const uint8_t* x, uint8_t* y, const float* w2d, const float* k1d_x, const float* k1d_y,
                     int64_t planes, int h, int w, int ky, int kx, int border, hipStream_t s) {
  DwkU8Args a = {};
  a.x = x, a.y = y, a.h = h, a.wdt = w;
  fill_frames(a.fp);
  for (int j = 0; j < ky; ++j)
    for (int i = 0; i < kx; ++i) a.w[j * kx + i] = w2d ? w2d[j * kx + i] : k1d_y[j] * k1d_x[i];
  dwk_plan(a, planes, h, w, 64);  // 64 rows: measured best of 16..540 on 4K frames
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "dwk_u8: batch too large for one launch");
  switch (ky * 10 + kx) {
    case 35: return dwk_launch<3, 5, false>(a, border, s);
    case 37: return dwk_launch<3, 7, false>(a, border, s);
    case 53: return dwk_launch<5, 3, false>(a, border, s);
    case 55: return dwk_launch<5, 5, false>(a, border, s);
    case 57: return dwk_launch<5, 7, false>(a, border, s);
    case 73: return dwk_launch<7, 3, false>(a, border, s);
    case 75: return dwk_launch<7, 5, false>(a, border, s);
    case 77: return dwk_launch<7, 7, false>(a, border, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "dwk_u8: kernel size (%d, %d)", ky, kx);
}

// Separable form: kernel sides in {3, 5, 7} (the caller zero-pads 1 -> 3: an exact no-op of the fma chains) and 9 x 9, 9 x 7, 7 x 9
// (a lane's halo is the 4 bytes either side of its 16 pixels: sides up to 9), reflect border
bool sep_u8x16_supported(int h, int w, int ky, int kx) {
  const bool small = (ky == 3 || ky == 5 || ky == 7) && (kx == 3 || kx == 5 || kx == 7);
  const bool nine = (ky == 9 && (kx == 7 || kx == 9)) || (ky == 7 && kx == 9);
  const bool ks = small || nine;
  return ks && w >= 16 && h >= 1;  // sides may be zero-padded past the image: the reflect map is total and a zero tap is exact
}

bool sep_u8x16_ties_supported(int h, int w, int ky, int kx) {
  if (!sep_u8x16_supported(h, w, ky, kx)) return false;
  int lpr = kWave;
  while (lpr > 1 && (lpr / 2) * 16 >= w) lpr /= 2;
  return !dwk_needs_bytes(w, lpr, (w + 1023) / 1024);  // any lane layout without the byte path (images of 500 x 375, 224 x 224, ... too)
}

int launch_sep_u8x16(const uint8_t* x, uint8_t* y, const float* k1d_x, const float* k1d_y, int64_t planes, int h, int w, int ky,
                     int kx, hipStream_t s, TieList* ties, float tie_thresh) {
  DwkU8Args a = {};
  a.x = x, a.y = y, a.h = h, a.wdt = w;
  a.ties = ties, a.tie_thresh = tie_thresh;
  fill_frames(a.fp);
  for (int i = 0; i < kx; ++i) a.w[i] = k1d_x[i];
  for (int j = 0; j < ky; ++j) a.w[kSepY + j] = k1d_y[j];
#ifndef MV_SEPU8_ROWS
#define MV_SEPU8_ROWS 64
#endif
  int rows = MV_SEPU8_ROWS;
  if (ky >= 7 || kx >= 7) {
    // the 7-tap instantiations hold 3 waves per SIMD = 3 workgroups per CU: with so few in flight a partial last round of
    // workgroups shows (32 x 4K: 64-row strips = 4.25 rounds; 90-row strips = 3.0 rounds, 0.394 -> 0.371 ms).  Pick the strip
    // height in [48, 128] that minimises rounds x rows-per-strip (halo included).
    const long long segs = (w + 1023) / 1024, cap = 256 * (ky == 9 ? 2 : 3);  // (nine column taps: two waves per SIMD)
    long long best = -1;
    for (int r = 48; r <= 128; r += 2) {
      const long long wgs = (planes * ((h + r - 1) / r) * segs + 3) / 4;
      const long long cost = ((wgs + cap - 1) / cap) * (r + ky - 1);
      if (best < 0 || cost <= best) best = cost, rows = r;
    }
  }
  dwk_plan(a, planes, h, w, rows);
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "separable uint8 blur: batch too large for one launch");
  switch (ky * 10 + kx) {
    case 33: return dwk_launch<3, 3, true>(a, MV_BORDER_REFLECT, s);
    case 35: return dwk_launch<3, 5, true>(a, MV_BORDER_REFLECT, s);
    case 37: return dwk_launch<3, 7, true>(a, MV_BORDER_REFLECT, s);
    case 53: return dwk_launch<5, 3, true>(a, MV_BORDER_REFLECT, s);
    case 55: return dwk_launch<5, 5, true>(a, MV_BORDER_REFLECT, s);
    case 57: return dwk_launch<5, 7, true>(a, MV_BORDER_REFLECT, s);
    case 73: return dwk_launch<7, 3, true>(a, MV_BORDER_REFLECT, s);
    case 75: return dwk_launch<7, 5, true>(a, MV_BORDER_REFLECT, s);
    case 77: return dwk_launch<7, 7, true>(a, MV_BORDER_REFLECT, s);
    case 79: return dwk_launch<7, 9, true>(a, MV_BORDER_REFLECT, s);
    case 97: return dwk_launch<9, 7, true>(a, MV_BORDER_REFLECT, s);
    case 99: return dwk_launch<9, 9, true>(a, MV_BORDER_REFLECT, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "separable uint8 blur: kernel size (%d, %d)", ky, kx);
}

}  // namespace mv
