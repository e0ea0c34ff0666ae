// tiefix_u8.hip -- the reference's uint8 Gaussian blur, bit for bit, at the cost of the separable pair.
//
// gaussian_blur_image on uint8 (transforms/v2/functional/_misc.py:147-163): .to(float32) -> pad(reflect) -> conv2d with the
// outer-product kernel k2d[j][i] = fl(ky[j] * kx[i]) -> round_() -> .to(uint8).  Per pixel that is ONE fp32 fma chain over the
// ky * kx taps in row-major order from +0 (V2 below; oracle/oracle.c::orc_gaussian_blur_u8, which equals the reference's outputs
// on every uint8 fixture pixel and on 37.7 M random ones).  The chain is what makes the exact form slow: 25 / 49 / 529 dependent
// fmas per pixel for 5x5 / 7x7 / 23x23 (0.60 / 1.27 / 49 ms on 32 x 4K uint8), against kx + ky for the separable pair
// (0.35 / 0.39 / 1.2 ms).  The pair is another association of the same sum, so its ROUNDED result can differ only where the
// sum sits within the two forms' rounding noise of a tie n + 0.5 -- about 1e-5 of the pixels.  This file makes that precise and
// repairs exactly those pixels:
//
//   inputs x are integers in [0, 255], taps kx[i], ky[j] >= 0 with sum <= 1 (a Gaussian), every partial sum is < 256, so every
//   fp32 rounding of a partial sum errs by at most ulp(256) / 2 = 2^-17.
//     S2 = sum fl(ky[j] kx[i]) x        the real number the 2-D chain approximates      |V2 - S2| <= kx ky 2^-17
//     S1 = sum ky[j] kx[i] x            the real number the separable pair approximates |S2 - S1| <= 255 * 2^-24 < 2 * 2^-17
//     V1 = column chain over row chains: kx roundings per row value, carried through weights that sum to <= 1, plus ky
//          roundings of the column chain                                                |V1 - S1| <= (kx + ky) 2^-17
//   =>  |V1 - V2| <= M = (kx ky + kx + ky + 2) 2^-17   (5x5: 2.8e-4, 7x7: 5.0e-4, 23x23: 4.4e-3, 63x63: 0.031).
//   If V1 is farther than M from every n + 0.5, V2 lies on the same side of every tie: round(V2) == round(V1).
//
//   pass 1   the separable kernel (k_dwk_u8<.., separable, TIES> for sides <= 9, k_sepstream for sides <= 63) rounds and stores
//            every pixel and appends each LANE-ROW (16 / 4 / 2 pixels) that holds a value with |v - rint(v)| > 0.5 - M to a list
//            in the caller's workspace (one wave-aggregated atomic per flagged wave-row);
//   pass 2   k_u8_tie_fixup recomputes the pixels of the listed lane-rows with the 2-D chain and overwrites them.  5x5: 0.9 % of
//            the lane-rows, 23x23: 3.5 %.  If the list overflows its capacity the fix-up recomputes EVERY pixel (slow, exact).
// The result equals mv_gaussian_blur_u8 (the single 2-D pass) bit for bit; tests compare the two on random images, on images
// built to sit on ties, and with a list too small on purpose.
#include "mv_common.h"

namespace mv {

typedef unsigned int u32a1 __attribute__((aligned(1)));

struct TieFixArgs {
  const uint8_t* x;
  uint8_t* y;
  TieList* ties;
  Taps1D t;       // t.x[kx], t.y[ky]
  int h, w, kx, ky;
  long long total_pixels;
};

// One pixel of the reference's 2-D chain.  tx / ty: the 1-D taps in LDS (a run-time index into the by-value kernel argument would
// send it through scratch), tx zero-padded to KXB (a zero tap is an exact no-op of the chain: fma(0, x, acc) == acc for the
// finite x of a uint8 image).  The KXB bytes of a kernel row are loaded together, and the next row's while this row's fmas run:
// written as one load per tap the kernel waited a memory round trip per tap (25 taps: 15 us per pixel, 0.4 ms for the 1 % of a
// 32 x 4K batch that 5x5 flags).
template <int KXB>
__device__ inline uint8_t blur2d_u8(const uint8_t* xp, int h, int w, int oy, int ox, int kx, int ky, const float* tx, const float* ty) {
  const int ry = ky / 2, rx = kx / 2;
  constexpr int NW = (KXB + 3) / 4;  // dwords that cover a kernel row
  const bool inner = ox - rx >= 0 && ox - rx + 4 * NW <= w;  // 4 * NW consecutive bytes of the row exist: no reflection, no clamping
  auto load_row = [&](int j, unsigned char (&v)[KXB]) {
    const uint8_t* row = xp + (size_t)reflect_clamp(oy + j - ry, h) * w;
    if (inner) {
      // NW dword loads at any byte address (gfx950 takes them) instead of KXB byte loads: a scattered byte load costs the
      // address path as much as a dword load, and the fix-up was bound by exactly that
      const u32a1* seg = reinterpret_cast<const u32a1*>(row + (ox - rx));
      unsigned q[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) q[k] = seg[k];
#pragma unroll
      for (int i = 0; i < KXB; ++i) v[i] = (unsigned char)(q[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
      for (int i = 0; i < KXB; ++i) v[i] = row[reflect_clamp(ox + min(i, kx - 1) - rx, w)];
    }
  };
  unsigned char cur[KXB], nxt[KXB];
  load_row(0, cur);
  float acc = 0.f;
  for (int j = 0; j < ky; ++j) {
    if (j + 1 < ky) load_row(j + 1, nxt);
    const float wy = ty[j];
#pragma unroll
    for (int i = 0; i < KXB; ++i) acc = fmaf(wy * tx[i], (float)cur[i], acc);  // kernel2d[j][i] = fl(ky[j] * kx[i]) (_misc.py:97), then the chain
#pragma unroll
    for (int i = 0; i < KXB; ++i) cur[i] = nxt[i];
  }
  return (uint8_t)(int)__builtin_rintf(acc);  // round_() (half to even), then .to(uint8): the value lies in [0, 255]
}

// A flagged LANE-ROW (NPX consecutive pixels) in one thread: the NPX + KXB - 1 bytes of a kernel row are loaded and converted
// ONCE and feed all NPX chains (one pixel per thread re-loaded and re-converted them NPX times: 23 x 23 spent 4 of its 5.7 ms
// here).  Every pixel's chain is unchanged: taps in (row, column) order from +0, weight fl(ky[j] * kx[i]).  Needs the whole window
// inside the row (no reflection in x) and all NPX pixels inside the image; the caller sends the rest through blur2d_u8.
template <int KXB, int NPX>
__device__ inline void blur2d_u8_run(const uint8_t* xp, int h, int w, int oy, int ox0, int kx, int ky, const float* tx, const float* ty,
                                     uint8_t* out) {
  const int ry = ky / 2, rx = kx / 2;
  // 16-byte loads at any byte address (gfx950 takes them): every lane reads its own row segment
  constexpr int NB = NPX + KXB - 1, NQ = (NB + 15) / 16, NWD = 4 * NQ;
  typedef unsigned int u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
  float txr[KXB];
#pragma unroll
  for (int i = 0; i < KXB; ++i) txr[i] = tx[i];
  unsigned cur[NWD], nxt[NWD];
  auto load_row = [&](int j, unsigned (&q)[NWD]) {
    const u32x4a1* seg = reinterpret_cast<const u32x4a1*>(xp + (size_t)reflect_clamp(oy + j - ry, h) * w + (ox0 - rx));
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const u32x4a1 v = seg[k];
      q[4 * k] = v.x, q[4 * k + 1] = v.y, q[4 * k + 2] = v.z, q[4 * k + 3] = v.w;
    }
  };
  load_row(0, cur);
  float acc[NPX];
#pragma unroll
  for (int p = 0; p < NPX; ++p) acc[p] = 0.f;
  for (int j = 0; j < ky; ++j) {
    if (j + 1 < ky) load_row(j + 1, nxt);
    const float wy = ty[j];
    float f[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) f[b] = (float)((cur[b >> 2] >> (8 * (b & 3))) & 0xffu);
#pragma unroll
    for (int i = 0; i < KXB; ++i) {
      const float wgt = wy * txr[i];  // kernel2d[j][i] = fl(ky[j] * kx[i]) (_misc.py:97)
#pragma unroll
      for (int p = 0; p < NPX; ++p) acc[p] = fmaf(wgt, f[p + i], acc[p]);
    }
#pragma unroll
    for (int k = 0; k < NWD; ++k) cur[k] = nxt[k];
  }
#pragma unroll
  for (int p = 0; p < NPX; ++p) out[p] = (uint8_t)(int)__builtin_rintf(acc[p]);
}

template <int KXB, int NPX>
__global__ __launch_bounds__(256, KXB <= 23 ? 4 : 2) void k_u8_tie_fixup(const TieFixArgs A) {  // <= 128 VGPRs up to 23 taps: 2.4 -> 1.9 ms at 23 x 23
  __shared__ float tx[64], ty[64];
  if (threadIdx.x < 64) {
    tx[threadIdx.x] = (int)threadIdx.x < A.kx ? A.t.x[min((int)threadIdx.x, kMaxTaps1D - 1)] : 0.f;
    ty[threadIdx.x] = (int)threadIdx.x < A.ky ? A.t.y[min((int)threadIdx.x, kMaxTaps1D - 1)] : 0.f;
  }
  __syncthreads();
  TieList* T = A.ties;
  const unsigned npx = T->npx, cap = T->capacity;
  // the segments' fill counts -> exclusive prefix sums: entry e of the whole list is entry e - pre[s] of segment s
  __shared__ unsigned pre[kTieSegs + 1], cnt[kTieSegs];
  __shared__ unsigned overflow;
  if (threadIdx.x < kTieSegs) cnt[threadIdx.x] = T->seg_count[threadIdx.x];  // 64 loads in flight, not one after the other
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned sum = 0, raw = 0, ovf = 0;
    for (int sg = 0; sg < kTieSegs; ++sg) {
      const unsigned c = cnt[sg];
      pre[sg] = sum, sum += c < cap ? c : cap, raw += c, ovf |= c > cap;
    }
    pre[kTieSegs] = sum, overflow = ovf;
    if (blockIdx.x == 0) T->count = raw;
  }
  __syncthreads();
  const unsigned count = pre[kTieSegs];
  auto entry = [&](long long e) -> unsigned long long {  // binary search over 64 prefix sums
    int lo = 0;
#pragma unroll
    for (int step = kTieSegs / 2; step >= 1; step >>= 1)
      if (pre[lo + step] <= (unsigned)e) lo += step;
    return T->idx[(size_t)lo * cap + ((unsigned)e - pre[lo])];
  };
  const long long stride = (long long)gridDim.x * 256;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long plane_px = (long long)A.h * A.w;
  if (overflow || T->pad != 0) {  // a segment (or a wave's batch) overflowed: every pixel takes the reference's chain
    for (long long i = gid; i < A.total_pixels; i += stride) {
      const long long plane = i / plane_px, r = i - plane * plane_px;
      const int oy = (int)(r / A.w), ox = (int)(r - (long long)oy * A.w);
      A.y[i] = blur2d_u8<KXB>(A.x + plane * plane_px, A.h, A.w, oy, ox, A.kx, A.ky, tx, ty);
    }
    return;
  }
  (void)npx;  // == NPX (launch_gaussian_blur_u8_hybrid picks the instantiation from the same hybrid_npx())
  constexpr int NB = NPX + KXB - 1, NWD = 4 * ((NB + 15) / 16);  // dwords blur2d_u8_run loads per kernel row
  const int rx = A.kx / 2;
  // pass 1: the lane-rows whose window lies inside the row (all but the ones at the image's left / right edge)
  for (long long e = gid; e < (long long)count; e += stride) {  // one flagged lane-row per thread
    const unsigned long long first = entry(e);
    const long long plane = (long long)(first / (unsigned long long)plane_px), r = (long long)first - plane * plane_px;
    const int oy = (int)(r / A.w), ox0 = (int)(r - (long long)oy * A.w);
    if (ox0 - rx >= 0 && ox0 - rx + 4 * NWD <= A.w && ox0 + NPX <= A.w) {
      uint8_t o[NPX];
      blur2d_u8_run<KXB, NPX>(A.x + plane * plane_px, A.h, A.w, oy, ox0, A.kx, A.ky, tx, ty, o);
      uint8_t* yrow = A.y + plane * plane_px + (long long)oy * A.w;
#pragma unroll
      for (int p = 0; p < NPX; ++p) yrow[ox0 + p] = o[p];
    }
  }
  // pass 2: the edge lane-rows (reflection inside the window, or a ragged lane-row), ONE PIXEL per thread.  Done inside pass 1 -- a
  // lane walking its NPX pixels one after the other -- every wave that held a single edge lane-row (40-50 % of them) ran 4 / 16
  // per-pixel chains behind its fast path: 1.9 ms instead of 1.3 at 23 x 23 (-DMV_TIEFIX_ABLATE_EDGES build).
#ifndef MV_TIEFIX_ABLATE_EDGES
  const long long work = (long long)count * NPX;
  for (long long i = gid; i < work; i += stride) {
    const unsigned long long first = entry(i / NPX);
    const long long plane = (long long)(first / (unsigned long long)plane_px), r = (long long)first - plane * plane_px;
    const int oy = (int)(r / A.w), ox0 = (int)(r - (long long)oy * A.w), ox = ox0 + (int)(i % NPX);
    const bool fast = ox0 - rx >= 0 && ox0 - rx + 4 * NWD <= A.w && ox0 + NPX <= A.w;
    if (!fast && ox < A.w)
      A.y[plane * plane_px + (long long)oy * A.w + ox] = blur2d_u8<KXB>(A.x + plane * plane_px, A.h, A.w, oy, ox, A.kx, A.ky, tx, ty);
  }
#endif
}

// ---------------------------------------------------------------------------------------------- host side
static int hybrid_npx(int kx, int ky) {
  const bool small = (kx <= 7 && ky <= 7) || (kx == 9 && (ky == 7 || ky == 9)) || (kx == 7 && ky == 9);
  return small ? 16 : sepstream_u8_pixels_per_lane(kx, ky);
}

bool gaussian_blur_u8_hybrid_supported(int h, int w, int kx, int ky) {
  // From 25 taps up pair + tie check + fix-up beats the plain 2-D pass (32 x 4K uint8, tools/ab_u8_hybrid_threshold.py, identical
  // bits): 5x5 0.59 against 0.64 ms, 7x7 0.78 against 1.14, 9x9 0.92 against 1.68, 15x15 1.6 against ~20, 23x23 3.1 against 49.
  // (Until the tie list got its 64 counters, the fix-up its lane-row threads and tie_push lost its branch, the break-even was 49
  // taps: 7x7 1.19 against 1.25.)  Smaller kernels (3x5, 3x7, ...) keep the 2-D pass.
  // Narrow images (up to 512 pixels: several strips per wave) come in batches of 100-200 MB whose whole blur takes 0.2-0.4 ms: there
  // the three launches and the fix-up's fixed cost outweigh the pair's saving up to 7x7 (1024 x 3 x 224 x 224: 5x5 0.17 against 0.23 ms,
  // 7x7 0.32 against 0.42; 9x9 0.56 against 0.34 -- profiles/r03_perf_u8_narrow.log)
  int min_taps = w <= 512 ? 49 : 24;  // the plain 2-D pass up to this many taps
  if (const char* e = tune_env("MV_U8_HYBRID_MIN_TAPS")) min_taps = atoi(e);
  if (kx > 63 || ky > 63 || kx * ky <= min_taps || h < 1) return false;
  if (tune_env("MV_U8_NO_HYBRID")) return false;
  if (hybrid_npx(kx, ky) == 16) {
    const int tx = kx < 3 ? 3 : kx, ty = ky < 3 ? 3 : ky;
    return ky > 1 && sep_u8x16_ties_supported(h, w, ty, tx);
  }
  return sepstream_supported(nullptr, nullptr, true, h, w, kx, ky);
}

static int64_t tie_capacity(int64_t planes, int h, int w, int npx) {
  // lane-rows of the batch / 8 (the rigorous bound flags 1-12 % of them), at least 4096
  const int64_t lane_rows = planes * h * ((w + npx - 1) / npx);
  int64_t cap = lane_rows / 8;
  if (cap < 4096) cap = 4096;
  if (cap > 0x7fffffffLL) cap = 0x7fffffffLL;
  return cap;
}

int64_t u8_tie_workspace_bytes(int64_t planes, int h, int w, int npx) {
  return (int64_t)offsetof(TieList, idx) + 8 * tie_capacity(planes, h, w, npx);
}

struct TieHeader {
  unsigned count, capacity, npx, pad;
};
__global__ void k_u8_tie_reset(TieList* T, unsigned capacity, unsigned npx) {
  if (threadIdx.x == 0) T->count = 0, T->capacity = capacity, T->npx = npx, T->pad = 0;
  T->seg_count[threadIdx.x] = 0;
}

int launch_gaussian_blur_u8_hybrid(const uint8_t* x, uint8_t* y, int64_t planes, int h, int w, const float* k1d_x, int kx,
                                   const float* k1d_y, int ky, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  const int npx = hybrid_npx(kx, ky);
  int64_t cap = (workspace_bytes - (int64_t)offsetof(TieList, idx)) / 8;
  if (workspace == nullptr || cap < 1 || (uintptr_t)workspace % 8)
    return set_error(MV_ERR_INVALID_ARGUMENT, "gaussian_blur_u8 (exact, separable cost): workspace of mv_gaussian_blur_u8_workspace_bytes() "
                     "bytes, 8-byte aligned, needed (got %lld bytes)", (long long)workspace_bytes);
  const int64_t want = tie_capacity(planes, h, w, npx);
  if (const char* e = tune_env("MV_U8_TIE_CAP")) cap = atoll(e) > 0 ? atoll(e) : cap;  // tests: force the overflow path
  else if (cap > want) cap = want;
  TieList* T = static_cast<TieList*>(workspace);
  const int64_t seg_cap = cap / kTieSegs;  // entries per segment (0 for a tiny workspace: nothing fits, the fix-up recomputes every pixel)
  hipLaunchKernelGGL(k_u8_tie_reset, dim3(1), dim3(kTieSegs), 0, s, T, (unsigned)seg_cap, (unsigned)npx);
  const float thresh = tie_threshold(kx, ky);
  int rc;
  if (npx == 16) {
    float px[9], py[9];
    const int tx = kx < 3 ? 3 : kx, ty = ky < 3 ? 3 : ky;
    for (int i = 0; i < 9; ++i) px[i] = 0.f, py[i] = 0.f;
    for (int i = 0; i < kx; ++i) px[(tx - kx) / 2 + i] = k1d_x[i];
    for (int i = 0; i < ky; ++i) py[(ty - ky) / 2 + i] = k1d_y[i];
    rc = launch_sep_u8x16(x, y, px, py, planes, h, w, ty, tx, s, T, thresh);
  } else {
    rc = launch_sepstream(x, y, true, planes, h, w, k1d_x, kx, k1d_y, ky, s, T, thresh);
  }
  if (rc) return rc;
  TieFixArgs a = {};
  a.x = x, a.y = y, a.ties = T, a.h = h, a.w = w, a.kx = kx, a.ky = ky;
  a.total_pixels = (long long)planes * h * w;
  for (int i = 0; i < kx; ++i) a.t.x[i] = k1d_x[i];
  for (int j = 0; j < ky; ++j) a.t.y[j] = k1d_y[j];
  // a persistent grid that strides over the list (its length is only known on the device); the row width is a template bucket
  int blocks = 8192;  // measured on 32 x 4K: 2048 workgroups 17-20 % slower, 4096 +10 %, 16384 the same, 32768 +5 % (MV_TIEFIX_BLOCKS, tuning build)
  if (const char* e = tune_env("MV_TIEFIX_BLOCKS")) blocks = atoi(e) > 0 ? atoi(e) : blocks;
  const dim3 grid((unsigned)blocks), block(256);
#define MV_TIEFIX(KXB_, NPX_) hipLaunchKernelGGL((k_u8_tie_fixup<KXB_, NPX_>), grid, block, 0, s, a)
  if (npx == 16) {  // k_dwk_u8's lane-rows (kx <= 9)
    if (kx <= 5) MV_TIEFIX(5, 16);
    else if (kx <= 7) MV_TIEFIX(7, 16);
    else MV_TIEFIX(9, 16);
  } else if (npx == 4) {  // k_sepstream's, kernel sides up to 31
    if (kx <= 9) MV_TIEFIX(9, 4);
    else if (kx <= 15) MV_TIEFIX(15, 4);
    else if (kx <= 23) MV_TIEFIX(23, 4);
    else MV_TIEFIX(31, 4);
  } else {  // 2 pixels per lane above 31
    if (kx <= 31) MV_TIEFIX(31, 2);
    else if (kx <= 47) MV_TIEFIX(47, 2);
    else MV_TIEFIX(63, 2);
  }
#undef MV_TIEFIX
  return check_launchf("%s+k_u8_tie_fixup", npx == 16 ? "k_dwk_u8<separable,ties>" : "k_sepstream<u8,ties>");
}

}  // namespace mv
