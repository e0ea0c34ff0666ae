// mv_common.h -- shared by the gfx950 kernels and the C-ABI shim.  gfx950 (CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/mi355vision.h"

// Tuning knobs (MV_FORCE_*, MV_*_ROWS, ...) exist only in -DMV_TUNING builds: the hook that reads the environment lives
// outside the product sources (tools/tuning/mv_tuning.h, on the include path of those builds only).  The product library
// performs no environment lookup: tune_env() is a constant there and every `if (tune_env(..))` folds away.
#ifdef MV_TUNING
#include "mv_tuning.h"
#else
namespace mv {
constexpr const char* tune_env(const char*) { return nullptr; }
}  // namespace mv
#endif

namespace mv {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kXcds = 8;           // MI355X: 8 XCDs, each with a private 4 MiB L2
constexpr int kMaxTaps2D = MV_MAX_HOST_TAPS_2D;
constexpr int kMaxTaps1D = MV_MAX_TAPS_1D;

// ---- error plumbing (thread-local message, integer status across the ABI) --------------------
int set_error(int code, const char* fmt, ...);
// check_launch also records `what` as the calling thread's last launched kernel (mv_last_kernel()); launchers with
// template variants describe the instantiation ("k_dwtile<f32,3x3,rpt4,vec16,tw256>").
int check_launch(const char* what);
int check_launchf(const char* fmt, ...);

// ---- by-value filter taps (kernel arguments live in SGPRs: every tap is a scalar operand) ----
struct Taps2D {
  float w[kMaxTaps2D];
};
struct Taps1D {
  float x[kMaxTaps1D];
  float y[kMaxTaps1D];
};

// ---- separately allocated frames in ONE launch (mv_*_v entry points) ------------------------------
// A DataLoader hands over frames that were allocated one by one; a launch per 1080p frame is launch-bound (14 us = 44 % of HBM
// peak).  The *_v entries pass a table of per-frame base pointers BY VALUE in the kernel arguments (no device allocation, no
// host-to-device copy): plane p belongs to frame p / ppf and starts (p % ppf) planes into it.  n == 0: the contiguous layout.
constexpr int kMaxFrames = 112;  // 112 x 16 B = 1792 B of the 4 KB kernel-argument segment; longer lists go in several launches
struct FramePtrs {
  const void* x[kMaxFrames];
  void* y[kMaxFrames];
  int ppf;  // planes per frame
  int n;    // frames in the table (0 = unused)
};
// The table of the *_v call in progress on this thread (nullptr outside one); launchers copy it into their kernel arguments.
const FramePtrs* call_frames();
inline void fill_frames(FramePtrs& dst) {
  if (const FramePtrs* f = call_frames()) dst = *f; else dst.n = 0, dst.ppf = 1;
}
template <typename T>
__device__ inline const T* frame_in(const FramePtrs& fp, const void* x, long long plane, size_t plane_elems) {
  if (fp.n == 0) return static_cast<const T*>(x) + (size_t)plane * plane_elems;
  const long long f = plane / fp.ppf;
  return static_cast<const T*>(fp.x[f]) + (size_t)(plane - f * fp.ppf) * plane_elems;
}
template <typename T>
__device__ inline T* frame_out(const FramePtrs& fp, void* y, long long plane, size_t plane_elems) {
  if (fp.n == 0) return static_cast<T*>(y) + (size_t)plane * plane_elems;
  const long long f = plane / fp.ppf;
  return static_cast<T*>(fp.y[f]) + (size_t)(plane - f * fp.ppf) * plane_elems;
}

// ---- index helpers ---------------------------------------------------------------------------
// ATen reflection_pad2d index map, made total by a final clamp (positions that no valid output
// needs may land outside one reflection).
__host__ __device__ inline int reflect_clamp(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

// XCD-aware block remap (T1 in the CDNA guide): blocks are dealt round-robin over the 8 XCDs, so
// block b and b+8 share an L2.  Give every XCD one contiguous chunk of the work list, so that
// vertically adjacent strips (which share halo rows) meet in the same L2.  Bijective for any grid.
__device__ inline unsigned xcd_remap(unsigned bid, unsigned nblocks) {
  unsigned per = nblocks / kXcds, rem = nblocks % kXcds;
  unsigned xcd = bid % kXcds, idx = bid / kXcds;
  // XCDs [0, rem) own per+1 blocks, the rest own per blocks
  unsigned start = xcd * per + (xcd < rem ? xcd : rem);
  return start + idx;
}

// round-half-to-even to uint8, as torch.round_() followed by .to(torch.uint8) on in-range values
__device__ inline unsigned char round_u8(float v) { return (unsigned char)(int)__builtin_rintf(v); }

// ---- exact uint8 Gaussian blur at separable cost (tiefix_u8.hip) --------------------------------------------------------------
// The reference rounds ONE 2-D fp32 chain per pixel (V2); the separable pair (V1) is another association of the same sum and
// differs from it by at most M = (kx * ky + kx + ky + 2) * 2^-17 (proof in tiefix_u8.hip), so round(V1) == round(V2) whenever V1
// is farther than M from every rounding tie n + 0.5.  The separable kernels therefore flag the few lane-rows that hold a value
// within M of a tie and append them here; k_u8_tie_fixup recomputes exactly those pixels with the reference's 2-D chain.
// The list is kTieSegs independent segments, each with its own counter: every wave publishes its strip's batch with ONE atomic
// add, and ~10 k of those to a single address serialised in L2 (9 x 9 on 32 x 4K: 0.83 ms against 0.60 ms with the atomic
// compiled out).  A wave picks its segment from its workgroup id, so 64 counters see 1 / 64 of the adds each.
constexpr int kTieSegs = 64;
struct TieList {
  unsigned count;      // total lane-rows appended (written by k_u8_tie_fixup: a diagnostic, tools/dbg_ties.py)
  unsigned capacity;   // entries that fit PER SEGMENT
  unsigned npx;        // pixels per entry (16: k_dwk_u8, 4 or 2: k_sepstream)
  unsigned pad;        // non-zero: some wave's LDS batch overflowed -> the fix-up recomputes every pixel
  unsigned seg_count[kTieSegs];  // lane-rows appended to each segment (may exceed capacity: then the fix-up recomputes EVERY pixel)
  unsigned long long idx[1];     // segment s at idx[s * capacity]: linear index (plane * h + y) * w + x of an entry's first pixel
};
// Flagged lane-rows are collected per WAVE in LDS and published ONCE, after the wave's strip: no global memory operation sits
// inside the row loop.  (A global atomic per flagged wave-row -- 475 k adds to one counter for 5x5 on 32 x 4K -- serialised in
// L2 and cost 12 ms; a flush inside the loop, however rare, made the compiler wait with vmcnt(0) for the prefetch ring on
// every row and doubled the kernel's time.)  A wave whose buffer fills up (an image built on rounding ties) drops the rest and
// raises TieList::pad: the fix-up then recomputes every pixel.  Every lane of the wave calls tie_push / tie_flush.
struct TieWave {
  unsigned long long* buf;  // wave-private LDS: `cap` entries + 64 dump slots (tie_push)
  int n, cap, over;         // wave-uniform
};
// BRANCH-FREE on purpose: tie_push sits in the row loop of kernels whose prefetch ring depends on that loop being one basic block
// (the compiler counts the loads in flight only across straight-line code).  With `if (no lane flagged) return;` in front, the 9 x 9
// pair + tie check ran 0.86 ms against 0.51 ms with the push compiled out.  Lanes without a flag (and every lane once the buffer
// is full) store into a dump area of 64 entries behind the `cap` usable ones: the buffer holds cap + 64.
__device__ inline void tie_push(TieWave& W, bool flag, unsigned long long first_pixel, int lane) {
  const unsigned long long m = __ballot(flag);
  const int add = __popcll(m);
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  const bool fits = W.n + add <= W.cap;  // wave-uniform
  W.buf[(flag && fits) ? W.n + rank : W.cap + lane] = first_pixel;
  W.over |= fits ? 0 : 1;
  W.n += fits ? add : 0;
}
__device__ inline void tie_flush(TieList* T, const TieWave& W, int lane, unsigned seg) {
  if (W.over && lane == 0) atomicOr(&T->pad, 1u);
  if (W.n == 0) return;
  seg &= kTieSegs - 1;
  unsigned base = 0;
#ifdef MV_TIE_ABLATE_ATOMIC  // profiling builds only (wrong results): every wave writes at the list's start
  if (lane == 0) base = 0;
#else
  if (lane == 0) base = atomicAdd(&T->seg_count[seg], (unsigned)W.n);
#endif
  base = __builtin_amdgcn_readfirstlane(base);
  const unsigned cap = T->capacity;
  for (int i = lane; i < W.n; i += kWave)
    if (base + i < cap) T->idx[(size_t)seg * cap + base + i] = W.buf[i];
}
// flag threshold on |v - rint(v)|: a value is "near a tie" when that distance exceeds 0.5 - M (with a little slack)
inline float tie_threshold(int kx, int ky) {
  const float m = (float)(kx * ky + kx + ky + 2) * (1.0f / 131072.0f);
  return 0.5f - (m * 1.0625f + 1e-6f);
}
int64_t u8_tie_workspace_bytes(int64_t planes, int h, int w, int npx);
bool gaussian_blur_u8_hybrid_supported(int h, int w, int kx, int ky);
int launch_gaussian_blur_u8_hybrid(const uint8_t* x, uint8_t* y, int64_t planes, int h, int w, const float* k1d_x, int kx,
                                   const float* k1d_y, int ky, void* workspace, int64_t workspace_bytes, hipStream_t s);

// ---- launchers implemented in the .hip files -------------------------------------------------
// 3x3 family (dw3x3.hip).  mode: 0 = single filter, 1 = sobel (two outputs), 2 = sharpness v2, 3 = sharpness v1
int launch_dw3x3_f32(const float* x, float* y0, float* y1, const float* w9a, const float* w9b, int64_t planes, int h,
                     int w, int border, hipStream_t s);
int launch_dw3x3_u8(const uint8_t* x, uint8_t* y, const float* w9, int64_t planes, int h, int w, int border,
                    hipStream_t s);
int launch_sharpness(const void* x, void* y, bool u8, int64_t planes, int h, int w, double factor, int v1,
                     float bound, int round_blur, hipStream_t s);
// uint8 KY x KX in {3,5,7}^2 \ {3x3}, 16 pixels per lane (dwk_u8.hip)
bool dwk_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w, int ky, int kx, int border);
int launch_dwk_u8x16(const uint8_t* x, uint8_t* y, const float* w2d, const float* k1d_x, const float* k1d_y,
                     int64_t planes, int h, int w, int ky, int kx, int border, hipStream_t s);
// the same lane layout, separable form (row pass then systolic column chain): kernel sides in {3, 5, 7}, reflect border
bool sep_u8x16_supported(int h, int w, int ky, int kx);
bool sep_u8x16_ties_supported(int h, int w, int ky, int kx);  // the instantiation with the tie check: full 64-lane rows, no byte path
int launch_sep_u8x16(const uint8_t* x, uint8_t* y, const float* k1d_x, const float* k1d_y, int64_t planes, int h, int w, int ky,
                     int kx, hipStream_t s, TieList* ties = nullptr, float tie_thresh = 0.f);
// generic LDS-tiled depthwise (dwtile.hip)
// storage types of the tile kernel
enum { kDtF32 = 0, kDtU8 = 1, kDtF16 = 2, kDtBF16 = 3 };
int launch_dwtile(const void* x, void* y, int dtype, const float* w2d_host, const float* w_dev, const float* k1d_x,
                  const float* k1d_y, int64_t planes, int h, int w, int ky, int kx, int border, hipStream_t s);
// float64 images (dwf64.hip): taps as fp64 1-D pairs (outer product formed in-kernel) or a device (ky, kx) array
int launch_dwf64(const double* x, double* y, const double* w2d_dev, const double* k1d_x, const double* k1d_y, int64_t planes,
                 int h, int w, int ky, int kx, int border, hipStream_t s);
int launch_sharpness_f64(const double* x, double* y, int64_t planes, int h, int w, double factor, int v1, hipStream_t s);
// separable blur and blur+sobel (separable.hip)
int launch_separable(const float* x, float* y, float* gx, float* gy, bool sobel, int64_t planes, int h, int w,
                     const float* k1d_x, int kx, const float* k1d_y, int ky, hipStream_t s);
// register-streaming separable / fused sobel for K in {3,5,7}, aligned rows (sepfast.hip)
bool sepfast_supported(const float* x, const float* o1, const float* o2, int h, int w, int kx, int ky, bool sobel);
int launch_sepfast(const float* x, float* y, float* gx, float* gy, bool sobel, int64_t planes, int h, int w,
                   const float* k1d_x, const float* k1d_y, int k, hipStream_t s);
// row-streaming separable blur for large kernels, 8 < K <= 63 (sepstream.hip)
bool sepstream_supported(const void* x, const void* y, bool u8, int h, int w, int kx, int ky);
int launch_sepstream(const void* x, void* y, bool u8, int64_t planes, int h, int w, const float* k1d_x, int kx,
                     const float* k1d_y, int ky, hipStream_t s, TieList* ties = nullptr, float tie_thresh = 0.f);
int sepstream_u8_pixels_per_lane(int kx, int ky);
// implicit-GEMM conv3x3 + bias + relu on the fp32 MFMA (conv3x3_mfma.hip)
int launch_conv3x3(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                   int cout, int relu, hipStream_t s);

// uint8 3x3 with 16 pixels per lane (dw3x3_u8.hip): epi 0 = filter, 2 = sharpness v2, 3 = sharpness v1
bool dw3x3_u8x16_supported(const uint8_t* x, const uint8_t* y, int h, int w);
int launch_dw3x3_u8x16(const uint8_t* x, uint8_t* y, const float* w9, int64_t planes, int h, int w, int border, int epi,
                       double factor, hipStream_t s);
// first-layer specialisation: cin = 3, cout <= 64, 16-byte stores (conv3x3_c3.hip)
bool conv3x3_c3_supported(const float* x, const float* y, int cin, int cout, int h, int w);
int launch_conv3x3_c3(const void* x, bool in_u8, const float* mean3, const float* std3, const float* w, const float* b,
                      float* y, int64_t n, int h, int wdt, int cout, int relu, hipStream_t s);
int launch_to_float_normalize(const void* x, float* y, bool u8, int64_t n, int c, int64_t hw, const float* mean,
                              const float* stdv, hipStream_t s);

// general-cin conv3x3 (K-chunked MFMA, conv3x3_gen.hip) and the pooling layers (cnn_ops.hip)
bool conv3x3_gen_supported(int cin, int cout, int h, int w);
int launch_conv3x3_gen(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                       int cout, int relu, hipStream_t s);
int launch_maxpool2x2(const float* x, float* y, int64_t planes, int h, int w, hipStream_t s);
int launch_linear(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                  hipStream_t s);
void linear_plan(int64_t n, int k, int m, int* slices, int* slice_len);
int launch_linear_sliced(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                         float* ws, hipStream_t s);
int launch_adaptive_avgpool(const float* x, float* y, int64_t planes, int h, int w, int oh, int ow, hipStream_t s);

// Conv2dNormActivation family (convnorm.hip): conv -> [+bias] -> folded norm -> [+residual] -> activation
struct Epilogue {
  const float* bias;   // [cout] or null
  const float* alpha;  // [cout] (affine != 0)
  const float* beta;
  const float* res;    // same shape as y, or null
  int affine, act;     // affine: 0 none, 1 x*a then +b (FrozenBatchNorm2d), 2 fma(x, a, b) (BatchNorm2d eval)
};
int launch_dwpc3x3(const float* x, const float* w, float* y, int64_t n, int c, int h, int wd, int stride, const Epilogue& e,
                   hipStream_t s);
int launch_conv3x3_smallcin(const float* x, const float* w, float* y, int64_t n, int cin, int h, int wd, int cout, int stride,
                            const Epilogue& e, hipStream_t s);
// x_img_stride / y_img_stride: floats between consecutive images (0 = dense: cin*hw / cout*hw); a channel slice of a
// wider tensor passes the full tensor's strides (deform_conv2d's weight groups)
int launch_conv1x1(const float* x, const float* w, float* y, int64_t n, int cin, int64_t hw, int cout, const Epilogue& e,
                   hipStream_t s, int64_t x_img_stride = 0, int64_t y_img_stride = 0, bool allow_k_slices = true);
// any nn.Conv2d geometry as an implicit GEMM (no columns in HBM, no workspace); one weight group per call
bool conv2d_implicit_supported(int cg, int kh, int kw, int oh, int ow);
int launch_conv2d_implicit(const float* x, const float* w, float* y, int64_t n, int cg, int h, int wd, int mg, int kh, int kw, int sh,
                           int sw, int ph, int pw, int dh, int dw, int oh, int ow, const Epilogue& e, hipStream_t s,
                           int64_t x_img_stride, int64_t y_img_stride);
// the K slicing launch_conv1x1 applies to this shape: slices == 1 -> one ascending-k chain per output; otherwise `slices`
// chains over `slice_len` channels each (the last may be shorter), added in ascending slice order
void conv1x1_plan(int64_t n, int cin, int64_t hw, int cout, int* slices, int* slice_len);
// deform_conv2d forward (deform.hip)
int64_t deform_workspace_bytes_per_image(int cin, int kh, int kw, int oh, int ow);
int launch_deform_conv2d(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                         int64_t n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                         int groups, int offset_groups, int use_mask, void* workspace, int64_t workspace_bytes, hipStream_t s, int act = 0);
int launch_maxpool2d(const float* x, float* y, int64_t planes, int h, int w, int k, int stride, hipStream_t s);

// F.resize(bilinear, antialias) [+ center_crop] [+ preset tail] (resize.hip)
int64_t resize_workspace_bytes(int64_t planes, int h, int w, int oh, int ow, int ct, int cl, int ch, int cw);
int launch_resize(const void* x, void* y, bool u8, int64_t planes, int channels, int h, int w, int oh, int ow, int ct,
                  int cl, int ch, int cw, int preset, const float* mean, const float* stdv, void* workspace,
                  int64_t workspace_bytes, hipStream_t s);

}  // namespace mv
