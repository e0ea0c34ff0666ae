// convnorm.hip -- Conv2dNormActivation (ops/misc.py:68-128) for the MobileNet family on gfx950:
//   conv (no bias when a norm follows) -> folded norm -> [+ residual] -> activation, in ONE kernel per block, for
//     * the stem: dense 3x3, stride 1|2, cin <= 4 (models/mobilenetv2.py:126)            k_conv3x3_smallcin  (VALU)
//     * depthwise 3x3 with a kernel per channel, stride 1|2 (mobilenetv2.py:43-50)        k_dwpc3x3           (VALU)
//     * pointwise 1x1 (mobilenetv2.py:39-41, 52-53) [+ `x + self.conv(x)`, :61-63]        k_conv1x1           (MFMA)
// Arithmetic = oracle/oracle.c::orc_conv2d_affine_act_f32, bit for bit: one fp32 accumulator per output fed by fmaf
// in (channel, ky, kx) order from +0 (the 1x1 kernel's v_mfma_f32_32x32x2_f32 is that chain), `+ bias`, then
//   affine 1  FrozenBatchNorm2d.forward (ops/misc.py:52-61): `x * scale` then `+ bias`, two roundings
//   affine 2  nn.BatchNorm2d in eval mode (ATen batch_norm_cpu): fma(x, alpha, beta)
// then `res + y`, then ReLU / ReLU6 / Hardswish (exact) or SiLU (expf: last-ulp differences from the CPU's).
#include <cstdlib>

#include "mv_common.h"
#include "mv_epilogue.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte access

// ============================================================================================= depthwise, per channel
// thread = 4 consecutive output pixels x a strip of `rows` output rows of one plane: the three input rows of the window
// roll through registers (one new row per output row at stride 1, two at stride 2), 16-byte loads / stores, so a strip
// of R rows costs about (R + 2) row loads and R stores instead of 21 memory instructions per output.  HBM-bound work
// (2 x 4 B per pixel at stride 1); taps in (ky, kx) order from +0 like the oracle.
struct DwpcArgs {
  const float* x;
  const float* w;  // [c][3][3]
  float* y;
  Epilogue e;
  long long total;  // planes * strips * groups_x
  int c, h, wd, oh, ow;
  int rows, strips, groups_x;
};

template <int STRIDE>
__global__ __launch_bounds__(256) void k_dwpc3x3(const DwpcArgs A) {
  constexpr int NIN = 3 + 3 * STRIDE;  // input columns feeding 4 outputs: 6 (stride 1) or 9 (stride 2)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.total) return;
  const int gx = (int)(idx % A.groups_x);
  const long long t = idx / A.groups_x;
  const int st = (int)(t % A.strips);
  const long long plane = t / A.strips;
  const int ch = (int)(plane % A.c);
  const int H = A.h, W = A.wd, OW = A.ow;
  const float* xp = A.x + (size_t)plane * H * W;
  float* yp = A.y + (size_t)plane * A.oh * OW;
  const float* rp = A.e.res ? A.e.res + (size_t)plane * A.oh * OW : nullptr;
  const int ox0 = 4 * gx, ix0 = ox0 * STRIDE - 1;
  const int oy0 = st * A.rows, oy1 = min(oy0 + A.rows, A.oh);

  float wk[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) wk[i] = A.w[(size_t)ch * 9 + i];
  const ChannelTerms ct = channel_terms(A.e, ch);
  const Clamp cl = make_clamp(A.e.act);

  // Aligned rows (W a multiple of 4 * STRIDE: every MobileNet map): a thread's columns ix0 + 1 .. are whole 16-byte pieces
  // and its left / right neighbour columns are the adjacent LANE's pieces -- they come by shuffle instead of two more (scalar)
  // load instructions per row; only lane 0 / lane 63 of a wave fetch theirs when the neighbour sits in another wave.  The
  // kernel was bound by the number of load instructions, not by bytes (3 per row and thread: 3.4-3.8 TB/s on 112 / 56-pixel maps).
  const bool aligned = (W % (4 * STRIDE) == 0) && A.total % 64 == 0 && ((reinterpret_cast<uintptr_t>(A.x) & 15) == 0);
  const int lane = threadIdx.x & 63;
  const bool need_l = aligned && lane == 0 && gx > 0, need_r = aligned && STRIDE == 1 && lane == 63 && gx + 1 < A.groups_x;
  auto load_row_aligned = [&](int iy, float (&r)[NIN]) {
    const bool ok = iy >= 0 && iy < H;
    const float* row = xp + (size_t)(ok ? iy : 0) * W;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + ix0 + 1);
    f32x4 b = a;
    if (STRIDE == 2) b = *reinterpret_cast<const f32x4*>(row + ix0 + 5);
    float el = 0.f, er = 0.f;
    if (need_l) el = row[ix0];
    if (need_r) er = row[ix0 + 5];
    const float last = STRIDE == 1 ? a.w : b.w;
    const float from_l = __shfl_up(last, 1), from_r = __shfl_down(a.x, 1);
    r[0] = gx == 0 ? 0.f : (lane == 0 ? el : from_l);
    r[1] = a.x, r[2] = a.y, r[3] = a.z, r[4] = a.w;
    if (STRIDE == 1) {
      r[5] = gx + 1 == A.groups_x ? 0.f : (lane == 63 ? er : from_r);
    } else {
      r[5] = b.x, r[6] = b.y, r[7] = b.z, r[8] = b.w;
    }
    if (!ok) {
#pragma unroll
      for (int i = 0; i < NIN; ++i) r[i] = 0.f;
    }
  };
  auto load_row_general = [&](int iy, float (&r)[NIN]) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) r[i] = 0.f;
    if (iy < 0 || iy >= H) return;
    const float* row = xp + (size_t)iy * W;
    // 16-byte loads wherever the 4 columns lie inside the row; gfx950 global memory needs only dword alignment for them
    // (14- and 7-pixel-wide planes are never 16-byte aligned row to row)
    if (ix0 >= 0) r[0] = row[ix0];
    if (ix0 + 4 < W) {
      const f32x4u a = *reinterpret_cast<const f32x4u*>(row + ix0 + 1);
      r[1] = a.x, r[2] = a.y, r[3] = a.z, r[4] = a.w;
    } else {
#pragma unroll
      for (int i = 1; i < 5; ++i)
        if (ix0 + i < W) r[i] = row[ix0 + i];
    }
    if (STRIDE == 1) {
      if (ix0 + 5 < W) r[5] = row[ix0 + 5];
    } else if (ix0 + 8 < W) {
      const f32x4u b = *reinterpret_cast<const f32x4u*>(row + ix0 + 5);
      r[5] = b.x, r[6] = b.y, r[7] = b.z, r[8] = b.w;
    } else {
#pragma unroll
      for (int i = 5; i < NIN; ++i)
        if (ix0 + i < W) r[i] = row[ix0 + i];
    }
  };

  auto load_row = [&](int iy, float (&r)[NIN]) {
    if (aligned) load_row_aligned(iy, r); else load_row_general(iy, r);  // launch-uniform
  };
  // (Requesting the rows of output row oy + 1 before row oy is computed -- two rows in flight per thread -- measured no faster:
  // 0.105 -> 0.120 ms on the 96-channel stride-2 layer; the launch is not bound by one wave's latency chain.)
  float r0[NIN], r1[NIN], r2[NIN];
  load_row(oy0 * STRIDE - 1, r0);
  load_row(oy0 * STRIDE, r1);
  for (int oy = oy0; oy < oy1; ++oy) {
    load_row(oy * STRIDE + 1, r2);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = fmaf(wk[0], r0[j * STRIDE], 0.f);
      acc = fmaf(wk[1], r0[j * STRIDE + 1], acc);
      acc = fmaf(wk[2], r0[j * STRIDE + 2], acc);
      acc = fmaf(wk[3], r1[j * STRIDE], acc);
      acc = fmaf(wk[4], r1[j * STRIDE + 1], acc);
      acc = fmaf(wk[5], r1[j * STRIDE + 2], acc);
      acc = fmaf(wk[6], r2[j * STRIDE], acc);
      acc = fmaf(wk[7], r2[j * STRIDE + 1], acc);
      acc = fmaf(wk[8], r2[j * STRIDE + 2], acc);
      v[j] = epi_norm(acc, ct, A.e);
    }
    const size_t o = (size_t)oy * OW + ox0;
    if (rp) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ox0 + j < OW) v[j] = rp[o + j] + v[j];
    }
    if (A.e.act >= 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (A.e.act == 3) ? epi_act<1>(v[j], cl) : epi_act<2>(v[j], cl);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = epi_act<0>(v[j], cl);
    }
    if (ox0 + 3 < OW) {
      *reinterpret_cast<f32x4u*>(yp + o) = (f32x4u){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ox0 + j < OW) yp[o + j] = v[j];
    }
    if (STRIDE == 1) {
#pragma unroll
      for (int i = 0; i < NIN; ++i) r0[i] = r1[i], r1[i] = r2[i];
    } else {
#pragma unroll
      for (int i = 0; i < NIN; ++i) r0[i] = r2[i];
      load_row(oy * STRIDE + 2, r1);
    }
  }
}

// ---- small maps (28 x 28, 14 x 14, 7 x 7: MobileNet's last 13 depthwise layers).  k_dwpc3x3 gives every thread a few 16-byte
// pieces of 14-pixel rows: a wave's load touches 16 different planes and the layers run at 1.2-1.5 TB/s out of L2.  Planes of
// one tensor are contiguous in memory, so here a workgroup streams PB whole planes into LDS as one flat, fully coalesced copy,
// a thread computes one output row from three input rows held in registers (compile-time width: no per-pixel predicates; the
// zero padding is fed through the same fmaf chain as everywhere), and the outputs leave through LDS as a flat copy again.
struct DwSmallArgs {
  const float* x;
  const float* w;  // [c][3][3]
  float* y;
  Epilogue e;
  long long planes;
  int c, h, oh, pb;  // planes per workgroup
};

template <int W, int STRIDE>
__global__ __launch_bounds__(256) void k_dwpc3x3_small(const DwSmallArgs A) {
  constexpr int OW = (W + 2 - 3) / STRIDE + 1;
  extern __shared__ __attribute__((aligned(16))) float dwl[];
  const int H = A.h, OH = A.oh, HW = H * W, OHW = OH * OW;
  const long long p0 = (long long)blockIdx.x * A.pb;
  const int np = (int)min((long long)A.pb, A.planes - p0);
  float* const in = dwl;                                   // [np][H][W]
  float* const out = dwl + ((A.pb * HW + 3) & ~3);         // [np][OH][OW]
  const int tid = threadIdx.x;
  {  // flat copy in: the block's planes are contiguous (16-byte aligned: pb is a multiple of 4)
    const float* src = A.x + p0 * HW;
    const int total = np * HW, quads = total / 4;
    for (int i = tid; i < quads; i += 256) *reinterpret_cast<f32x4*>(in + 4 * i) = *reinterpret_cast<const f32x4*>(src + 4 * i);
    for (int i = 4 * quads + tid; i < total; i += 256) in[i] = src[i];
  }
  __syncthreads();
  for (int r = tid; r < np * OH; r += 256) {  // one output row per thread
    const int pl = r / OH, oy = r - pl * OH;
    const int ch = (int)((p0 + pl) % A.c);
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = A.w[(size_t)ch * 9 + i];
    const ChannelTerms ct = channel_terms(A.e, ch);
    float rows[3][W + 2];  // columns -1 .. W
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * STRIDE - 1 + ky;
      const bool ok = iy >= 0 && iy < H;
      const float* rp = in + pl * HW + (ok ? iy : 0) * W;
      rows[ky][0] = 0.f, rows[ky][W + 1] = 0.f;
#pragma unroll
      for (int i = 0; i < W; ++i) rows[ky][i + 1] = ok ? rp[i] : 0.f;
    }
    float* op = out + pl * OHW + oy * OW;
    const size_t gbase = (size_t)(p0 + pl) * OHW + (size_t)oy * OW;
#pragma unroll
    for (int ox = 0; ox < OW; ++ox) {
      float acc = fmaf(wk[0], rows[0][ox * STRIDE], 0.f);
      acc = fmaf(wk[1], rows[0][ox * STRIDE + 1], acc);
      acc = fmaf(wk[2], rows[0][ox * STRIDE + 2], acc);
      acc = fmaf(wk[3], rows[1][ox * STRIDE], acc);
      acc = fmaf(wk[4], rows[1][ox * STRIDE + 1], acc);
      acc = fmaf(wk[5], rows[1][ox * STRIDE + 2], acc);
      acc = fmaf(wk[6], rows[2][ox * STRIDE], acc);
      acc = fmaf(wk[7], rows[2][ox * STRIDE + 1], acc);
      acc = fmaf(wk[8], rows[2][ox * STRIDE + 2], acc);
      op[ox] = epi_apply(acc, ct, gbase + ox, A.e);
    }
  }
  __syncthreads();
  {  // flat copy out
    float* dst = A.y + p0 * OHW;
    const int total = np * OHW;
    const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    const int quads = vec ? total / 4 : 0;
    for (int i = tid; i < quads; i += 256) *reinterpret_cast<f32x4*>(dst + 4 * i) = *reinterpret_cast<const f32x4*>(out + 4 * i);
    for (int i = 4 * quads + tid; i < total; i += 256) dst[i] = out[i];
  }
}

template <int W>
static int dw_small_launch(const DwSmallArgs& a, int stride, unsigned blocks, size_t lds, hipStream_t s) {
  if (stride == 1) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_dwpc3x3_small<W, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_dwpc3x3_small<W, 1>), dim3(blocks), dim3(256), lds, s, a);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_dwpc3x3_small<W, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_dwpc3x3_small<W, 2>), dim3(blocks), dim3(256), lds, s, a);
  }
  return check_launchf("k_dwpc3x3_small<%d,s%d>", W, stride);
}

int launch_dwpc3x3(const float* x, const float* w, float* y, int64_t n, int c, int h, int wd, int stride, const Epilogue& e,
                   hipStream_t s) {
  // small square maps: whole planes through LDS (k_dwpc3x3_small)
  if ((wd == 7 || wd == 14 || wd == 28) && h <= 28 && (uintptr_t)x % 16 == 0 && !tune_env("MV_DWPC_NO_SMALL")) {
    DwSmallArgs a = {};
    a.x = x, a.w = w, a.y = y, a.e = e;
    a.planes = (long long)n * c, a.c = c, a.h = h, a.oh = (h + 2 - 3) / stride + 1;
    const int ow = (wd + 2 - 3) / stride + 1;
    // planes per workgroup: ~24 KB of input (4 workgroups per CU with their outputs), a multiple of 4, and at least ~1000 workgroups
    int pb = 6144 / (h * wd);
    while (pb > 4 && a.planes / pb < 1024) pb /= 2;
    pb = (pb + 3) & ~3;
    if (const char* ev = tune_env("MV_DWPC_PB")) pb = atoi(ev) > 0 ? (atoi(ev) + 3) & ~3 : pb;
    a.pb = pb;
    const size_t lds = sizeof(float) * (((size_t)pb * h * wd + 3 & ~(size_t)3) + (size_t)pb * a.oh * ow);
    const long long blocks = (a.planes + pb - 1) / pb;
    if (a.planes == 0) return MV_OK;
    if (lds <= 64 * 1024 && blocks <= 0x7fffffffLL) {
      if (wd == 7) return dw_small_launch<7>(a, stride, (unsigned)blocks, lds, s);
      if (wd == 14) return dw_small_launch<14>(a, stride, (unsigned)blocks, lds, s);
      return dw_small_launch<28>(a, stride, (unsigned)blocks, lds, s);
    }
  }
  DwpcArgs a = {};
  a.x = x, a.w = w, a.y = y, a.e = e;
  a.c = c, a.h = h, a.wd = wd;
  a.oh = (h + 2 - 3) / stride + 1, a.ow = (wd + 2 - 3) / stride + 1;
  a.groups_x = (a.ow + 3) / 4;
  // strip height: tall enough to amortise the two halo rows, short enough to keep >= ~200k threads in flight
  const long long planes = (long long)n * c;
  int rows = a.oh;
  while (rows > 4 && planes * ((a.oh + rows - 1) / rows) * a.groups_x < 200000) rows = (rows + 1) / 2;
  if (rows > 16) rows = 16;
  if (const char* ev = tune_env("MV_DWPC_ROWS")) rows = atoi(ev) > 0 ? atoi(ev) : rows;
  a.rows = rows;
  a.strips = (a.oh + rows - 1) / rows;
  a.total = planes * a.strips * a.groups_x;
  if (a.total > 256LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "depthwise: batch too large for one launch");
  if (a.total == 0) return MV_OK;
  const unsigned nb = (unsigned)((a.total + 255) / 256);
  if (stride == 1)
    hipLaunchKernelGGL((k_dwpc3x3<1>), dim3(nb), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_dwpc3x3<2>), dim3(nb), dim3(256), 0, s, a);
  return check_launch("k_dwpc3x3");
}

// ============================================================================================= stem: dense 3x3, cin <= 4
struct StemArgs {
  const float* x;
  const float* w;  // [cout][cin][3][3]
  float* y;
  Epilogue e;
  long long total;  // n * oh * groups_x
  int cin, cout, h, wd, oh, ow, groups_x;
  int mchunk;       // output channels per blockIdx.y
};

constexpr int kStemMaxChunk = 64;

// thread = 4 consecutive output pixels x `mchunk` output channels: the CIN x 3 x (3 + 3*STRIDE) input window is loaded
// once (16-byte loads) and stays in registers; the channel loop reads its 9*CIN taps from LDS and writes one 16-byte
// store per channel (coalesced along x within the channel plane).  Output-write-bound.
template <int CIN, int STRIDE>
__global__ __launch_bounds__(256) void k_conv3x3_smallcin(const StemArgs A) {
  constexpr int NIN = 3 + 3 * STRIDE;
  constexpr int KW = CIN * 9;
  // the block's taps and channel terms go to LDS once: read from global inside the channel loop, every iteration would
  // wait out a full load round trip behind the previous iteration's store (the compiler cannot prove y does not alias them)
  __shared__ float wsm[kStemMaxChunk * KW];
  __shared__ ChannelTerms tsm[kStemMaxChunk];
  const int m0 = blockIdx.y * A.mchunk, m1 = min(m0 + A.mchunk, A.cout);
  for (int i = threadIdx.x; i < (m1 - m0) * KW; i += 256) wsm[i] = A.w[(size_t)m0 * KW + i];
  for (int i = threadIdx.x; i < m1 - m0; i += 256) tsm[i] = channel_terms(A.e, m0 + i);
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.total) return;
  const int gx = (int)(idx % A.groups_x);
  const long long t = idx / A.groups_x;
  const int oy = (int)(t % A.oh);
  const long long b = t / A.oh;
  const int H = A.h, W = A.wd, OW = A.ow;
  const int ox0 = 4 * gx, ix0 = ox0 * STRIDE - 1, iy0 = oy * STRIDE - 1;
  float xin[CIN][3][NIN];
  if (ix0 + NIN - 1 < W) {
    // the whole window lies inside the row on the right (every thread when W is a multiple of 4 * STRIDE): all 3 * CIN * 3 loads
    // are unconditional -- row and first column clamped into the image, zeros selected afterwards -- so that they are in flight
    // together; under the guards of the general path below each waits for its own round trip (124 -> 72 -> see DESIGN.md 3.8)
    const int cx = max(ix0, 0);
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float* xp = A.x + ((size_t)b * CIN + c) * H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = iy0 + ky;
        const float* row = xp + (size_t)min(max(iy, 0), H - 1) * W;
        float(&r)[NIN] = xin[c][ky];
        r[0] = row[cx];
        const f32x4u a = *reinterpret_cast<const f32x4u*>(row + ix0 + 1);
        r[1] = a.x, r[2] = a.y, r[3] = a.z, r[4] = a.w;
        if (STRIDE == 1) {
          r[5] = row[ix0 + 5];
        } else {
          const f32x4u q = *reinterpret_cast<const f32x4u*>(row + ix0 + 5);
          r[5] = q.x, r[6] = q.y, r[7] = q.z, r[8] = q.w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const bool rowok = iy0 + ky >= 0 && iy0 + ky < H;
#pragma unroll
        for (int i = 0; i < NIN; ++i) xin[c][ky][i] = (rowok && (i > 0 || ix0 >= 0)) ? xin[c][ky][i] : 0.f;
      }
  } else {
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    const float* xp = A.x + ((size_t)b * CIN + c) * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float(&r)[NIN] = xin[c][ky];
#pragma unroll
      for (int i = 0; i < NIN; ++i) r[i] = 0.f;
      const int iy = iy0 + ky;
      if (iy >= 0 && iy < H) {
        const float* row = xp + (size_t)iy * W;
        if (ix0 >= 0) r[0] = row[ix0];
        if (ix0 + 4 < W) {
          const f32x4u a = *reinterpret_cast<const f32x4u*>(row + ix0 + 1);
          r[1] = a.x, r[2] = a.y, r[3] = a.z, r[4] = a.w;
        } else {
#pragma unroll
          for (int i = 1; i < 5; ++i)
            if (ix0 + i < W) r[i] = row[ix0 + i];
        }
        if (STRIDE == 1) {
          if (ix0 + 5 < W) r[5] = row[ix0 + 5];
        } else if (ix0 + 8 < W) {
          const f32x4u q = *reinterpret_cast<const f32x4u*>(row + ix0 + 5);
          r[5] = q.x, r[6] = q.y, r[7] = q.z, r[8] = q.w;
        } else {
#pragma unroll
          for (int i = 5; i < NIN; ++i)
            if (ix0 + i < W) r[i] = row[ix0 + i];
        }
      }
    }
  }
  }
  const Clamp cl = make_clamp(A.e.act);
  const size_t plane = (size_t)A.oh * OW;
  const size_t obase = (size_t)b * A.cout * plane + (size_t)oy * OW + ox0;
#pragma unroll 2
  for (int m = m0; m < m1; ++m) {  // wave-uniform
    const float* wm = wsm + (m - m0) * KW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float wv = wm[(c * 3 + ky) * 3 + kx];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, xin[c][ky][j * STRIDE + kx], acc[j]);
        }
    const ChannelTerms ct = tsm[m - m0];
    const size_t o = obase + (size_t)m * plane;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = epi_norm(acc[j], ct, A.e);
      if (A.e.res && ox0 + j < OW) v[j] = A.e.res[o + j] + v[j];
      v[j] = (A.e.act == 3) ? epi_act<1>(v[j], cl) : ((A.e.act == 4) ? epi_act<2>(v[j], cl) : epi_act<0>(v[j], cl));
    }
    if (ox0 + 3 < OW) {
      *reinterpret_cast<f32x4u*>(A.y + o) = (f32x4u){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ox0 + j < OW) A.y[o + j] = v[j];
    }
  }
}

template <int CIN>
static int stem_launch(const StemArgs& a, int stride, hipStream_t s) {
  const dim3 grid((unsigned)((a.total + 255) / 256), (unsigned)((a.cout + a.mchunk - 1) / a.mchunk));
  if (stride == 1)
    hipLaunchKernelGGL((k_conv3x3_smallcin<CIN, 1>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_conv3x3_smallcin<CIN, 2>), grid, dim3(256), 0, s, a);
  return check_launch("k_conv3x3_smallcin");
}

int launch_conv3x3_smallcin(const float* x, const float* w, float* y, int64_t n, int cin, int h, int wd, int cout, int stride,
                            const Epilogue& e, hipStream_t s) {
  StemArgs a = {};
  a.x = x, a.w = w, a.y = y, a.e = e;
  a.cin = cin, a.cout = cout, a.h = h, a.wd = wd;
  a.oh = (h + 2 - 3) / stride + 1, a.ow = (wd + 2 - 3) / stride + 1;
  a.groups_x = (a.ow + 3) / 4;
  a.total = (long long)n * a.oh * a.groups_x;
  // 16 channels per thread measured best on the 224 -> 112 stem (8: 83 us, 16: 72 us, 32: 77 us at batch 64);
  // fewer when the grid would be too small to fill the chip
  a.mchunk = cout < 16 ? cout : 16;
  while (a.mchunk > 4 && a.total * ((cout + a.mchunk - 1) / a.mchunk) < 150000) a.mchunk = (a.mchunk + 1) / 2;
  if (const char* ev = tune_env("MV_STEM_MCHUNK")) a.mchunk = (atoi(ev) > 0 && atoi(ev) <= kStemMaxChunk) ? atoi(ev) : a.mchunk;
  if (a.total > 256LL * 0x7fffffffLL || (cout + a.mchunk - 1) / a.mchunk > 65535)
    return set_error(MV_ERR_UNSUPPORTED, "conv3x3 (small cin): problem too large for one launch");
  if (a.total == 0) return MV_OK;
  switch (cin) {
    case 1: return stem_launch<1>(a, stride, s);
    case 2: return stem_launch<2>(a, stride, s);
    case 3: return stem_launch<3>(a, stride, s);
    case 4: return stem_launch<4>(a, stride, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "conv3x3 (small cin): cin = %d, 1..4 supported", cin);
}

// ============================================================================================= pointwise 1x1 on MFMA
// Per image: D[p][m] = sum_k X[k][p] * W[m][k]  (pixels are the MFMA rows, channels its columns; X rows are contiguous
// in p: NCHW), one v_mfma_f32_32x32x2_f32 chain per output in ascending k = the oracle's fmaf chain.
//   workgroup  MW*32 output channels x (4/MW)*NT*32 pixels of one image: wave w owns channel tile w % MW and the NT
//              pixel tiles of pixel group w / MW.  MW = 1, 2, 4 for cout <= 32, <= 64, larger -- MobileNet's
//              bottleneck outputs (16..96 channels) would leave 3 of 4 waves multiplying zeros in a fixed 128-channel tile
//   K loop     chunks of 32 channels: W chunk -> LDS [channel][k] (pitch 33: conflict-free operand reads), X chunk ->
//              LDS [k][pixel] (pitch = pixels | 32: the two k rows a wave reads per step fall in different bank halves);
//              the next chunk's global loads are in flight in registers while the 16 k-steps of this one run
//   epilogue   a LANE owns one output channel and its registers 4g..4g+3 are 4 CONSECUTIVE pixels: one set of channel
//              terms per lane, one 16-byte store per 4 outputs.  (With channels as rows every output needed its own
//              address, predicate and 4-byte store and the epilogue cost 2.5x the matrix work.)
#ifndef MV_PK
#define MV_PK 32
#endif
constexpr int kPK = MV_PK;  // channels per K chunk (tools/tune_convnorm.py builds -DMV_PK=64 variants)
constexpr int kWP = kPK + 1;

struct PwArgs {
  const float* x;
  const float* w;  // [cout][cin]
  float* y;
  Epilogue e;
  int cin, cout, hw;
  int chunks, mblocks, ptiles;
  int cps;  // K chunks per K slice (== chunks without slicing)
  int vec_x, vec_w, vec_y;
  long long x_img_stride, y_img_stride;  // floats between consecutive images (cin*hw / cout*hw unless a channel slice)
};


struct PwTile {
  int m, p_base, ntiles, img;  // first channel and first pixel of the wave's tiles; valid pixel tiles; image
};

constexpr int kTP = 36;  // transpose buffer pitch (floats): rows stay 16-byte aligned, b128 accesses spread over banks

// The accumulators hold (pixel rows) x (one channel per lane).  Stored like that, one instruction would write 32 B into
// each of 32 different cache lines.  Each 32x32 tile therefore goes through a wave-private LDS buffer [channel][pixel]
// and comes back with 8 consecutive lanes covering one channel's 32 pixels: every store instruction writes 8 full 128-byte
// lines (16 B per lane), and the residual is read the same way.
// FAST: BatchNorm2d(eval) fold, no conv bias, none / ReLU / ReLU6 -- every block of the MobileNet / ResNet family
template <int NT, int ACTK, bool RES, bool FAST>
__device__ __forceinline__ void pw_store(const f32x16 (&acc)[NT], const PwTile& T, const PwArgs& A, float* tb, int lane) {
  const int M = A.cout, HW = A.hw;
  const Clamp cl = make_clamp(A.e.act);
  const int l31 = lane & 31, hf = lane >> 5;
  const int r0 = lane >> 3, q = (lane & 7) * 4;  // after the transpose: channel rows r0 + 8 j, pixels q .. q + 3
  ChannelTerms c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = channel_terms(A.e, min(T.m + r0 + 8 * j, M - 1));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < T.ntiles) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tb + l31 * kTP + 8 * g + 4 * hf) =
            (f32x4){acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
      const int p = T.p_base + 32 * t + q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = T.m + r0 + 8 * j;
        const f32x4 a = *reinterpret_cast<const f32x4*>(tb + (r0 + 8 * j) * kTP + q);
        if (m < M && p < HW) {
          const size_t row = (size_t)T.img * A.y_img_stride + (size_t)m * HW;
          float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = FAST ? fmaf(v[i], c[j].alpha, c[j].beta) : epi_norm(v[i], c[j], A.e);
          if (A.vec_y && p + 3 < HW) {
            if (RES) {
              const f32x4 r = *reinterpret_cast<const f32x4*>(A.e.res + row + p);
              v[0] = r.x + v[0], v[1] = r.y + v[1], v[2] = r.z + v[2], v[3] = r.w + v[3];
            }
#ifdef MV_ABLATE_STORE
            if (v[0] == 12345.678f)
#endif
            *reinterpret_cast<f32x4*>(A.y + row + p) =
                (f32x4){epi_act<ACTK>(v[0], cl), epi_act<ACTK>(v[1], cl), epi_act<ACTK>(v[2], cl), epi_act<ACTK>(v[3], cl)};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (p + i < HW) {
                if (RES) v[i] = A.e.res[row + p + i] + v[i];
                A.y[row + p + i] = epi_act<ACTK>(v[i], cl);
              }
          }
        }
      }
    }
  }
}

// KS > 1: K SLICES INSIDE the workgroup (MobileNet's 7x7 / 14x14 layers: 256-512 workgroups, K up to 960).  Such a launch is one
// or two workgroups per CU, each walking 12-30 chunks whose loads nothing else on the CU overlaps -- every wave waits half
// its life (PMC).  Here the workgroup has KS x 4 waves: wave group s walks chunks [s * cps, (s + 1) * cps) with
// its own staging buffers, all groups keep the same barrier cadence, and at the end groups 1 .. KS-1 hand their accumulators
// to group 0 through LDS, which adds them IN ASCENDING SLICE ORDER and runs the epilogue.  No atomics, no workspace, no
// second launch; the summation order -- one ascending-k chain per slice from +0, slices added in order -- is stated by
// mv_conv1x1_k_slices() and restated by the oracle (orc_pointwise_sliced_affine_act_f32): bit-exact against that, within
// 1e-6 relative of the single chain.
template <int NT, int MW, int KS>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void k_conv1x1(const PwArgs A) {
  constexpr int PG = 4 / MW;              // pixel groups (waves along pixels)
  constexpr int PXB = PG * NT * 32;       // pixels per workgroup
  constexpr int PITCH = PXB | 32;
  constexpr int XU = PXB / 32 * (kPK / 32);  // float4 per thread for the X chunk: kPK rows x PXB/4
  constexpr int WU = MW * (kPK / 32);        // float4 per thread for the W chunk: MW*32 rows x kPK/4
  constexpr int WQ = kPK / 4;                // float4 per W row
  constexpr int WLS = MW * 32 * kWP > 4 * NT * 1024 ? MW * 32 * kWP : 4 * NT * 1024;  // also a slice's 4 x NT accumulator tiles
  __shared__ __attribute__((aligned(16))) float wl_all[KS][WLS];  // [channel][k]
  constexpr int XS = kPK * PITCH > 4 * 32 * kTP ? kPK * PITCH : 4 * 32 * kTP;  // also the 4 waves' transpose buffers
  __shared__ __attribute__((aligned(16))) float xs_all[KS][XS];   // [k][pixel]
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = KS == 1 ? 0 : wave_all >> 2;  // wave-uniform K slice
  float* const wl = wl_all[slice];
  float* const xs = xs_all[slice];
  const int tid = threadIdx.x & 255, lane = tid & (kWave - 1);   // index inside the slice's 256 threads
  const int wave = wave_all & 3;
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.cin, M = A.cout, HW = A.hw;
  const int mb = blockIdx.x % A.mblocks, pb = blockIdx.x / A.mblocks;
  const int img = blockIdx.y;
  const int mt = wave % MW, pg = wave / MW;
  const int j0 = mb * (MW * 32), p0 = pb * PXB;
  const int pw0 = p0 + pg * (NT * 32);                                      // the wave's first pixel
  const int ntiles = max(0, min(NT, (HW - pw0 + 31) / 32));                 // wave-uniform
  const bool live = (j0 + mt * 32 < M) && ntiles > 0;                       // wave-uniform: anything to compute?
  const float* X = A.x + (size_t)img * A.x_img_stride;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  f32x4 wreg[WU], xreg[XU];
  auto gload = [&](int ch) {
    const int kc = ch * kPK;
#pragma unroll
    for (int u = 0; u < WU; ++u) {  // W chunk: MW*32 rows x kPK/4 float4
      const int idx = tid + 256 * u;
      const int row = idx / WQ, q = idx % WQ;
      const int j = j0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < M) {
        const float* src = A.w + (size_t)j * K + kc + 4 * q;
        if (A.vec_w && kc + 4 * q + 3 < K) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (kc + 4 * q + 0 < K) v.x = src[0];
          if (kc + 4 * q + 1 < K) v.y = src[1];
          if (kc + 4 * q + 2 < K) v.z = src[2];
          if (kc + 4 * q + 3 < K) v.w = src[3];
        }
      }
      wreg[u] = v;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {  // X chunk: 32 channel rows x PXB/4 float4
      const int idx = tid + 256 * u;
      const int row = idx / (PXB / 4), q = idx % (PXB / 4);
      const int k = kc + row, p = p0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#ifdef MV_ABLATE_XLOAD
      if (k < K && HW == 12345) {
#else
      if (k < K) {
#endif
        const float* src = X + (size_t)k * HW + p;
        if (A.vec_x && p + 3 < HW) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (p + 0 < HW) v.x = src[0];
          if (p + 1 < HW) v.y = src[1];
          if (p + 2 < HW) v.z = src[2];
          if (p + 3 < HW) v.w = src[3];
        }
      }
      xreg[u] = v;
    }
  };
  auto lstore = [&]() {
#ifndef MV_ABLATE_WLDS
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / WQ, q = idx % WQ;
      const float e[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) wl[row * kWP + 4 * q + i] = e[i];
    }
#endif
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (PXB / 4), q = idx % (PXB / 4);
      *reinterpret_cast<f32x4*>(xs + row * PITCH + 4 * q) = xreg[u];
    }
  };

  const int ch_first = slice * A.cps;
  const int ch_end = min(ch_first + A.cps, A.chunks);  // this slice's chunks; every slice runs A.cps barrier rounds
  if (ch_first < ch_end) gload(ch_first);
  for (int it = 0; it < A.cps; ++it) {
    const int ch = ch_first + it;
    __syncthreads();  // previous chunk fully consumed
    if (ch < ch_end) lstore();
    __syncthreads();
    if (ch + 1 < ch_end) gload(ch + 1);
    if (live && ch < ch_end) {
      const float* wp = wl + (mt * 32 + l31) * kWP + hf;          // W[channel l31 of my tile][2s + hf]
      const float* xp = xs + hf * PITCH + pg * (NT * 32) + l31;   // X[2s + hf][pixel l31 of my tile t]
      // k-steps that hold real channels (the tail of the last chunk is zero padding: exact no-ops, skipped)
      const int ksteps = min(kPK / 2, (K - ch * kPK + 1) / 2);
      if (ntiles == NT && ksteps == kPK / 2) {
        // full tile, full chunk: straight-line code, so that the operand reads of later k-steps are issued ahead of
        // the MFMAs of earlier ones (the generic loop below waits for an LDS read in front of every MFMA)
        // software pipeline: the operands of k-step s+1 are read (into their own registers) before the MFMAs of
        // k-step s issue; the scheduling barriers keep the compiler from sinking the reads back below them
        float wc = wp[0], xc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) xc[t] = xp[t * 32];
#pragma unroll
        for (int s = 0; s < kPK / 2; ++s) {
          float wn = 0.f, xn[NT];
          if (s + 1 < kPK / 2) {
            wn = wp[2 * (s + 1)];
#pragma unroll
            for (int t = 0; t < NT; ++t) xn[t] = xp[2 * (s + 1) * PITCH + t * 32];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
#ifdef MV_ABLATE_MFMA
            acc[t][s & 15] += wc * xc[t];
#else
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[t], wc, acc[t], 0, 0, 0);  // rows = pixels, columns = channels
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < kPK / 2) {
            wc = wn;
#pragma unroll
            for (int t = 0; t < NT; ++t) xc[t] = xn[t];
          }
        }
      } else {  // partial tile or short last chunk: run-time bounds
        for (int s = 0; s < ksteps; ++s) {
          const float wv = wp[2 * s];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t < ntiles) {
              const float xv = xp[2 * s * PITCH + t * 32];
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, wv, acc[t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // ---- K slices: groups 1 .. KS-1 park their accumulators in their own (now dead) W tile; group 0 adds them in order
  if constexpr (KS > 1) {
    __syncthreads();
    if (slice > 0 && live) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) wl[((wave * NT + t) * 16 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (slice > 0) return;
    if (live) {
#pragma unroll
      for (int sl = 1; sl < KS; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] = acc[t][i] + wl_all[sl][((wave * NT + t) * 16 + i) * 64 + lane];
    }
  }
  // ---- epilogue, instantiated per activation kind / residual so that the per-element code is straight-line; the
  //      operand tiles are dead now: each wave takes 32 x kTP floats of the X tile's space as its transpose buffer
  if constexpr (KS == 1) __syncthreads();  // (with K slices the two barriers above already ended every operand read)
  if (!live) return;
  float* tb = xs + wave * (32 * kTP);
  const PwTile tile = {j0 + mt * 32, pw0, ntiles, img};
  const bool res = A.e.res != nullptr;
#ifdef MV_ABLATE_EPI
  if (acc[0][0] == 12345.678f) A.y[tid] = acc[NT - 1][15] + acc[0][3];
  return;
#endif
  if (A.e.act <= 2 && A.e.affine == 2 && A.e.bias == nullptr) {
    res ? pw_store<NT, 0, true, true>(acc, tile, A, tb, lane) : pw_store<NT, 0, false, true>(acc, tile, A, tb, lane);
  } else if (A.e.act == 3) {
    res ? pw_store<NT, 1, true, false>(acc, tile, A, tb, lane) : pw_store<NT, 1, false, false>(acc, tile, A, tb, lane);
  } else if (A.e.act == 4) {
    res ? pw_store<NT, 2, true, false>(acc, tile, A, tb, lane) : pw_store<NT, 2, false, false>(acc, tile, A, tb, lane);
  } else {
    res ? pw_store<NT, 0, true, false>(acc, tile, A, tb, lane) : pw_store<NT, 0, false, false>(acc, tile, A, tb, lane);
  }
}

// ---- shape -> (channel tiles per workgroup MW, pixel tiles per wave NT, K slices KS): one place, used by the launcher and by
//      conv1x1_plan() (mv_conv1x1_k_slices), so that the stated summation order is the one that runs
struct PwPlan {
  int mw, nt, ks, cps;
};

static PwPlan pw_plan(int64_t n, int cin, int64_t hw, int cout) {
  PwPlan p;
  p.mw = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
  if (const char* e2 = tune_env("MV_PW_MW")) p.mw = atoi(e2) == 1 ? 1 : (atoi(e2) == 2 ? 2 : 4);
  // pixel tiles per wave: fewer when the grid would otherwise leave CUs idle
  const long long wave_tiles = (long long)((cout + 31) / 32) * ((hw + 31) / 32) * n;
  p.nt = wave_tiles <= 8192 ? 1 : ((wave_tiles <= 32768 || p.mw == 1) ? 2 : 4);  // MW = 1, NT = 4 would need 74 KB of LDS
  // long K (the im2col GEMMs of deform_conv2d / AlexNet: K = cin * kh * kw): the W chunk is staged once per workgroup and
  // chunk, so wider pixel tiles amortise it -- as long as the grid still has two workgroups per CU (8 x 256 x 64 x 64 -> 256,
  // K = 2304: NT = 1 0.41 ms, NT = 4 0.35 ms; profiles/r02_perf_deform_conv2d.log)
  if (cin >= 512 && p.mw == 4) {
    const long long mb = (cout + 127) / 128;
    if (mb * ((hw + 127) / 128) * n >= 512) p.nt = 4;
    else if (mb * ((hw + 63) / 64) * n >= 512) p.nt = 2;
  }
  if (const char* e = tune_env("MV_PW_NT")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || (v == 4 && p.mw > 1)) p.nt = v;
  }
  // one K chunk (cin <= 32: MobileNet's expansions 16 -> 96 @ 112 x 112, 24 -> 144 @ 56 x 56 ...): the launch is the epilogue and its
  // stores; the smallest tiles (4 waves x one 32 x 32 tile) give the most workgroups to overlap them -- 206 -> 98 us, 63 -> 45 us,
  // 44 -> 35 us at batch 64 (profiles/r02_ab_pointwise_tiles.log).  X is re-read per 32-channel block, from L2
  const int chunks = (cin + kPK - 1) / kPK;
  if (chunks == 1 && !tune_env("MV_PW_MW") && !tune_env("MV_PW_NT")) p.mw = 1, p.nt = 1;
  // K slices inside the workgroup (NT == 1 only): few workgroups and many chunks -> the launch is latency-bound
  const int pxb = (4 / p.mw) * p.nt * 32;
  const long long workgroups = (long long)((cout + p.mw * 32 - 1) / (p.mw * 32)) * ((hw + pxb - 1) / pxb) * n;
  p.ks = 1;
  if (p.nt == 1 && workgroups <= 640 && chunks >= 6) p.ks = chunks >= 12 ? 4 : 2;
  if (const char* e = tune_env("MV_PW_KS")) p.ks = (p.nt == 1 && (atoi(e) == 2 || atoi(e) == 4)) ? atoi(e) : 1;
  p.cps = (chunks + p.ks - 1) / p.ks;
  return p;
}

void conv1x1_plan(int64_t n, int cin, int64_t hw, int cout, int* slices, int* slice_len) {
  const PwPlan p = pw_plan(n, cin, hw, cout);
  *slices = p.ks;
  *slice_len = p.ks > 1 ? p.cps * kPK : cin;
  if (p.ks > 1) *slices = ((cin + kPK - 1) / kPK + p.cps - 1) / p.cps;  // slices that actually hold chunks
}

template <int NT, int MW>
static int pw_launch(PwArgs& a, int64_t n, const PwPlan& p, hipStream_t s) {
  constexpr int PXB = (4 / MW) * NT * 32;
  a.mblocks = (a.cout + MW * 32 - 1) / (MW * 32);
  a.ptiles = (a.hw + PXB - 1) / PXB;
  a.cps = p.cps;
  const long long nb = (long long)a.mblocks * a.ptiles;
  if (nb > 0x7fffffffLL || n > 65535) return set_error(MV_ERR_UNSUPPORTED, "conv1x1: problem too large for one launch");
  if constexpr (NT == 1) {
    if (p.ks > 1) {
      if (p.ks == 2) hipLaunchKernelGGL((k_conv1x1<1, MW, 2>), dim3((unsigned)nb, (unsigned)n), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((k_conv1x1<1, MW, 4>), dim3((unsigned)nb, (unsigned)n), dim3(1024), 0, s, a);
      return check_launchf("k_conv1x1<1,%d,ks%d>", MW, p.ks);
    }
  }
  hipLaunchKernelGGL((k_conv1x1<NT, MW, 1>), dim3((unsigned)nb, (unsigned)n), dim3(256), 0, s, a);
  return check_launchf("k_conv1x1<%d,%d>", NT, MW);
}

template <int MW>
static int pw_pick_nt(PwArgs& a, int64_t n, const PwPlan& p, hipStream_t s) {
  if (p.nt == 1) return pw_launch<1, MW>(a, n, p, s);
  if (p.nt == 2 || MW == 1) return pw_launch<2, MW>(a, n, p, s);
  return pw_launch<(MW == 1 ? 2 : 4), MW>(a, n, p, s);
}

int launch_conv1x1(const float* x, const float* w, float* y, int64_t n, int cin, int64_t hw, int cout, const Epilogue& e,
                   hipStream_t s, int64_t x_img_stride, int64_t y_img_stride, bool allow_k_slices) {
  if (hw > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "conv1x1: plane too large");
  PwArgs a = {};
  a.x = x, a.w = w, a.y = y, a.e = e;
  a.cin = cin, a.cout = cout, a.hw = (int)hw;
  a.x_img_stride = x_img_stride > 0 ? x_img_stride : (long long)cin * hw;
  a.y_img_stride = y_img_stride > 0 ? y_img_stride : (long long)cout * hw;
  a.chunks = (cin + kPK - 1) / kPK;
  a.vec_w = (cin % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (hw % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.vec_y = (hw % 4 == 0) && ((uintptr_t)y % 16 == 0) && (e.res == nullptr || (uintptr_t)e.res % 16 == 0);
  if (n == 0 || hw == 0) return MV_OK;
  PwPlan p = pw_plan(n, cin, hw, cout);
  if (!allow_k_slices) p.ks = 1, p.cps = a.chunks;  // the im2col GEMMs keep the single chain their oracle states
  if (p.mw == 1) return pw_pick_nt<1>(a, n, p, s);
  if (p.mw == 2) return pw_pick_nt<2>(a, n, p, s);
  return pw_pick_nt<4>(a, n, p, s);
}

}  // namespace mv
