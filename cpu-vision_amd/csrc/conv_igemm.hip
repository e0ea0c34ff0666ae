// conv_igemm.hip -- nn.Conv2d of any kernel size / stride / padding / dilation (one weight group per launch) as an IMPLICIT GEMM
// on v_mfma_f32_32x32x2_f32: mv_conv2d_bias_act_f32 (AlexNet's 11x11 stride 4 and 5x5, models/alexnet.py:22-26; any other
// nn.Conv2d the specialised 3x3 / 1x1 / depthwise kernels do not take).
//
// D[pixel][channel] = sum_k X[pixel][k] * W[channel][k],  k = (c, ky, kx) ascending -- one accumulator per output fed in
// ascending k from +0, then `+ bias`, then the activation: the oracle's chain (orc_conv2d_affine_act_f32), bit for bit, and the
// same bits as the columns form (plain im2col into a workspace + the pointwise GEMM).
//
// Round 3 first put the gather into the pointwise kernel's staging (k_conv1x1<.., IMPL>): the columns left HBM, but that kernel
// was shaped for MobileNet's short K -- PMC on AlexNet's conv2 showed 11 vector instructions per MFMA (profiles/
// r03_pmc_conv2_implicit.txt) and 34-38 TFLOP/s.  This kernel is built for long K:
//   workgroup   64*CT channels x 64*PT pixels of one image; 2 x 2 waves, a wave owns CT x PT MFMA tiles (4-8 accumulators), so
//               every gathered input element feeds 64*CT channels and every staged weight 64*PT pixels;
//   K chunks    of 32: the W chunk (rows of the weight matrix, 16-byte loads) and the X chunk (im2col elements gathered from the
//               input: consecutive lanes = consecutive output pixels; a thread's pixel and its top-left input coordinate are
//               fixed for the whole kernel; k -> (c, ky, kx) on the scalar unit, the k row of an element being wave-uniform)
//               are requested into registers while the previous chunk's MFMAs run, and stored to LDS as [row][k parity][k / 2]:
//               a lane walks ONE parity of k, so four k-steps of an operand are one ds_read_b128 -- (CT + PT) LDS reads per
//               4 * CT * PT MFMAs;
//   epilogue    each 32 x 32 tile is transposed through a wave-private LDS buffer so that a store instruction writes 8 full
//               128-byte lines of 8 channel planes.
// Zero padding of the image = zeros in the X chunk; k past K = zeros in both operands (an exact no-op of the chain).
#include <cstdlib>

#include "mv_common.h"
#include "mv_conv.h"
#include "mv_epilogue.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kIgK = 32;   // k per chunk
constexpr int kIgP = 36;   // floats per operand row: [parity][16] + 4 (pitch / 4 odd: conflict-free 16-byte reads of 16 lanes)

struct IgArgs {
  const float* x;     // input at the group's first channel; images x_img_stride floats apart
  const float* w;     // (mg, K) row-major: the group's weights
  const float* bias;  // [mg] or null
  float* y;           // output at the group's first channel; images y_img_stride floats apart
  long long x_img_stride, y_img_stride;
  int mg, K, HW;
  int ih, iw, ow, kw, taps, sh, sw, ph, pw, dh, dw;
  unsigned m_taps, m_kw, m_ow;  // floor(2^32 / d) + 1 (d == 1: the quotient is the dividend)
  int mblocks, chunks, act, vec_w, vec_y;
};

__device__ __forceinline__ unsigned ig_div(unsigned n, unsigned d, unsigned m) { return d == 1 ? n : __umulhi(n, m); }

template <int CT, int PT>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(const IgArgs A) {
  constexpr int MB = 64 * CT, PB = 64 * PT;    // workgroup tile: channels x pixels
  constexpr int WU = MB / 32;                  // float4 of W per thread and chunk (MB rows x 8 float4)
  constexpr int XU = PB / 8;                   // gathered X elements per thread and chunk (32 k x PB pixels / 256)
  constexpr int XROWS = 256 / PB > 0 ? 256 / PB : 1;  // k rows covered by the 256 threads at once (PB <= 256)
  static_assert(PB == 64 || PB == 128 || PB == 256, "pixel tile");
  __shared__ __attribute__((aligned(16))) float ws[MB * kIgP];
  __shared__ __attribute__((aligned(16))) float xs[PB * kIgP > 4 * 32 * kIgP ? PB * kIgP : 4 * 32 * kIgP];  // also 4 transpose buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int wc = wave & 1, wp = wave >> 1;
  const int mb = blockIdx.x % A.mblocks, pb = blockIdx.x / A.mblocks, img = blockIdx.y;
  const int co0 = mb * MB, p0 = pb * PB;
  const int K = A.K, HW = A.HW;
  const float* X = A.x + (size_t)img * A.x_img_stride;

  // ---- this thread's pixel: fixed for the whole kernel
  const int px_in_tile = tid % PB;
  const int row_base = __builtin_amdgcn_readfirstlane(tid / PB);  // k row of the thread's first element: wave-uniform (PB >= 64)
  const int p = p0 + px_in_tile;
  const bool p_ok = p < HW;
  const unsigned oy = ig_div((unsigned)min(p, HW - 1), (unsigned)A.ow, A.m_ow);
  const int ox = min(p, HW - 1) - (int)oy * A.ow;
  const int iy0 = (int)oy * A.sh - A.ph, ix0 = ox * A.sw - A.pw;
  const int off0 = iy0 * A.iw + ix0;

  f32x4 wr[WU];
  float xr[XU];
  auto gload = [&](int ch) {
    const int kc = ch * kIgK;
#pragma unroll
    for (int u = 0; u < WU; ++u) {  // row = idx / 8, k = kc + 4 * (idx % 8) .. + 3
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const int co = co0 + row, k = kc + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (co < A.mg) {
        const float* src = A.w + (size_t)co * K + k;
        if (A.vec_w && k + 3 < K) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (k + 0 < K) v.x = src[0];
          if (k + 1 < K) v.y = src[1];
          if (k + 2 < K) v.z = src[2];
          if (k + 3 < K) v.w = src[3];
        }
      }
      wr[u] = v;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const unsigned k = (unsigned)(kc + row_base + XROWS * u);  // wave-uniform: scalar arithmetic down to row_off
      const unsigned c = ig_div(k, (unsigned)A.taps, A.m_taps);
      const unsigned r = k - c * A.taps;
      const unsigned ky = ig_div(r, (unsigned)A.kw, A.m_kw);
      const unsigned kx = r - ky * A.kw;
      const int dy = (int)ky * A.dh, dx = (int)kx * A.dw;
      const int row_off = ((int)c * A.ih + dy) * A.iw + dx;
      const int iy = iy0 + dy, ix = ix0 + dx;
      const bool ok = p_ok && (int)k < K && (unsigned)iy < (unsigned)A.ih && (unsigned)ix < (unsigned)A.iw;
      xr[u] = ok ? X[off0 + row_off] : 0.f;  // outside the image: the zero padding; k past K: an exact no-op
    }
  };
  auto lstore = [&]() {  // k = 4 q .. 4 q + 3 of a W row -> parities 0, 1, 0, 1 at s = 2 q, 2 q, 2 q + 1, 2 q + 1
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      float* d = ws + row * kIgP + 2 * q;
      *reinterpret_cast<f32x2*>(d) = (f32x2){wr[u].x, wr[u].z};
      *reinterpret_cast<f32x2*>(d + 16) = (f32x2){wr[u].y, wr[u].w};
    }
    float* xd = xs + px_in_tile * kIgP;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int kk = row_base + XROWS * u;  // 0 .. 31; wave-uniform
      xd[(kk & 1) * 16 + (kk >> 1)] = xr[u];
    }
  };

  f32x16 acc[PT][CT];
#pragma unroll
  for (int j = 0; j < PT; ++j)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  gload(0);
  for (int ch = 0; ch < A.chunks; ++ch) {
    __syncthreads();  // the previous chunk's operand reads are done
    lstore();
    __syncthreads();
    if (ch + 1 < A.chunks) gload(ch + 1);
    const float* ap = xs + ((wp * PT) * 32 + l31) * kIgP + hf * 16;  // + j * 32 rows
    const float* bp = ws + ((wc * CT) * 32 + l31) * kIgP + hf * 16;  // + i * 32 rows
    f32x4 aq[2][PT], bq[2][CT];
#pragma unroll
    for (int j = 0; j < PT; ++j) aq[0][j] = *reinterpret_cast<const f32x4*>(ap + j * 32 * kIgP);
#pragma unroll
    for (int i = 0; i < CT; ++i) bq[0][i] = *reinterpret_cast<const f32x4*>(bp + i * 32 * kIgP);
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // 4 groups of 4 k-steps; the next group's operands are read before this group's MFMAs issue
      if (g + 1 < 4) {
#pragma unroll
        for (int j = 0; j < PT; ++j) aq[(g + 1) & 1][j] = *reinterpret_cast<const f32x4*>(ap + j * 32 * kIgP + 4 * (g + 1));
#pragma unroll
        for (int i = 0; i < CT; ++i) bq[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(bp + i * 32 * kIgP + 4 * (g + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int i = 0; i < CT; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g & 1][j][e], bq[g & 1][i][e], acc[j][i], 0, 0, 0);  // rows = pixels, columns = channels
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: + bias, activation; every tile through a wave-private [channel][pixel] buffer, then 128-byte lines per channel
  __syncthreads();
  float* const tb = xs + wave * (32 * kIgP);
  const int r0 = lane >> 3, q4 = (lane & 7) * 4;
  const Clamp cl = make_clamp(A.act);
  float* const Y = A.y + (size_t)img * A.y_img_stride;
  // this lane's 4 CT bias values in ONE batch before the tiles: loaded where they are used, each sat in its own branch with its own
  // `s_waitcnt vmcnt(0)` -- CT * PT * 4 memory round trips in a row at the end of every workgroup
  float biasv[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int co = co0 + (wc * CT + i) * 32 + r0 + 8 * jj;
      biasv[i][jj] = A.bias ? A.bias[min(co, A.mg - 1)] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int cbase = co0 + (wc * CT + i) * 32;
    if (cbase >= A.mg) continue;  // wave-uniform
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int pbase = p0 + (wp * PT + j) * 32;
      if (pbase >= HW) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tb + l31 * kIgP + 8 * g + 4 * hf) =
            (f32x4){acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
      const int pq = pbase + q4;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int co = cbase + r0 + 8 * jj;
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tb + (r0 + 8 * jj) * kIgP + q4);
        if (co >= A.mg || pq >= HW) continue;
        const float bv = biasv[i][jj];
        float v[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (A.bias) v[r] = v[r] + bv;
          v[r] = (A.act == 3) ? epi_act<1>(v[r], cl) : ((A.act == 4) ? epi_act<2>(v[r], cl) : epi_act<0>(v[r], cl));
        }
        float* dst = Y + (size_t)co * HW + pq;
        if (A.vec_y && pq + 3 < HW) {
          *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (pq + r < HW) dst[r] = v[r];
        }
      }
    }
  }
}

template <int CT, int PT>
static int ig_launch(IgArgs& a, int64_t n, hipStream_t s) {
  a.mblocks = (a.mg + 64 * CT - 1) / (64 * CT);
  const long long pblocks = (a.HW + 64 * PT - 1) / (64 * PT);
  const long long nb = a.mblocks * pblocks;
  if (nb > 0x7fffffffLL || n > 65535) return set_error(MV_ERR_UNSUPPORTED, "conv2d (implicit GEMM): problem too large for one launch");
  hipLaunchKernelGGL((k_conv_igemm<CT, PT>), dim3((unsigned)nb, (unsigned)n), dim3(256), 0, s, a);
  return check_launchf("k_conv_igemm<%dch,%dpx,implicit %dx%d>", 64 * CT, 64 * PT, a.taps / a.kw, a.kw);
}

// Workgroups the implicit kernel would launch with its smallest tiles (64 channels x 64 pixels): a launch of fewer than 128 is a
// few workgroups walking all of K in sequence (AlexNet's conv2 at batch 1: 36 workgroups x 50 chunks) -- the columns form cuts
// smaller tiles and is faster there, so mv_conv2d_needs_workspace() answers 2 (optional) and the caller may bring the workspace.
long long conv2d_implicit_min_workgroups(int64_t n, int mg, int oh, int ow) {
  return (long long)n * ((mg + 63) / 64) * (((long long)oh * ow + 63) / 64);
}

bool conv2d_implicit_supported(int cg, int kh, int kw, int oh, int ow) {
  const long long K = (long long)cg * kh * kw, HW = (long long)oh * ow;
  return K < 65536 && HW < (1 << 20) && ow < 4096 && kh * kw < 4096 && !tune_env("MV_CONV_COLUMNS");
}

int launch_conv2d_implicit(const float* x, const float* w, float* y, int64_t n, int cg, int h, int wd, int mg, int kh, int kw, int sh,
                           int sw, int ph, int pw_, int dh, int dw, int oh, int ow, const Epilogue& e, hipStream_t s,
                           int64_t x_img_stride, int64_t y_img_stride) {
  if ((long long)cg * h * wd >= (1LL << 30)) return set_error(MV_ERR_UNSUPPORTED, "conv2d (implicit GEMM): one image of a group has 2^30 elements or more");
  if (e.affine != 0 || e.res != nullptr) return set_error(MV_ERR_UNSUPPORTED, "conv2d (implicit GEMM): bias + activation only");
  IgArgs a = {};
  a.x = x, a.w = w, a.bias = e.bias, a.y = y;
  a.x_img_stride = x_img_stride, a.y_img_stride = y_img_stride;
  a.mg = mg, a.K = cg * kh * kw, a.HW = oh * ow;
  a.ih = h, a.iw = wd, a.ow = ow, a.kw = kw, a.taps = kh * kw, a.sh = sh, a.sw = sw, a.ph = ph, a.pw = pw_, a.dh = dh, a.dw = dw;
  auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; };
  a.m_taps = magic((unsigned)a.taps), a.m_kw = magic((unsigned)kw), a.m_ow = magic((unsigned)ow);
  a.chunks = (a.K + kIgK - 1) / kIgK;
  a.act = e.act;
  a.vec_w = (a.K % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_y = (a.HW % 4 == 0) && ((uintptr_t)y % 16 == 0) && (y_img_stride % 4 == 0);
  if (n == 0 || a.HW == 0) return MV_OK;
  // channel tile: 64, 128 or 192 channels per workgroup, whichever pads the least (ties: the larger -- every gathered element
  // then feeds more channels); pixel tile: 256 pixels with one channel tile per wave, else 128; small problems take 64-pixel
  // tiles so that the grid still covers the chip
  auto padded = [](long long v, long long t) { return (v + t - 1) / t * t; };
  int ct = 3;
  long long best = padded(mg, 192);
  if (padded(mg, 128) < best) ct = 2, best = padded(mg, 128);
  if (padded(mg, 64) < best) ct = 1, best = padded(mg, 64);
  if (const char* ev = tune_env("MV_IG_CT")) ct = atoi(ev) >= 1 && atoi(ev) <= 3 ? atoi(ev) : ct;
  const long long wgs256 = padded(mg, 64 * ct) / (64 * ct) * ((a.HW + 255) / 256) * n;
  if (ct == 1) {
    if (wgs256 >= 512 || a.HW > 4096) return ig_launch<1, 4>(a, n, s);
    if (padded(mg, 64) / 64 * ((a.HW + 127) / 128) * n >= 256) return ig_launch<1, 2>(a, n, s);
    return ig_launch<1, 1>(a, n, s);
  }
  if (ct == 2) {
    if (padded(mg, 128) / 128 * ((a.HW + 127) / 128) * n >= 256) return ig_launch<2, 2>(a, n, s);
    return ig_launch<2, 1>(a, n, s);
  }
  if (padded(mg, 192) / 192 * ((a.HW + 127) / 128) * n >= 256) return ig_launch<3, 2>(a, n, s);
  return ig_launch<3, 1>(a, n, s);
}

}  // namespace mv
