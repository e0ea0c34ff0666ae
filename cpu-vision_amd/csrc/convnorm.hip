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
#include "mv_common.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float epi_act(float v, int act) {
  switch (act) {
    case 1: return v < 0.f ? 0.f : v;
    case 2: return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    case 3: {
      float t = v + 3.f;
      t = t < 0.f ? 0.f : (t > 6.f ? 6.f : t);
      return v * t / 6.f;
    }
    case 4: return v / (1.f + expf(-v));
    default: return v;
  }
}

__device__ inline float epi_apply(float acc, int m, size_t out_index, const Epilogue& e) {
  if (e.bias) acc = acc + e.bias[m];
  if (e.affine == 1) {
    acc = acc * e.alpha[m];
    acc = acc + e.beta[m];
  } else if (e.affine == 2) {
    acc = fmaf(acc, e.alpha[m], e.beta[m]);
  }
  if (e.res) acc = e.res[out_index] + acc;
  return epi_act(acc, e.act);
}

// ============================================================================================= depthwise, per channel
struct DwpcArgs {
  const float* x;
  const float* w;  // [c][3][3]
  float* y;
  Epilogue e;
  long long total;  // n * c * oh * ow
  int c, h, wd, oh, ow;
};

template <int STRIDE>
__global__ __launch_bounds__(256) void k_dwpc3x3(const DwpcArgs A) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= A.total) return;
  const int ox = (int)(idx % A.ow);
  const long long t = idx / A.ow;
  const int oy = (int)(t % A.oh);
  const long long plane = t / A.oh;
  const int ch = (int)(plane % A.c);
  const float* xp = A.x + (size_t)plane * A.h * A.wd;
  const float* wp = A.w + (size_t)ch * 9;
  const int iy0 = oy * STRIDE - 1, ix0 = ox * STRIDE - 1;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = iy0 + ky;
    const bool rok = iy >= 0 && iy < A.h;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ix0 + kx;
      const float v = (rok && ix >= 0 && ix < A.wd) ? xp[(size_t)iy * A.wd + ix] : 0.f;
      acc = fmaf(wp[ky * 3 + kx], v, acc);
    }
  }
  A.y[idx] = epi_apply(acc, ch, (size_t)idx, A.e);
}

int launch_dwpc3x3(const float* x, const float* w, float* y, int64_t n, int c, int h, int wd, int stride, const Epilogue& e,
                   hipStream_t s) {
  DwpcArgs a = {};
  a.x = x, a.w = w, a.y = y, a.e = e;
  a.c = c, a.h = h, a.wd = wd;
  a.oh = (h + 2 - 3) / stride + 1, a.ow = (wd + 2 - 3) / stride + 1;
  a.total = (long long)n * c * a.oh * a.ow;
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
  long long pixels;  // n * oh * ow
  int cin, cout, h, wd, oh, ow;
  int mchunks;       // ceil(cout / 8)
};

// thread = one output pixel x 8 output channels (blockIdx.y picks the channel chunk): the CIN*9 inputs stay in
// registers, the weights are wave-uniform (scalar loads), stores are coalesced along x per channel plane.
template <int CIN, int STRIDE>
__global__ __launch_bounds__(256) void k_conv3x3_smallcin(const StemArgs A) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= A.pixels) return;
  const int ox = (int)(pix % A.ow);
  const long long t = pix / A.ow;
  const int oy = (int)(t % A.oh);
  const long long b = t / A.oh;
  const int iy0 = oy * STRIDE - 1, ix0 = ox * STRIDE - 1;
  float xin[CIN * 9];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    const float* xp = A.x + ((size_t)b * CIN + c) * A.h * A.wd;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = iy0 + ky;
      const bool rok = iy >= 0 && iy < A.h;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ix0 + kx;
        xin[(c * 3 + ky) * 3 + kx] = (rok && ix >= 0 && ix < A.wd) ? xp[(size_t)iy * A.wd + ix] : 0.f;
      }
    }
  }
  const int m0 = blockIdx.y * 8;
  const size_t plane = (size_t)A.oh * A.ow;
  const size_t obase = (size_t)b * A.cout * plane + (size_t)oy * A.ow + ox;
#pragma unroll
  for (int mm = 0; mm < 8; ++mm) {
    const int m = m0 + mm;  // wave-uniform
    if (m < A.cout) {
      const float* wm = A.w + (size_t)m * CIN * 9;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < CIN * 9; ++i) acc = fmaf(wm[i], xin[i], acc);
      const size_t oi = obase + (size_t)m * plane;
      A.y[oi] = epi_apply(acc, m, oi, A.e);
    }
  }
}

template <int CIN>
static int stem_launch(const StemArgs& a, int stride, hipStream_t s) {
  const dim3 grid((unsigned)((a.pixels + 255) / 256), (unsigned)a.mchunks);
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
  a.pixels = (long long)n * a.oh * a.ow;
  a.mchunks = (cout + 7) / 8;
  if (a.pixels > 256LL * 0x7fffffffLL || a.mchunks > 65535) return set_error(MV_ERR_UNSUPPORTED, "conv3x3 (small cin): problem too large for one launch");
  if (a.pixels == 0) return MV_OK;
  switch (cin) {
    case 1: return stem_launch<1>(a, stride, s);
    case 2: return stem_launch<2>(a, stride, s);
    case 3: return stem_launch<3>(a, stride, s);
    case 4: return stem_launch<4>(a, stride, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "conv3x3 (small cin): cin = %d, 1..4 supported", cin);
}

// ============================================================================================= pointwise 1x1 on MFMA
// Per image: D[m][p] = sum_k W[m][k] * X[k][p], M = cout, K = cin, p = pixel (X rows are contiguous in p: NCHW).
//   workgroup  128 output channels x NT*32 pixels of one image; wave w owns channel tile w and all NT pixel tiles
//   K loop     chunks of 32 channels: W chunk -> LDS in MFMA fragment order, X chunk -> LDS [k][pixel] (pitch
//              NT*32 | 32: the two k rows a wave reads per step fall in different bank halves); the next chunk's
//              global loads are in flight in registers while the 16 k-steps of this one run.
constexpr int kPK = 32;

struct PwArgs {
  const float* x;
  const float* w;  // [cout][cin]
  float* y;
  Epilogue e;
  int cin, cout, hw;
  int chunks, mblocks, ptiles;
  int vec_x, vec_w;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void k_conv1x1(const PwArgs A) {
  constexpr int PITCH = (NT * 32) | 32;
  __shared__ __attribute__((aligned(16))) float wfr[16 * 4 * 64];  // [s][mt][lane]
  __shared__ __attribute__((aligned(16))) float xs[kPK * PITCH];   // [k][pixel]
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.cin, M = A.cout, HW = A.hw;
  const int mb = blockIdx.x % A.mblocks, pb = blockIdx.x / A.mblocks;
  const int img = blockIdx.y;
  const int j0 = mb * 128, p0 = pb * (NT * 32);
  const int ntiles = min(NT, (HW - p0 + 31) / 32);  // wave-uniform
  const float* X = A.x + (size_t)img * K * HW;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  f32x4 wreg[4], xreg[NT];
  auto gload = [&](int ch) {
    const int kc = ch * kPK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // W chunk: 128 rows x 8 float4
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
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
    for (int u = 0; u < NT; ++u) {  // X chunk: 32 channel rows x NT*8 float4
      const int idx = tid + 256 * u;
      const int row = idx / (NT * 8), q = idx % (NT * 8);
      const int k = kc + row, p = p0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < K) {
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
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const float e[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kl = 4 * q + i;
        wfr[(((kl >> 1) * 4 + (row >> 5)) << 6) + (kl & 1) * 32 + (row & 31)] = e[i];
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (NT * 8), q = idx % (NT * 8);
      *reinterpret_cast<f32x4*>(xs + row * PITCH + 4 * q) = xreg[u];
    }
  };

  gload(0);
  for (int ch = 0; ch < A.chunks; ++ch) {
    __syncthreads();  // previous chunk fully consumed
    lstore();
    __syncthreads();
    if (ch + 1 < A.chunks) gload(ch + 1);
    const float* ap = wfr + wave * 64 + lane;
    const float* bp = xs + hf * PITCH + l31;
#pragma unroll 8
    for (int s = 0; s < kPK / 2; ++s) {
      const float av = ap[s * 256];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < ntiles) {
          const float bv = bp[2 * s * PITCH + t * 32];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: register i <-> channel (i & 3) + 8 (i >> 2) + 4 hf of the wave's tile, lane <-> pixel
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < ntiles) {
      const int p = p0 + 32 * t + l31;
      if (p < HW) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = j0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * hf;
          if (m < M) {
            const size_t oi = ((size_t)img * M + m) * HW + p;
            A.y[oi] = epi_apply(acc[t][i], m, oi, A.e);
          }
        }
      }
    }
  }
}

template <int NT>
static int pw_launch(PwArgs& a, int64_t n, hipStream_t s) {
  a.ptiles = (a.hw + NT * 32 - 1) / (NT * 32);
  const long long nb = (long long)a.mblocks * a.ptiles;
  if (nb > 0x7fffffffLL || n > 65535) return set_error(MV_ERR_UNSUPPORTED, "conv1x1: problem too large for one launch");
  hipLaunchKernelGGL((k_conv1x1<NT>), dim3((unsigned)nb, (unsigned)n), dim3(256), 0, s, a);
  return check_launch("k_conv1x1");
}

int launch_conv1x1(const float* x, const float* w, float* y, int64_t n, int cin, int64_t hw, int cout, const Epilogue& e,
                   hipStream_t s) {
  if (hw > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "conv1x1: plane too large");
  PwArgs a = {};
  a.x = x, a.w = w, a.y = y, a.e = e;
  a.cin = cin, a.cout = cout, a.hw = (int)hw;
  a.chunks = (cin + kPK - 1) / kPK;
  a.mblocks = (cout + 127) / 128;
  a.vec_w = (cin % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (hw % 4 == 0) && ((uintptr_t)x % 16 == 0);
  if (n == 0 || hw == 0) return MV_OK;
  // pixel tiles per workgroup: fewer when the grid would otherwise leave CUs idle
  const long long tiles = (long long)a.mblocks * ((hw + 31) / 32) * n;
  if (tiles <= 2048) return pw_launch<1>(a, n, s);
  if (tiles <= 8192) return pw_launch<2>(a, n, s);
  return pw_launch<4>(a, n, s);
}

}  // namespace mv
