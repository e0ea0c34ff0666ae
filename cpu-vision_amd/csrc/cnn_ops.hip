// cnn_ops.hip -- the non-conv layers of the small CNNs' feature extractor:
//   nn.MaxPool2d(kernel_size=2, stride=2)   models/vgg.py:78-79      (HBM-bound: 4 B read + 1 B... 5 B per input element)
//   nn.AdaptiveAvgPool2d((7, 7))            models/vgg.py:41         (identity on the 7x7 map of a 224 input)
#include "mv_common.h"

namespace mv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float max_nan(float a, float b) { return (b > a || b != b) ? b : a; }  // NaN propagates (ATen)

// one lane -> 4 consecutive outputs of one output row: 2 x 32 B in, 16 B out
template <bool VEC>
__global__ __launch_bounds__(256) void k_maxpool2x2(const float* __restrict__ x, float* __restrict__ y, long long planes,
                                                     int h, int w, int oh, int ow, int qpr /* quads per out row */) {
  const long long total = planes * oh * qpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % qpr);
    const long long t = i / qpr;
    const int oy = (int)(t % oh);
    const long long p = t / oh;
    const float* r0 = x + ((size_t)p * h + 2 * oy) * w + 8 * q;
    const float* r1 = r0 + w;
    float* dst = y + ((size_t)p * oh + oy) * ow + 4 * q;
    if (VEC) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(r1), b1 = *reinterpret_cast<const f32x4*>(r1 + 4);
      f32x4 o;
      // window order of the oracle / ATen: (0,0), (0,1), (1,0), (1,1)
      o.x = max_nan(max_nan(max_nan(a0.x, a0.y), b0.x), b0.y);
      o.y = max_nan(max_nan(max_nan(a0.z, a0.w), b0.z), b0.w);
      o.z = max_nan(max_nan(max_nan(a1.x, a1.y), b1.x), b1.y);
      o.w = max_nan(max_nan(max_nan(a1.z, a1.w), b1.z), b1.w);
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < ow) {
          const int c = 8 * q + 2 * j;
          dst[j] = max_nan(max_nan(max_nan(r0[c - 8 * q], r0[c - 8 * q + 1]), r1[c - 8 * q]), r1[c - 8 * q + 1]);
        }
    }
  }
}

int launch_maxpool2x2(const float* x, float* y, int64_t planes, int h, int w, hipStream_t s) {
  const int oh = h / 2, ow = w / 2;
  if (oh == 0 || ow == 0 || planes == 0) return MV_OK;
  const int qpr = (ow + 3) / 4;
  const bool vec = (w % 8 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  const long long total = planes * oh * qpr;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
  if (vec)
    hipLaunchKernelGGL(k_maxpool2x2<true>, dim3(blocks), dim3(256), 0, s, x, y, (long long)planes, h, w, oh, ow, qpr);
  else
    hipLaunchKernelGGL(k_maxpool2x2<false>, dim3(blocks), dim3(256), 0, s, x, y, (long long)planes, h, w, oh, ow, qpr);
  return check_launch("k_maxpool2x2");
}

// one thread per output; windows [floor(i*h/oh), ceil((i+1)*h/oh)), row-major fp32 sum / count (oracle order)
__global__ __launch_bounds__(256) void k_adaptive_avgpool(const float* __restrict__ x, float* __restrict__ y,
                                                           long long planes, int h, int w, int oh, int ow) {
  const long long total = planes * oh * ow;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const long long t = i / ow;
    const int oy = (int)(t % oh);
    const long long p = t / oh;
    const int y0 = (oy * h) / oh, y1 = ((oy + 1) * h + oh - 1) / oh;
    const int x0 = (ox * w) / ow, x1 = ((ox + 1) * w + ow - 1) / ow;
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) acc += x[((size_t)p * h + yy) * w + xx];
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
  }
}

// Global average (output 1 x 1) of small planes -- MobileNetV2's 7 x 7 head (mobilenetv2.py:160), 1280 x batch planes of 49
// floats: one thread per plane as above, but the plane's <= 64 elements are LOADED FIRST (clamped indices: every load
// unconditional and in flight together), then summed in the oracle's row-major order.  The generic loop's loads sit under
// run-time bounds and go one round trip at a time: 21 us for 64 x 1280 planes, this form 4-5 us.
__global__ __launch_bounds__(256) void k_global_avgpool_small(const float* __restrict__ x, float* __restrict__ y, long long planes, int n) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= planes) return;
  const float* xp = x + (size_t)p * n;
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = xp[min(i, n - 1)];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = i < n ? acc + v[i] : acc;
  y[p] = acc / (float)n;
}

int launch_adaptive_avgpool(const float* x, float* y, int64_t planes, int h, int w, int oh, int ow, hipStream_t s) {
  const long long total = (long long)planes * oh * ow;
  if (total == 0) return MV_OK;
  if (oh == 1 && ow == 1 && h * w <= 64 && (planes + 255) / 256 <= 0x7fffffffLL) {
    hipLaunchKernelGGL(k_global_avgpool_small, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, s, x, y, (long long)planes, h * w);
    return check_launch("k_global_avgpool_small");
  }
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_adaptive_avgpool, dim3(blocks), dim3(256), 0, s, x, y, (long long)planes, h, w, oh, ow);
  return check_launch("k_adaptive_avgpool");
}


// ---- preset tail: ToDtype(float32, scale=True) [+ Normalize(mean, std)], one pass --------------
// to_dtype_image: image.to(float32).mul_(1/255) (transforms/v2/functional/_misc.py:286-288);
// normalize_image: image.sub(mean).div_(std)   (_misc.py:54-66).  HBM-bound: 1 B (or 4 B) read + 4 B written.
struct NormArgs {
  const void* x;
  float* y;
  long long hw;
  int c;
  int chunks;  // 4096-element chunks per plane
  int normalize;
  float mean[16], stdv[16];
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_to_float_normalize(const NormArgs A) {
  const long long plane = blockIdx.x / A.chunks;
  const int chunk = blockIdx.x % A.chunks;
  const int ch = (int)(plane % A.c);
  const float m = A.normalize ? A.mean[ch] : 0.f, sd = A.normalize ? A.stdv[ch] : 1.f;
  const float scale = (float)(1.0 / 255.0);
  const T* xp = static_cast<const T*>(A.x) + plane * A.hw;
  float* yp = A.y + plane * A.hw;
  // a lane takes 4 consecutive elements in each of 4 steps 1024 elements apart: every store instruction of the wave
  // writes 1 KiB contiguous (a lane owning 16 consecutive elements would write 16 B out of every 64 B per store)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long e0 = (long long)chunk * 4096 + j * 1024 + threadIdx.x * 4;
    float v[4];
    if (VEC) {
      if (e0 >= A.hw) return;
      if constexpr (sizeof(T) == 1) {
        const unsigned q = *reinterpret_cast<const unsigned*>(xp + e0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)((q >> (8 * i)) & 0xffu) * scale;
      } else {
        const f32x4 q = *reinterpret_cast<const f32x4*>(xp + e0);
        v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
      }
      if (A.normalize) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (v[i] - m) / sd;
      }
      __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(yp + e0));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long e = e0 + i;
        if (e < A.hw) {
          float t = (sizeof(T) == 1) ? (float)xp[e] * scale : (float)xp[e];
          if (A.normalize) t = (t - m) / sd;
          yp[e] = t;
        }
      }
    }
  }
}

int launch_to_float_normalize(const void* x, float* y, bool u8, int64_t n, int c, int64_t hw, const float* mean,
                              const float* stdv, hipStream_t s) {
  if (c > 16) return set_error(MV_ERR_UNSUPPORTED, "normalize: up to 16 channels, got %d", c);
  NormArgs a = {};
  a.x = x, a.y = y, a.hw = hw, a.c = c;
  a.chunks = (int)((hw + 4095) / 4096);
  a.normalize = (mean != nullptr && stdv != nullptr);
  for (int i = 0; i < c; ++i) a.mean[i] = a.normalize ? mean[i] : 0.f, a.stdv[i] = a.normalize ? stdv[i] : 1.f;
  const long long nb = (long long)n * c * a.chunks;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "normalize: batch too large for one launch");
  if (nb == 0) return MV_OK;
  const bool vec = (hw % 4 == 0) && ((uintptr_t)x % (u8 ? 4 : 16) == 0) && ((uintptr_t)y % 16 == 0);
  dim3 grid((unsigned)nb), block(256);
  if (u8) {
    if (vec) hipLaunchKernelGGL((k_to_float_normalize<uint8_t, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_to_float_normalize<uint8_t, false>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((k_to_float_normalize<float, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_to_float_normalize<float, false>), grid, block, 0, s, a);
  }
  return check_launch("k_to_float_normalize");
}

// nn.MaxPool2d(kernel_size=k, stride=s), no padding, floor mode (AlexNet's 3x3 stride 2, models/alexnet.py:24,27,34):
// thread per output, window read with bounds (always inside for floor mode); NaN propagates like ATen's (a > m || a != a)
__global__ __launch_bounds__(256) void k_maxpool2d(const float* x, float* y, long long total, int h, int w, int oh, int ow, int k,
                                                    int stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % ow);
  const long long t = idx / ow;
  const int oy = (int)(t % oh);
  const long long plane = t / oh;
  const float* xp = x + (size_t)plane * h * w + (size_t)(oy * stride) * w + ox * stride;
  float m = xp[0];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      const float a = xp[(size_t)i * w + j];
      if (a > m || a != a) m = a;
    }
  y[idx] = m;
}

int launch_maxpool2d(const float* x, float* y, int64_t planes, int h, int w, int k, int stride, hipStream_t s) {
  const int oh = (h - k) / stride + 1, ow = (w - k) / stride + 1;
  const long long total = (long long)planes * oh * ow;
  if (total > 256LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "maxpool2d: batch too large for one launch");
  if (total == 0) return MV_OK;
  hipLaunchKernelGGL(k_maxpool2d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, total, h, w, oh, ow, k, stride);
  return check_launch("k_maxpool2d");
}

}  // namespace mv
