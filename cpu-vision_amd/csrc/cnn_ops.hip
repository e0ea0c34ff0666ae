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

int launch_adaptive_avgpool(const float* x, float* y, int64_t planes, int h, int w, int oh, int ow, hipStream_t s) {
  const long long total = (long long)planes * oh * ow;
  if (total == 0) return MV_OK;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_adaptive_avgpool, dim3(blocks), dim3(256), 0, s, x, y, (long long)planes, h, w, oh, ow);
  return check_launch("k_adaptive_avgpool");
}

}  // namespace mv
