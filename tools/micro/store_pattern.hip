// Store-pattern microbenchmark for BASELINE cfg4's output (256 x 64 x 224 x 224 fp32 = 3.29 GB, write-only): how fast can the
// 64-plane NCHW write stream of conv3x3+ReLU leave the chip, depending on how a workgroup's waves shape their stores?
// hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
//   V0  k_conv3x3_c3's shape: a wave owns 32 channels and every other 128-pixel group of a 16-row band; one instruction =
//       16 B per lane, lanes 0-31 -> 512 B of channel c, lanes 32-63 -> 512 B of channel c + 4
//   V1  the same bytes per wave, but one instruction = 1 KB contiguous of ONE channel (256-pixel groups)
//   V2  V1 with the wave finishing a channel's whole band (14 KB contiguous) before the next channel
//   V3  V0 with plain (not non-temporal) stores;  V4  V1 plain;  V5 linear memset-like (whole tensor, 16 B per lane)
//   band heights 8 / 16 / 32 rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 224, W = 224, C = 64, N = 256;

template <bool NT>
__device__ inline void st(float* p, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}

template <int V, bool NT>
__global__ __launch_bounds__(256, 2) void k(float* y, int th, float val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
  const int bands = H / th;
  const int band = blockIdx.x % bands, img = blockIdx.x / bands;
  const size_t plane = (size_t)H * W;
  float* base = y + (size_t)img * C * plane + (size_t)band * th * W;  // band start inside plane 0 of the image
  const int m = wave >> 1;  // channels 32m .. 32m+31
  const f32x4 v = {val, val + lane, val, val};
  const int band_px = th * W;
  if (V == 0) {
    const int groups = band_px / 128;
    for (int g = (wave & 1); g < groups; g += 2)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int cu = (i & 3) + 8 * (i >> 2);
        st<NT>(base + (size_t)(32 * m + cu + 4 * hf) * plane + g * 128 + 4 * l31, v);
      }
  } else if (V == 1) {
    const int groups = band_px / 256;
    for (int g = (wave & 1); g < groups; g += 2)
#pragma unroll
      for (int c = 0; c < 32; ++c) st<NT>(base + (size_t)(32 * m + c) * plane + g * 256 + 4 * lane, v);
  } else if (V == 2) {
    const int groups = band_px / 256;
    for (int c = (wave & 1); c < 32; c += 2)
      for (int g = 0; g < groups; ++g) st<NT>(base + (size_t)(32 * m + c) * plane + g * 256 + 4 * lane, v);
  } else if (V == 6) {  // V1 with the two waves of a channel half splitting the band in halves instead of alternating groups
    const int groups = band_px / 256, half = groups / 2;
    const int g0 = (wave & 1) ? half : 0, g1 = (wave & 1) ? groups : half;
    for (int g = g0; g < g1; ++g)
#pragma unroll
      for (int c = 0; c < 32; ++c) st<NT>(base + (size_t)(32 * m + c) * plane + g * 256 + 4 * lane, v);
  }
}

__global__ __launch_bounds__(256) void k_linear(float* y, size_t n4, float val) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const f32x4 v = {val, val, val, val};
  for (size_t j = i; j < n4; j += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y) + j);
}

// each block fills one contiguous chunk (what an elementwise fill kernel does), plain or non-temporal stores
template <bool NT>
__global__ __launch_bounds__(256) void k_chunk(float* y, size_t chunk4, float val) {
  f32x4* p = reinterpret_cast<f32x4*>(y) + (size_t)blockIdx.x * chunk4;
  const f32x4 v = {val, val, val, val};
  for (size_t j = threadIdx.x; j < chunk4; j += 256) {
    if (NT) __builtin_nontemporal_store(v, p + j);
    else p[j] = v;
  }
}

template <int V, bool NT>
static float run(float* y, int th) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9f, sum = 0.f;
  const int reps = 9;
  for (int r = 0; r < reps + 2; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, NT>), dim3(N * (H / th)), dim3(256), 0, 0, y, th, 1.0f + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) best = ms < best ? ms : best, sum += ms;
  }
  return sum / reps;
}

int main() {
  const size_t n = (size_t)N * C * H * W;
  float* y;
  if (hipMalloc(&y, n * sizeof(float)) != hipSuccess) return 1;
  const double gb = n * 4 / 1e9;
  for (int th : {8, 16, 32}) {
    printf("band %2d rows: ", th);
    float t;
    t = run<0, true>(y, th);  printf(" V0 2x512B nt %.3f ms (%.0f GB/s)", t, gb / t * 1e3);
    t = run<3 - 3, false>(y, th); printf(" | V0 plain %.3f", t);
    t = run<1, true>(y, th);  printf(" | V1 1KB nt %.3f (%.0f)", t, gb / t * 1e3);
    t = run<1, false>(y, th); printf(" | V1 plain %.3f", t);
    t = run<2, true>(y, th);  printf(" | V2 chan-seq nt %.3f (%.0f)", t, gb / t * 1e3);
    t = run<6, true>(y, th);  printf(" | V6 halves nt %.3f (%.0f)\n", t, gb / t * 1e3);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int blocks : {2048, 8192}) {
    float sum = 0;
    for (int r = 0; r < 7; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, y, n / 4, 2.0f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) sum += ms;
    }
    printf("linear fill, %d blocks: %.3f ms (%.0f GB/s)\n", blocks, sum / 5, gb / (sum / 5) * 1e3);
  }
  for (int nt = 0; nt < 2; ++nt)
    for (size_t chunk_kb : {16, 64, 256, 1024}) {
      const size_t chunk4 = chunk_kb * 1024 / 16;
      const unsigned blocks = (unsigned)(n / 4 / chunk4);
      float sum = 0;
      for (int r = 0; r < 7; ++r) {
        hipEventRecord(e0);
        if (nt) hipLaunchKernelGGL(k_chunk<true>, dim3(blocks), dim3(256), 0, 0, y, chunk4, 3.0f);
        else hipLaunchKernelGGL(k_chunk<false>, dim3(blocks), dim3(256), 0, 0, y, chunk4, 3.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) sum += ms;
      }
      printf("chunk fill %s, %4zu KB per block: %.3f ms (%.0f GB/s)\n", nt ? "nt   " : "plain", chunk_kb, sum / 5, gb / (sum / 5) * 1e3);
    }
  {
    float sum = 0;
    for (int r = 0; r < 7; ++r) {
      hipEventRecord(e0);
      hipMemsetAsync(y, 0, n * sizeof(float), 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) sum += ms;
    }
    printf("hipMemsetAsync: %.3f ms (%.0f GB/s)\n", sum / 5, gb / (sum / 5) * 1e3);
  }
  return 0;
}
