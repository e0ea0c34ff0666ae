// VALU issue-rate microbenchmark (gfx950): plain v_fma_f32 vs v_pk_fma_f32, 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  f32x2 q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = (f32x2){r[2 * i], r[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (PK) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q[i]) : "v"((f32x2){a, a}), "v"((f32x2){b, b}));
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += q[i].x + q[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
      const int blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) {
          const double instr = (double)iters * (pk ? 64 : 128);       // per wave
          const double fma = (double)iters * 128 * 64 * 4.0 * blocks;  // lane-FMAs in total
          printf("%s waves/SIMD=%d  %.3f ms  %.2f TFMA/s  (%.2f ns per wave-instr; at 2.4 GHz %.2f clk)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ",
                 wps, ms, fma / ms / 1e9, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
        }
      }
    }
  return 0;
}
