// How fast does ONE dependent chain of v_mfma_f32_32x32x2_f32 issue on gfx950?  (Each output of the general 3x3 conv is one such
// chain; a 1x1 wave tile gives a wave a single accumulator.)  NACC independent accumulators per wave, WAVES / 4 waves per SIMD,
// KL LDS reads per MFMA; time per MFMA issued by a wave.
// hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int KL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 64 * WAVES) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 acc[NACC] = {};
  float a = lane * 0.01f, b = 1.0f;
  const float* q = lds + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float av = a, bv = b;
      if (KL >= 1) av = q[(s * 2) * 64 & 8191];
      if (KL >= 2) bv = q[(s * 2 + 1) * 64 & 8191];
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) r += acc[t][i];
  out[blockIdx.x * 64 * WAVES + tid] = r;
}

template <int NACC, int KL, int WAVES>
static void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, KL, WAVES>), dim3(256), dim3(64 * WAVES), 32768, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%-64s %7.1f ns per MFMA of a wave (%5.1f clocks at 2.4 GHz)\n", name, best * 1e6 / (iters * 16.0 * NACC), best * 1e6 / (iters * 16.0 * NACC) * 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<1, 0, 4>(out, "1 wave / SIMD, 1 accumulator");
  run<2, 0, 4>(out, "1 wave / SIMD, 2 accumulators");
  run<4, 0, 4>(out, "1 wave / SIMD, 4 accumulators");
  run<1, 2, 4>(out, "1 wave / SIMD, 1 accumulator, operands from LDS");
  run<2, 2, 4>(out, "1 wave / SIMD, 2 accumulators, operands from LDS");
  run<4, 2, 4>(out, "1 wave / SIMD, 4 accumulators, operands from LDS");
  run<1, 0, 8>(out, "2 waves / SIMD, 1 accumulator each");
  run<1, 2, 8>(out, "2 waves / SIMD, 1 accumulator each, operands from LDS");
  run<2, 2, 8>(out, "2 waves / SIMD, 2 accumulators each, operands from LDS");
  run<1, 2, 16>(out, "4 waves / SIMD, 1 accumulator each, operands from LDS");
  return 0;
}
