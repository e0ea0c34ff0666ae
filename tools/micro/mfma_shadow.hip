// Does a wave's OWN VALU / LDS work issue in the shadow of its MFMAs on gfx950?  (Follow-up of mfma_valu_overlap.hip, where
// another wave's instructions on the same SIMD added their full time to the MFMA wave's.)  One wave per SIMD; per
// v_mfma_f32_32x32x2_f32 (64 cycles in the matrix pipe) the wave also issues K independent fp32 VALU ops, or K LDS reads.
// hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KV, int KL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 64 * WAVES) lds[i] = i * 0.001f;
  __syncthreads();
  f32x16 acc[4] = {};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = lane + i;
  float a = lane * 0.01f, b = 1.0f, sum = 0.f;
  const float* q = lds + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 18; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < KV; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < KL; ++i) sum += q[((s * 4 + t) * KL + i) * 64 & 8191];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float r = sum;
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) r += acc[t][i];
  for (int i = 0; i < 16; ++i) r += v[i];
  out[blockIdx.x * 64 * WAVES + tid] = r;
}

template <int KV, int KL, int WAVES>
static void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KV, KL, WAVES>), dim3(256), dim3(64 * WAVES), 32768, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%-58s %.3f us per 72 MFMAs\n", name, best * 1e3 / iters);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<0, 0, 4>(out, "1 wave / SIMD: MFMAs only");
  run<2, 0, 4>(out, "1 wave / SIMD: + 2 VALU per MFMA");
  run<4, 0, 4>(out, "1 wave / SIMD: + 4 VALU per MFMA");
  run<8, 0, 4>(out, "1 wave / SIMD: + 8 VALU per MFMA");
  run<12, 0, 4>(out, "1 wave / SIMD: + 12 VALU per MFMA");
  run<16, 0, 4>(out, "1 wave / SIMD: + 16 VALU per MFMA");
  run<0, 1, 4>(out, "1 wave / SIMD: + 1 LDS read per MFMA");
  run<0, 2, 4>(out, "1 wave / SIMD: + 2 LDS reads per MFMA");
  run<0, 4, 4>(out, "1 wave / SIMD: + 4 LDS reads per MFMA");
  run<4, 2, 4>(out, "1 wave / SIMD: + 4 VALU + 2 LDS reads per MFMA");
  run<0, 0, 8>(out, "2 waves / SIMD: MFMAs only");
  run<4, 0, 8>(out, "2 waves / SIMD: + 4 VALU per MFMA");
  run<8, 0, 8>(out, "2 waves / SIMD: + 8 VALU per MFMA");
  run<4, 2, 8>(out, "2 waves / SIMD: + 4 VALU + 2 LDS reads per MFMA");
  return 0;
}
