// Do one wave's MFMAs and another wave's VALU / LDS work on the SAME SIMD overlap on gfx950?  (Design question of the
// producer / consumer split in csrc/deform_fused.hip.)  One 512-thread workgroup per CU (LDS-limited): waves 0-3 run 4
// independent v_mfma_f32_32x32x2_f32 chains, waves 4-7 run either fp32 VALU chains, LDS reads (b128, scattered) or both.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = consumer MFMAs, 2 = producer VALU, 4 = producer LDS gather reads, 8 = consumer LDS operand reads,
//            16 = the MFMA accumulators live in AGPRs, 32 = the producer's VALU work is ONE dependent chain, 64 = producer LDS reads
//            with immediate offsets only (no VALU)
template <bool AGPR>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mode, int prio) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  if (wave < 4) {
    if (!(mode & 1)) return;
    if (prio) __builtin_amdgcn_s_setprio(2);
    f32x16 acc[4] = {};
    float a = lane * 0.01f, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        if (mode & 8) {
          a = lds[(it * 37 + s * 160 + lane) & 16383];
          b = lds[(lane * 38 + s * 2 + 8000) & 16383];
        }
        if (AGPR) {
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    float r = 0;
    for (int t = 0; t < 4; ++t)
      for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 512 + tid] = r;
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    unsigned h = tid * 2654435761u;
    for (int it = 0; it < iters; ++it) {
      if (mode & 4) {
#pragma unroll
        for (int u = 0; u < 20; ++u) {  // 20 scattered 16-byte reads, like 5 taps x 4 corners
          h = h * 1664525u + 1013904223u;
          const f32x4 q = *reinterpret_cast<const f32x4*>(lds + ((h >> 8) & 4095) * 4);
          v[u & 7] += q.x + q.w;
        }
      }
      if (mode & 64) {
        const float* q = lds + lane * 4;
#pragma unroll
        for (int u = 0; u < 40; ++u) v[u & 7] += q[u * 256];  // 40 ds_read_b32, addresses = one VGPR + immediates
      }
      if (mode & 32) {
#pragma unroll
        for (int u = 0; u < 80; ++u) v[0] = v[0] * 1.0001f + 0.5f;  // 80 dependent ops
      }
      if (mode & 2) {
#pragma unroll
        for (int u = 0; u < 40; ++u)  // ~320 dependent-free VALU ops
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f + 0.5f;
      }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + tid] = r;
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[] = {"MFMA only", "VALU only", "MFMA + VALU", "LDS gather only", "MFMA + LDS gather", "VALU + gather", "MFMA + VALU + gather",
                         "MFMA(+operand reads)", "MFMA(+operand reads) + VALU + gather", "MFMA(AGPR) only", "MFMA(AGPR) + VALU",
                         "dependent VALU chain only", "MFMA + dependent VALU chain", "MFMA(AGPR) + dependent VALU chain",
                         "plain LDS reads only", "MFMA + plain LDS reads", "MFMA(AGPR) + plain LDS reads"};
  const int modes[] = {1, 2, 3, 4, 5, 6, 7, 9, 15, 17, 19, 32, 33, 49, 64, 65, 81};
  for (int prio = 0; prio < 2; ++prio)
    for (int m = 0; m < 17; ++m) {
      float best = 1e9f;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        if (modes[m] & 16) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 128 * 1024, 0, out, iters, modes[m], prio);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 128 * 1024, 0, out, iters, modes[m], prio);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
      }
      printf("prio %d  %-40s %.3f ms  (%.2f us per iteration; MFMA work = 72 x 64 cycles)\n", prio, names[m], best, best * 1e3 / iters);
    }
  return 0;
}
