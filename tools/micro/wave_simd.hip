// Which SIMD does wave w of a workgroup land on?  (gfx9 HW_ID: bits 5:4 = SIMD_ID, 11:8 = CU_ID.)  The general 3x3 conv's
// loader / compute specialisation wants ONE compute wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o wave_simd wave_simd.hip && ./wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}

int main() {
  unsigned* out;
  hipMalloc(&out, 4096 * sizeof(unsigned));
  for (int threads : {256, 512, 1024}) {
    hipMemset(out, 0, 4096 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(3), dim3(threads), 0, 0, out);
    hipDeviceSynchronize();
    unsigned h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    for (int b = 0; b < 3; ++b) {
      printf("%4d threads, workgroup %d: SIMD of waves 0..%d =", threads, b, waves - 1);
      for (int w = 0; w < waves; ++w) printf(" %u", (h[b * waves + w] >> 4) & 3u);
      printf("   (CU %u)\n", (h[b * waves] >> 8) & 15u);
    }
  }
  return 0;
}
