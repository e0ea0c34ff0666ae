// Byte-misaligned 4- and 16-byte global loads / stores on gfx950: correctness check (hipcc emits global_load_dwordx4 for
// under-aligned vector types: the amdhsa target runs with unaligned access mode).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned char u8x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned int u32x4b __attribute__((ext_vector_type(4), aligned(1)));
__global__ void k4(const unsigned char* x, unsigned char* y, int off, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *reinterpret_cast<u8x4u*>(y + off + 4 * i) = *reinterpret_cast<const u8x4u*>(x + off + 4 * i);
}
__global__ void k16(const unsigned char* x, unsigned char* y, int off, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *reinterpret_cast<u32x4b*>(y + off + 16 * i) = *reinterpret_cast<const u32x4b*>(x + off + 16 * i);
}
int main() {
  const int N = 1 << 20;
  std::vector<unsigned char> h(N + 64), o(N + 64);
  for (int i = 0; i < N + 64; ++i) h[i] = (unsigned char)(i * 131 + 7);
  unsigned char *x, *y;
  hipMalloc(&x, N + 64), hipMalloc(&y, N + 64);
  hipMemcpy(x, h.data(), N + 64, hipMemcpyHostToDevice);
  int bad = 0;
  for (int off = 0; off < 8; ++off) {
    hipMemset(y, 0, N + 64);
    hipLaunchKernelGGL(k4, dim3(N / 4 / 256), dim3(256), 0, 0, x, y, off, N / 4);
    hipMemcpy(o.data(), y, N + 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) bad += o[off + i] != h[off + i];
    hipMemset(y, 0, N + 64);
    hipLaunchKernelGGL(k16, dim3(N / 16 / 256), dim3(256), 0, 0, x, y, off, N / 16);
    hipMemcpy(o.data(), y, N + 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) bad += o[off + i] != h[off + i];
  }
  printf("byte-misaligned 4/16-byte accesses, offsets 0..7: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  return bad != 0;
}
