// Does v_cvt_pk_u8_f32 round to nearest even by itself (so that a v_rndne_f32 in front of it is redundant)?
// hipcc --offload-arch=gfx950 -O3 -o cvt_pk_u8 cvt_pk_u8.hip && ./cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* x, unsigned* direct, unsigned* rounded, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  direct[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
  rounded[i] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(x[i]), 0, 0u);
}
int main() {
  const int n = 1 << 20;
  float* hx = new float[n];
  unsigned seed = 12345u;
  for (int i = 0; i < n; ++i) {
    if (i < 1024) hx[i] = (i / 2) * 0.5f + ((i & 1) ? 1e-6f * (i / 2) : 0.f);  // every x.0 / x.5 and values just above
    else {
      seed = seed * 1664525u + 1013904223u;
      hx[i] = (float)(seed >> 8) / (float)(1u << 24) * 258.f - 1.5f;  // [-1.5, 256.5)
    }
  }
  float* dx; unsigned *dd, *dr;
  hipMalloc(&dx, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dd, dr, n);
  unsigned *hd = new unsigned[n], *hr = new unsigned[n];
  hipMemcpy(hd, dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost);
  int diff = 0, bad_ref = 0;
  for (int i = 0; i < n; ++i) {
    float r = nearbyintf(hx[i]);
    unsigned want = r < 0.f ? 0u : (r > 255.f ? 255u : (unsigned)r);
    if (hr[i] != want) ++bad_ref;
    if (hd[i] != hr[i]) { if (diff < 10) printf("x = %.7f direct %u rint-then-pack %u\n", hx[i], hd[i], hr[i]); ++diff; }
  }
  printf("%d of %d values differ between v_cvt_pk_u8_f32(x) and v_cvt_pk_u8_f32(rint(x)); rint-then-pack != host reference: %d\n", diff, n, bad_ref);
  return 0;
}
