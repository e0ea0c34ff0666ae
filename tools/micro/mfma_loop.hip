// What, in the K loop of the general 3x3 conv's 1x1 wave tile, takes a step from the 65 clocks of its MFMA to ~115?  One
// accumulator per wave, 18 steps per "chunk", operands from LDS; features are added one at a time:
//   BAR    one __syncthreads per chunk          PARTNERS  4 more waves that only take part in the barriers
//   VADDR  per-step LDS addresses from registers that change per chunk (double-buffer flip), as in the kernel
//   LOOK   steps of operand lookahead
// hipcc --offload-arch=gfx950 -O3 -o mfma_loop mfma_loop.hip && ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool BAR, bool PARTNERS, bool VADDR, int LOOK>
__global__ __launch_bounds__(PARTNERS ? 512 : 256) void k(float* out, int chunks, int pitch, int nrp) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 12288; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (PARTNERS && wave_all >= 4) {
    if (BAR)
      for (int c = 0; c < chunks; ++c) __syncthreads();
    return;
  }
  f32x16 acc = {};
  const int hf = lane >> 5, lb = (lane & 31) + (wave_all & 3) * 32;
  for (int c = 0; c < chunks; ++c) {
    const float* wfr = lds + (VADDR ? (c & 1) * 8192 : 0);
    const float* xin = wfr + 2048;
    auto fetch = [&](int s, float& av, float& bv) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = (k0 / 9) * nrp + ((k0 % 9) / 3) * pitch + (k0 % 9) % 3;
      const int o1 = (k1 / 9) * nrp + ((k1 % 9) / 3) * pitch + (k1 % 9) % 3;
      av = wfr[s * 64 + lane];
      bv = VADDR ? xin[lb + (hf ? o1 : o0)] : xin[lb + s * 64];
    };
    float av[LOOK + 1], bv[LOOK + 1];
#pragma unroll
    for (int d = 0; d < LOOK; ++d) fetch(d, av[d], bv[d]);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s + LOOK < 18) fetch(s + LOOK, av[(s + LOOK) % (LOOK + 1)], bv[(s + LOOK) % (LOOK + 1)]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % (LOOK + 1)], bv[s % (LOOK + 1)], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __syncthreads();
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[blockIdx.x * 256 + (tid & 255)] = r;
}

template <bool BAR, bool PARTNERS, bool VADDR, int LOOK>
static void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int chunks = 4000;
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BAR, PARTNERS, VADDR, LOOK>), dim3(200), dim3(PARTNERS ? 512 : 256), 49152, 0, out, chunks, 64, 384);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%-72s %6.1f clocks per step at 2.4 GHz\n", name, best * 1e6 / (chunks * 18.0) * 2.4);
}

// The same loop with the B operand's address for step s+1 computed BEFORE the MFMA of step s (own scheduling region), so that no
// VALU instruction stands between an MFMA and the LDS reads that follow it.
template <int EARLY>
__global__ __launch_bounds__(512) void k2(float* out, int chunks, int pitch, int nrp) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 12288; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave_all >= 4) {
    for (int c = 0; c < chunks; ++c) __syncthreads();
    return;
  }
  f32x16 acc = {};
  const int hf = lane >> 5, lb = (lane & 31) + (wave_all & 3) * 32;
  int boff[18];
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int o0 = (k0 / 9) * nrp + ((k0 % 9) / 3) * pitch + (k0 % 9) % 3;
    const int o1 = (k1 / 9) * nrp + ((k1 % 9) / 3) * pitch + (k1 % 9) % 3;
    boff[s] = lb + (hf ? o1 : o0);
  }
  for (int c = 0; c < chunks; ++c) {
    const float* wfr = lds + (c & 1) * 8192;
    const float* xin = wfr + 2048;
    const int xbase = ((c & 1) * 8192 + 2048) * 4 + (int)(unsigned)(size_t)lds;  // LDS byte address
    int pb[3];
    float av[2], bv[2];
    pb[0] = xbase + boff[0] * 4;
    pb[1] = xbase + boff[1] * 4;
    typedef const __attribute__((address_space(3))) float lds_cf;
    auto ldsb = [&](int addr) { return *reinterpret_cast<lds_cf*>((size_t)(unsigned)addr); };
    av[0] = wfr[lane], bv[0] = ldsb(pb[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (EARLY && s + 2 < 18) {
        pb[(s + 2) % 3] = xbase + boff[s + 2] * 4;
        asm volatile("" : "+v"(pb[(s + 2) % 3]));  // keep the add here
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1], acc, 0, 0, 0);
      if (s + 1 < 18) {
        av[(s + 1) & 1] = wfr[(s + 1) * 64 + lane];
        bv[(s + 1) & 1] = EARLY ? ldsb(pb[(s + 1) % 3]) : xin[boff[s + 1]];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[blockIdx.x * 256 + (tid & 255)] = r;
}

template <int EARLY>
static void run2(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int chunks = 4000;
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<EARLY>), dim3(200), dim3(512), 49152, 0, out, chunks, 64, 384);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%-72s %6.1f clocks per step at 2.4 GHz\n", name, best * 1e6 / (chunks * 18.0) * 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<false, false, false, 1>(out, "plain: 4 waves, look 1");
  run<true, false, false, 1>(out, "+ barrier per chunk");
  run<true, true, false, 1>(out, "+ barrier per chunk + 4 partner waves");
  run<false, false, true, 1>(out, "+ kernel-like addresses");
  run<true, false, true, 1>(out, "+ kernel-like addresses + barrier");
  run<true, true, true, 1>(out, "+ kernel-like addresses + barrier + partners (the kernel's loop)");
  run<true, true, true, 2>(out, "the kernel's loop, look 2");
  run<true, true, true, 3>(out, "the kernel's loop, look 3");
  run<false, true, true, 1>(out, "the kernel's loop without barriers (partners exit)");
  run2<0>(out, "offsets in registers, address add between the MFMA and its reads");
  run2<1>(out, "offsets in registers, address add BEFORE the MFMA (own region)");
  return 0;
}
