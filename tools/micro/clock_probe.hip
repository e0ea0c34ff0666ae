// Shader-clock probe for tools/launch_series.py: a one-workgroup kernel that spins ~20 us and records how many shader
// cycles (s_memtime, runs at the current shader clock) elapsed per tick of the constant 100 MHz reference counter
// (s_memrealtime).  Launched between timed launches of a kernel under study it says which clock the chip was running at
// that moment -- i.e. whether a launch-time trend is the clocks ramping after idle.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_clock_probe(uint64_t* out, int spin_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t r0 = wall_clock64(), c0 = clock64();
  uint64_t r1 = r0;
  while ((int64_t)(r1 - r0) < spin_ticks) r1 = wall_clock64();
  const uint64_t c1 = clock64();
  out[0] = c1 - c0;  // shader cycles
  out[1] = r1 - r0;  // 100 MHz ticks
}

extern "C" int clock_probe(void* out2, int spin_ticks, void* stream) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t*)out2, spin_ticks);
  return (int)hipGetLastError();
}
