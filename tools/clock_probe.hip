// clock_probe.hip — what clock64() (s_memtime) counts on this chip, against the 100 MHz wall clock (s_memrealtime) and a
// chain of dependent VALU instructions: one lone wave, then the same wave with every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void Probe(long long *out, int n, float seed) {
  const long long t0 = clock64(), w0 = wall_clock64();
  float x = seed;
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  const long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = static_cast<long long>(x); }
}
int main() {
  long long *d, h[3];
  hipMalloc(&d, 24);
  const int n = 4000000;
  for (int grid : {1, 256, 2048}) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(Probe, dim3(grid), dim3(grid == 1 ? 64 : 256), 0, 0, d, n, 1.0f);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      const double sec = h[1] / 1e8;
      printf("grid %4d: clock64 %lld ticks, wall %.3f ms -> clock64 at %.1f MHz; %d dependent v_fma in %.3f ms = %.2f ns each\n", grid, h[0],
             sec * 1e3, h[0] / sec / 1e6, n, sec * 1e3, sec * 1e9 / n);
    }
  }
  return 0;
}
