// Micro-test: within ONE workgroup (one CU), is a plain global store by wave B visible to a
// later plain global load by wave A (after __syncthreads) when A already had the line in L1?
// Also: L2 atomic by B, then plain load by A.
#include <hip/hip_runtime.h>
#include <cstdio>
#define GP(T) __attribute__((address_space(1))) T *
__global__ void __launch_bounds__(1024) Test(int *buf_, int *out, int iters, int mode) {
  GP(int) buf = (GP(int))buf_;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  int stale = 0;
  for (int it = 1; it <= iters; it++) {
    // every wave reads the whole 16-line region (brings lines into L1)
    int acc = 0;
    for (int k = lane; k < 512; k += 64) acc += buf[k];
    __syncthreads();
    // wave (it % 16) writes new values
    if (w == (it & 15)) {
      for (int k = lane; k < 512; k += 64) {
        if (mode == 0) buf[k] = it;                                   // plain store
        else if (mode == 1) __hip_atomic_store(&buf[k], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store
        else __hip_atomic_exchange(&buf[k], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // L2 atomic
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // everyone reads back with plain loads
    for (int k = lane; k < 512; k += 64) if (buf[k] != it) stale++;
    if (acc == 0x7fffffff) out[1] = acc;
    __syncthreads();
  }
  atomicAdd(&out[0], stale);
}
int main() {
  int *buf, *out;
  hipMalloc(&buf, 4096 * 4); hipMalloc(&out, 64);
  const char *names[3] = {"plain store -> plain load", "sc1 store -> plain load", "L2 atomic -> plain load"};
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(buf, 0, 4096 * 4); hipMemset(out, 0, 64);
    hipLaunchKernelGGL(Test, dim3(64), dim3(1024), 0, 0, buf + 0, out, 2000, mode);  // 64 blocks hammer the SAME buffer? no: keep 1 block
    hipDeviceSynchronize();
    hipMemset(buf, 0, 4096 * 4); hipMemset(out, 0, 64);
    hipLaunchKernelGGL(Test, dim3(1), dim3(1024), 0, 0, buf, out, 20000, mode);
    hipDeviceSynchronize();
    int h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("%-28s: stale reads = %d of %d\n", names[mode], h[0], 20000 * 512 * 16);
  }
  return 0;
}
