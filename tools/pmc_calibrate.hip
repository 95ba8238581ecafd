// Calibration of FETCH_SIZE / WRITE_SIZE on gfx950 in the decoder's own access shapes
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern
// before trusting an absolute").  Every kernel touches a KNOWN set of bytes of a 4 GiB buffer (16 x the
// Infinity Cache) exactly once; run under `rocprofv3 --pmc …` (tools/pmc_calibrate.sh) and compare the
// counters with the "known" column this program prints.
//
//   hipcc --offload-arch=gfx950 -O2 -o old-kaldi-git_amd/build/pmc_calibrate tools/pmc_calibrate.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

static constexpr int kThreads = 1024;   // the decoder's workgroup
static constexpr int kBlocks = 512;     // two per CU

__device__ __forceinline__ uint64_t Mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// a bijection on [0, 2^bits): every unit is touched exactly once, in scattered order
__device__ __forceinline__ uint64_t Perm(uint64_t i, int bits) {
  const uint64_t mask = (1ULL << bits) - 1;
  i = (i * 0x9E3779B97F4A7C15ULL) & mask;       // odd multiplier: a bijection mod 2^bits
  i ^= i >> (bits / 2);
  i = (i * 0xD6E8FEB86659FD93ULL) & mask;
  i ^= i >> (bits / 2 + 1);
  return i & mask;
}

// ---- reads ----
// coalesced, one dword per lane (256 B per wave instruction): token / link field sweeps
__global__ void CalStreamDword(const uint32_t *p, uint64_t n, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads) acc += p[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
// coalesced, 16 B per lane (1 KiB per wave instruction): the guide's calibrated case
__global__ void CalStreamDwordx4(const uint4 *p, uint64_t n, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// coalesced, 2 B per lane (128 B per wave instruction): the 16-bit pdf of an arc
__global__ void CalStreamUshort(const uint16_t *p, uint64_t n, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads) acc += p[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
// one dword out of every 128-B line, lines in scattered order (64 lanes -> 64 lines): gathers of costs
__global__ void CalLine128One(const uint32_t *p, int bits, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t n = 1ULL << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    acc += p[Perm(i, bits) * 32];
  if (acc == 0x12345678u) sink[0] = acc;
}
// the same lines, but BOTH 64-B halves of each line read (adjacent lanes): if the counters do not
// change against CalLine128One, a miss moves the whole 128-B line
__global__ void CalLine128Both(const uint32_t *p, int bits, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t n = 2ULL << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    acc += p[Perm(i >> 1, bits) * 32 + (i & 1) * 16];
  if (acc == 0x12345678u) sink[0] = acc;
}
// one dword out of every 64-B line (both halves of every 128-B line, but at unrelated times)
__global__ void CalLine64One(const uint32_t *p, int bits, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t n = 1ULL << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    acc += p[Perm(i, bits) * 16];
  if (acc == 0x12345678u) sink[0] = acc;
}
// the arc table: runs of `run` consecutive 16-B records (lanes of a group read consecutive records),
// the runs in scattered order; run = 8 -> 128-B runs, run = 4 -> 64-B runs
template <int kRun>
__global__ void CalArcRuns(const uint4 *p, int bits, uint32_t *sink) {
  uint32_t acc = 0;
  const uint64_t n = (uint64_t)kRun << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads) {
    const uint4 v = p[Perm(i / kRun, bits) * kRun + (i % kRun)];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// ---- writes ----
__global__ void CalStoreDword(uint32_t *p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads) p[i] = (uint32_t)i;
}
__global__ void CalStoreDwordx4(uint4 *p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// one dword into every 64-B line, scattered (extra_cost write-backs, remap tables)
__global__ void CalStoreLine64One(uint32_t *p, int bits) {
  const uint64_t n = 1ULL << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    p[Perm(i, bits) * 16] = (uint32_t)i;
}
// scattered atomicMin, one per 64-B line (what is left of the global atomics)
__global__ void CalAtomicLine64One(uint32_t *p, int bits) {
  const uint64_t n = 1ULL << bits;
  for (uint64_t i = blockIdx.x * (uint64_t)kThreads + threadIdx.x; i < n; i += (uint64_t)kBlocks * kThreads)
    atomicMin(&p[Perm(i, bits) * 16], (uint32_t)i);
}

template <class F>
static void Run(const char *name, double known_read, double known_write, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-22s known_read_bytes %.6e known_write_bytes %.6e  %.3f ms  %.2f TB/s of known bytes\n", name, known_read, known_write, ms,
         (known_read + known_write) / ms * 1e-9);
}

int main() {
  const uint64_t bytes = 4ULL << 30;
  void *buf = nullptr;
  uint32_t *sink = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  const int b128 = 25, b64 = 26;     // 2^25 lines of 128 B = 2^26 lines of 64 B = 4 GiB
  const double GiB4 = (double)bytes;
  Run("CalStreamDword", GiB4, 0, [&] { hipLaunchKernelGGL(CalStreamDword, kBlocks, kThreads, 0, 0, (const uint32_t *)buf, bytes / 4, sink); });
  Run("CalStreamDwordx4", GiB4, 0, [&] { hipLaunchKernelGGL(CalStreamDwordx4, kBlocks, kThreads, 0, 0, (const uint4 *)buf, bytes / 16, sink); });
  Run("CalStreamUshort", GiB4 / 2, 0, [&] { hipLaunchKernelGGL(CalStreamUshort, kBlocks, kThreads, 0, 0, (const uint16_t *)buf, bytes / 4, sink); });
  // known bytes of the scattered cases are given at 64-B granularity (lines of 64 B that hold a requested byte)
  Run("CalLine128One", 64.0 * (1ULL << b128), 0, [&] { hipLaunchKernelGGL(CalLine128One, kBlocks, kThreads, 0, 0, (const uint32_t *)buf, b128, sink); });
  Run("CalLine128Both", 128.0 * (1ULL << b128), 0, [&] { hipLaunchKernelGGL(CalLine128Both, kBlocks, kThreads, 0, 0, (const uint32_t *)buf, b128, sink); });
  Run("CalLine64One", 64.0 * (1ULL << b64), 0, [&] { hipLaunchKernelGGL(CalLine64One, kBlocks, kThreads, 0, 0, (const uint32_t *)buf, b64, sink); });
  Run("CalArcRuns8", GiB4, 0, [&] { hipLaunchKernelGGL(CalArcRuns<8>, kBlocks, kThreads, 0, 0, (const uint4 *)buf, b128, sink); });
  Run("CalArcRuns4", GiB4, 0, [&] { hipLaunchKernelGGL(CalArcRuns<4>, kBlocks, kThreads, 0, 0, (const uint4 *)buf, b64, sink); });
  Run("CalStoreDword", 0, GiB4, [&] { hipLaunchKernelGGL(CalStoreDword, kBlocks, kThreads, 0, 0, (uint32_t *)buf, bytes / 4); });
  Run("CalStoreDwordx4", 0, GiB4, [&] { hipLaunchKernelGGL(CalStoreDwordx4, kBlocks, kThreads, 0, 0, (uint4 *)buf, bytes / 16); });
  Run("CalStoreLine64One", 0, 4.0 * (1ULL << b64), [&] { hipLaunchKernelGGL(CalStoreLine64One, kBlocks, kThreads, 0, 0, (uint32_t *)buf, b64); });
  Run("CalAtomicLine64One", 0, 4.0 * (1ULL << b64), [&] { hipLaunchKernelGGL(CalAtomicLine64One, kBlocks, kThreads, 0, 0, (uint32_t *)buf, b64); });
  CK(hipFree(buf));
  CK(hipFree(sink));
  return 0;
}
