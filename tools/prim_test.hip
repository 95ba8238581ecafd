#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../old-kaldi-git_amd/csrc/kh_common.h"
namespace {
constexpr int NT = 1024;           // threads per workgroup (one utterance)
constexpr int NW = NT / 64;        // waves
constexpr uint32_t kEncInf = 0xFF800000u;  // Enc(+inf)
constexpr unsigned long long kEmpty = 0ull;

__host__ __device__ __forceinline__ uint32_t Enc(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float Dec(uint32_t e) {
  uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __builtin_bit_cast(float, u);
}

// Workgroup barrier that also waits for this wave's outstanding vector-memory
// operations.  hipcc's __syncthreads() is a WORKGROUP-scope fence: on gfx950 (one
// CU, shared L1) it does not wait for global stores to be performed at L2.  This
// kernel communicates between waves partly through L2 (atomics, sc1 loads/stores
// of words that atomics update), so a store issued before the barrier must have
// reached L2 before another wave's L2 read after it: s_waitcnt vmcnt(0) first.
__device__ __forceinline__ void KhSync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---------------------------------------------------------------- block helpers
struct Shared {
  int wsum[2][NW];                 // BlockExScan, double buffered
  unsigned long long wred[2][NW];  // block reductions, double buffered
  int orbuf[4];                    // BlockOr / BlockAny, 4 rotating slots
  unsigned long long wmin[NW];
  int flag;
  int bcast_i[4];
  float bcast_f[8];
  unsigned int hist[256];
  // running state (owned by thread 0, read after barriers)
  int tok_end, link_end;
  int front_b;  // first token of the frame under construction (frontier)
  int status;
  long long arcs_expanded, tokens_created;
  int max_tokens_frame;
  long long t_last;
  long long phase[16];
  int tok_hw;  // highest token slot dirtied by this slot's utterances so far
};

// Per-thread view of the workgroup state: the LDS block plus the (uniform)
// rotation counters of the barrier-light block primitives below.
struct Blk {
  Shared *p;
  int k_or, k_red, k_scan;
  __device__ __forceinline__ Shared *operator->() const { return p; }
};

// Block primitives with ONE barrier each.  Every primitive writes its per-wave
// partials into a buffer selected by a per-thread call counter (uniform across the
// workgroup) and reads all partials after the barrier; a buffer is rewritten two
// calls later, i.e. behind at least one more barrier than its last read.
#ifdef KH_OLD_SCAN
__device__ __forceinline__ int BlockExScan(int v, int *total, Blk &sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  if (lane == 63) sh->wsum[0][w] = inc;
  KhSync();
  if (w == 0) {
    int s = lane < NW ? sh->wsum[0][lane] : 0;
    int si = s;
#pragma unroll
    for (int o = 1; o < NW; o <<= 1) {
      int n = __shfl_up(si, o, 64);
      if (lane >= o) si += n;
    }
    if (lane < NW) sh->wsum[0][lane] = si - s;
    if (lane == NW - 1) sh->bcast_i[0] = si;
  }
  KhSync();
  const int res = sh->wsum[0][w] + inc - v;
  *total = sh->bcast_i[0];
  KhSync();
  return res;
}
#else
__device__ __forceinline__ int BlockExScan(int v, int *total, Blk &sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int buf = (sh.k_scan++) & 1;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  if (lane == 63) sh->wsum[buf][w] = inc;
  KhSync();
  int before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int t = sh->wsum[buf][i];
    before += i < w ? t : 0;
    all += t;
  }
  *total = all;
#ifdef KH_SCAN_TRAIL
  KhSync();
#endif
  return before + inc - v;
}
#endif

__device__ __forceinline__ unsigned long long BlockMinU64(unsigned long long v, Blk &sh) {
#ifdef KH_OLD_RED
  const int buf = 0;
#else
  const int buf = (sh.k_red++) & 1;
#endif
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long n = __shfl_xor(v, o, 64);
    v = n < v ? n : v;
  }
  if ((threadIdx.x & 63) == 0) sh->wred[buf][threadIdx.x >> 6] = v;
  KhSync();
  unsigned long long r = sh->wred[buf][0];
#pragma unroll
  for (int i = 1; i < NW; i++) r = sh->wred[buf][i] < r ? sh->wred[buf][i] : r;
#if defined(KH_RED_TRAIL) || defined(KH_OLD_RED)
  KhSync();
#endif
  return r;
}

__device__ __forceinline__ float BlockMinF(float v, Blk &sh) {
#ifdef KH_OLD_RED
  const int buf = 0;
#else
  const int buf = (sh.k_red++) & 1;
#endif
  v = kh_wave_min(v);
  if ((threadIdx.x & 63) == 0) sh->wred[buf][threadIdx.x >> 6] = __float_as_uint(v);
  KhSync();
  float r = __uint_as_float(static_cast<uint32_t>(sh->wred[buf][0]));
#pragma unroll
  for (int i = 1; i < NW; i++) r = fminf(r, __uint_as_float(static_cast<uint32_t>(sh->wred[buf][i])));
#if defined(KH_RED_TRAIL) || defined(KH_OLD_RED)
  KhSync();
#endif
  return r;
}

__device__ __forceinline__ long long BlockSumLL(long long v, Blk &sh) {
#ifdef KH_OLD_RED
  const int buf = 0;
#else
  const int buf = (sh.k_red++) & 1;
#endif
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh->wred[buf][threadIdx.x >> 6] = static_cast<unsigned long long>(v);
  KhSync();
  long long r = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) r += static_cast<long long>(sh->wred[buf][i]);
#if defined(KH_RED_TRAIL) || defined(KH_OLD_RED)
  KhSync();
#endif
  return r;
}

// OR over the workgroup.  Slot k is reset two calls ahead (by thread 0, before the
// barrier of call n it clears the slot of call n + 2): its previous use (call n - 2)
// was fully read before every thread reached the barrier of call n - 1.
#ifdef KH_OLD_OR
__device__ __forceinline__ int BlockOr(int bits, Blk &sh) {
  if (threadIdx.x == 0) sh->flag = 0;
  KhSync();
  if (bits) atomicOr(&sh->flag, bits);
  KhSync();
  const int r = sh->flag;
  KhSync();
  return r;
}
#else
__device__ __forceinline__ int BlockOr(int bits, Blk &sh) {
  const int k = (sh.k_or++) & 3;
  if (threadIdx.x == 0) sh->orbuf[(k + 2) & 3] = 0;
#ifdef KH_OR_LEAD
  KhSync();
#endif
  if (bits) atomicOr(&sh->orbuf[k], bits);
  KhSync();
  const int r = sh->orbuf[k];
#ifdef KH_OR_TRAIL
  KhSync();
#endif
  return r;
}
#endif

__device__ __forceinline__ bool BlockAny(bool p, Blk &sh) { return BlockOr(p ? 1 : 0, sh) != 0; }


__global__ void __launch_bounds__(NT) T(int *out, int iters) {
  __shared__ Shared shm;
  Blk sh{&shm, 0, 0, 0};
  __shared__ int vals[NT];
  if (threadIdx.x == 0) for (int i = 0; i < 4; i++) sh->orbuf[i] = 0;
  __syncthreads();
  unsigned rng = threadIdx.x * 2654435761u + 12345u;
  int bad = 0;
  for (int it = 0; it < iters; it++) {
    rng = rng * 1664525u + 1013904223u;
    const int v = (rng >> 20) & 7;
    // random per-wave delay to skew arrival
    if (((rng >> 8) & 15) == (threadIdx.x >> 6)) for (int d = 0; d < 3; d++) __builtin_amdgcn_s_sleep(20);
    vals[threadIdx.x] = v;
    int total;
    const int off = BlockExScan(v, &total, sh);
    // check against serial prefix (vals written before the scan's barrier)
    int ref = 0;
    for (int j = 0; j < threadIdx.x; j++) ref += vals[j];
    if (ref != off) bad++;
    const float m = BlockMinF((float)((rng >> 4) & 1023) + 1.0f, sh);
    if (!(m >= 1.0f)) bad++;
    const int o = BlockOr((v == 7) ? 2 : 0, sh);
    if (o != 0 && o != 2) bad++;
    const unsigned long long mm = BlockMinU64((unsigned long long)threadIdx.x + 5, sh);
    if (mm != 5) bad++;
    const long long ss = BlockSumLL(1, sh);
    if (ss != NT) bad++;
    const bool an = BlockAny(threadIdx.x == (unsigned)(it % NT), sh);
    if (!an) bad++;
    __syncthreads();  // vals reuse
  }
  atomicAdd(out, bad);
}
}
int main() {
  int *out; hipMalloc(&out, 4); hipMemset(out, 0, 4);
  hipLaunchKernelGGL(T, dim3(8), dim3(NT), 0, 0, out, 3000);
  hipDeviceSynchronize();
  int h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
  printf("primitive errors: %d\n", h);
  return 0;
}
