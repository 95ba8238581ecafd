// kh_common.h — shared internals of libkaldi_hip.so (gfx950 only).
#ifndef KH_COMMON_H_
#define KH_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/kaldi_hip.h"

namespace kh {

// ---- error reporting (KALDI_ERR / CU_SAFE_CALL equivalents) -------------------
void SetError(const char *fmt, ...);
const char *LastError();

#define KH_CHECK_ARG(cond)                                                  \
  do {                                                                      \
    if (!(cond)) {                                                          \
      kh::SetError("%s:%d: argument check failed: %s", __FILE__, __LINE__,  \
                   #cond);                                                  \
      return KH_EINVAL;                                                     \
    }                                                                       \
  } while (0)

#define KH_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      kh::SetError("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,       \
                   hipGetErrorString(e_));                                  \
      return KH_EDEVICE;                                                    \
    }                                                                       \
  } while (0)

#define KH_LAUNCH_CHECK() KH_HIP(hipGetLastError())

// Every compute entry point starts with this: fails loudly without a device.
int EnsureDevice();
hipStream_t Stream();

// caching allocator (kh_runtime.hip)
void *PoolMalloc(size_t bytes);
int PoolFree(void *p);
size_t PoolCachedBytes();  // freed blocks the pool still holds (given back to HIP when an allocation fails)

inline int DivUp(int a, int b) { return (a + b - 1) / b; }
inline int64_t DivUp64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Number of CUs of the selected device (256 on MI355X).
int NumCUs();

// Fused tail of the nnet2 output layer (kh_elementwise.hip): SoftmaxComponent
// (softmax + its 1e-20 floor, nnet-component.cc:929-943) -> SumGroupComponent, and
// when log_priors != nullptr DecodableAmNnet's floor / log / -log prior / scale
// (decodable-am-nnet.h:60-69), one pass over the logits.  Same operations in the same
// order as the separate kernels.  Requires d_in.cols <= SoftmaxLdsCols().
int FusedSoftmaxSumGroup(float *y, KhMatrixDim d_out, const float *x, KhMatrixDim d_in, const int32_t *ranges,
                         const float *log_priors, float prob_scale);
int SoftmaxLdsCols();

}  // namespace kh

// ---- device helpers --------------------------------------------------------------
#define KH_WAVE 64

__device__ __forceinline__ float kh_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float kh_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float kh_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double kh_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int kh_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#endif  // KH_COMMON_H_
