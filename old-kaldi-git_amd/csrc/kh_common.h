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

// Value of lane (lane ^ o), o = 32 ... 1: the butterfly steps of the reductions below, in the SAME order and pairing
// as `v op= __shfl_xor(v, o)` (results bit-identical to rounds 1-3), but only the o = 32 step is a ds_bpermute_b32:
// o = 16 is a ds_swizzle_b32 (no address register), o = 8 / 4 are two DPP moves (row_shl for the lower banks of the
// exchange, row_shr for the upper ones), o = 2 / 1 one quad_perm DPP move.  A __shfl_xor step costs a lane-address
// computation, an LDS-pipe round trip and, in register-starved kernels, the reload of the address.
template <int kO>
__device__ __forceinline__ int kh_lane_xor_i(int v) {
  if constexpr (kO == 32) {
    return __builtin_amdgcn_ds_bpermute(((static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) ^ 32) << 2), v);
  } else if constexpr (kO == 16) {
    return __builtin_amdgcn_ds_swizzle(v, 0x401F);   // bit mode: and 0x1f, or 0, xor 0x10
  } else if constexpr (kO == 8) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x108, 0xf, 0x3, false);   // row_shl:8 into banks 0-1 (lanes 0-7 of a row)
    return __builtin_amdgcn_update_dpp(t, v, 0x118, 0xf, 0xc, false);          // row_shr:8 into banks 2-3
  } else if constexpr (kO == 4) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0 and 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);          // row_shr:4 into banks 1 and 3
  } else if constexpr (kO == 2) {
    return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);           // quad_perm:[2,3,0,1]
  } else {
    return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);           // quad_perm:[1,0,3,2]
  }
}
template <int kO>
__device__ __forceinline__ float kh_lane_xor_f(float v) { return __int_as_float(kh_lane_xor_i<kO>(__float_as_int(v))); }
template <int kO>
__device__ __forceinline__ double kh_lane_xor_d(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = static_cast<unsigned int>(kh_lane_xor_i<kO>(static_cast<int>(static_cast<unsigned int>(b))));
  const unsigned int hi = static_cast<unsigned int>(kh_lane_xor_i<kO>(static_cast<int>(static_cast<unsigned int>(b >> 32))));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
#define KH_BUTTERFLY(v, OP, X)                                                        \
  v = OP(v, X<32>(v)); v = OP(v, X<16>(v)); v = OP(v, X<8>(v)); v = OP(v, X<4>(v));   \
  v = OP(v, X<2>(v)); v = OP(v, X<1>(v))
__device__ __forceinline__ float kh_op_max_f(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float kh_op_min_f(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float kh_op_add_f(float a, float b) { return a + b; }
__device__ __forceinline__ double kh_op_add_d(double a, double b) { return a + b; }
__device__ __forceinline__ int kh_op_add_i(int a, int b) { return a + b; }

__device__ __forceinline__ float kh_wave_max(float v) {
  KH_BUTTERFLY(v, kh_op_max_f, kh_lane_xor_f);
  return v;
}
__device__ __forceinline__ float kh_wave_min(float v) {
  KH_BUTTERFLY(v, kh_op_min_f, kh_lane_xor_f);
  return v;
}
__device__ __forceinline__ float kh_wave_sum(float v) {
  KH_BUTTERFLY(v, kh_op_add_f, kh_lane_xor_f);
  return v;
}
__device__ __forceinline__ double kh_wave_sum_d(double v) {
  KH_BUTTERFLY(v, kh_op_add_d, kh_lane_xor_d);
  return v;
}
__device__ __forceinline__ int kh_wave_sum_i(int v) {
  KH_BUTTERFLY(v, kh_op_add_i, kh_lane_xor_i);
  return v;
}

#endif  // KH_COMMON_H_
