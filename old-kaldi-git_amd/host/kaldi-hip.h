// kaldi-hip.h — C++ host layer above the C-ABI (include/kaldi_hip.h).
//
// Mirrors the reference's class interface for the hot path — same names, argument
// meaning and error behaviour — so that code written against cudamatrix/cu-matrix.h,
// cu-vector.h, cu-array.h, cu-device.h, cu-math.h, decoder/lattice-faster-decoder.h
// reads the same here:
//   kaldi::CuDevice            cudamatrix/cu-device.h:41-143
//   kaldi::CuMatrixBase<Real> / CuMatrix<Real> / CuSubMatrix<Real>   cudamatrix/cu-matrix.h:62-644  (forward-path
//                              subset; every operation also works on Range() / RowRange() / ColRange() views;
//                              templates like the reference's: float and double (the matrix primitives of the forward
//                              path have <double> kernels; what is float-only throws on a <double> object, see KhF below)
//   kaldi::CuVectorBase<Real> / CuVector<Real> / CuSubVector<Real>   cudamatrix/cu-vector.h
//   kaldi::CuValue<Real>       cudamatrix/cu-value.h:33-81
//   kaldi::Matrix<Real> / Vector<Real> (host)   kaldi-matrix-lite.h, or matrix/kaldi-matrix.h when already included
//   kaldi::CuArray<T>          cudamatrix/cu-array.h:36-105
//   kaldi::cu::Splice          cudamatrix/cu-math.h
//   kaldi::DecodableInterface, DecodableMatrixMapped   itf/decodable-itf.h:82-120, decoder/decodable-matrix.h:33-84
//   kaldi::LatticeFasterDecoderConfig / LatticeFasterDecoder
//                              decoder/lattice-faster-decoder.h:40-205
//   kaldi::LatticeFasterOnlineDecoder  decoder/lattice-faster-online-decoder.h:44-200
//   kaldi::Nnet, NnetComputation       nnet2/nnet-nnet.h, nnet2/nnet-compute.cc:159-166
//   kaldi::DiagGmm (likelihoods)       gmm/diag-gmm.h:83-135
//   kaldi::LatticeForwardBackward, LatticeForwardBackwardMpeVariants
//                              lat/lattice-functions.cc:272-354,740-919
//   kaldi::Mfcc (all MfccOptions but VTLN), ComputeDeltas, AccCmvnStats, ApplyCmvn
//   DeterminizeLatticePruned (decoder-wrappers.cc:264-274), OnlineIvectorExtractor (online2/online-ivector-feature.h)
//                              feat/feature-mfcc.h, feature-functions.cc:361-372, transform/cmvn.cc:49-113
// Errors throw std::runtime_error exactly as KALDI_ERR does
// (base/kaldi-error.cc:143,179-182); all operations are synchronous at the API
// (results visible on return), like the reference's CU_SAFE_CALL
// (cudamatrix/cu-common.h:37-44).  Header-only; link with libkaldi_hip.so.
#ifndef KALDI_HIP_HOST_H_
#define KALDI_HIP_HOST_H_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/kaldi_hip.h"
// Host-side Matrix<Real> / Vector<Real>: the reference's own header when the translation unit has it, the
// bundled subset otherwise (both define the typedefs, enums and KALDI_ASSERT used below).
#ifdef KALDI_MATRIX_KALDI_MATRIX_H_
#define KALDI_HIP_ASSERT(cond) KALDI_ASSERT(cond)
#else
#include "kaldi-matrix-lite.h"
#endif

namespace kaldi {

inline void KhCheck(int rc) {
  if (rc != KH_OK) throw std::runtime_error(std::string("ERROR (libkaldi_hip) ") + kh_last_error());
}

// ---- CuDevice cu-device.h:41-143 ---------------------------------------------------
class CuDevice {
 public:
  static CuDevice &Instantiate() {
    static CuDevice d;
    return d;
  }
  /// "yes" | "no" | "optional" | "wait" (cu-device.cc:93-97).  There is no CPU
  /// fallback in this build: "no"/"optional" without a device throw when used.
  void SelectGpuId(const std::string &use_gpu) {
    if (use_gpu == "no") throw std::runtime_error("SelectGpuId(\"no\"): libkaldi_hip has no CPU path");
    KhCheck(kh_select_gpu(-1));
  }
  /// Explicit ordinal for one-process-per-GPU sharding.
  void SelectGpuId(int32 ordinal) { KhCheck(kh_select_gpu(ordinal)); active_gpu_id_ = ordinal; }
  bool Enabled() const { return kh_enabled() != 0; }
  std::string DeviceGetName() const {
    char buf[256];
    KhCheck(kh_device_name(buf, sizeof(buf)));
    return buf;
  }
  void *Malloc(size_t size) {
    void *p = kh_malloc(size);
    if (!p) KhCheck(KH_ENOMEM);
    return p;
  }
  void *MallocPitch(size_t row_bytes, size_t num_rows, size_t *pitch) {
    void *p = kh_malloc_pitch(row_bytes, num_rows, pitch);
    if (!p) KhCheck(KH_ENOMEM);
    return p;
  }
  void Free(void *ptr) { KhCheck(kh_free(ptr)); }
  /// ActiveGpuId() cu-device.h:67: -1 until a device is selected.
  int32 ActiveGpuId() const { return Enabled() ? active_gpu_id_ : -1; }
  /// GetFreeMemory(int64 *free, int64 *total) cu-device.cc:425-462
  std::string GetFreeMemory(int64_t *free_b = NULL, int64_t *total_b = NULL) const {
    size_t f = 0, t = 0;
    KhCheck(kh_mem_info(&f, &t));
    if (free_b) *free_b = static_cast<int64_t>(f);
    if (total_b) *total_b = static_cast<int64_t>(t);
    char buf[128];
    snprintf(buf, sizeof(buf), "free:%lldM, used:%lldM, total:%lldM, free/total:%g", (long long)(f >> 20),
             (long long)((t - f) >> 20), (long long)(t >> 20), t ? double(f) / double(t) : 0.0);
    return buf;
  }
  void PrintMemoryUsage() const { fprintf(stderr, "Memory used: %s\n", GetFreeMemory().c_str()); }  // :465
  /// CheckGpuHealth() cu-device.cc:509-527: a 50x100 by 100x50 product on the device against
  /// the same product on the host, relative difference < 1 % (AssertEqual(c, c1, 0.01)).
  void CheckGpuHealth() {
    if (!Enabled()) return;
    const int m = 50, k = 100, n = 50;
    std::vector<float> a(m * k), b(k * n), c(m * n, 0.f), c1(m * n, 0.f);
    unsigned s = 12345u;
    for (size_t i = 0; i < a.size(); i++) { s = s * 1664525u + 1013904223u; a[i] = (int(s >> 8) % 2001 - 1000) * 1e-3f; }
    for (size_t i = 0; i < b.size(); i++) { s = s * 1664525u + 1013904223u; b[i] = (s >> 8) % 1000 * 1e-3f; }
    for (int i = 0; i < m; i++)
      for (int j = 0; j < n; j++) {
        float acc = 0.f;
        for (int q = 0; q < k; q++) acc += a[i * k + q] * b[q * n + j];
        c[i * n + j] = acc;
      }
    float *da = static_cast<float *>(Malloc(a.size() * 4)), *db = static_cast<float *>(Malloc(b.size() * 4)),
          *dc = static_cast<float *>(Malloc(c.size() * 4));
    KhCheck(kh_memcpy_2d(da, k * 4, a.data(), k * 4, k * 4, m, 0));
    KhCheck(kh_memcpy_2d(db, n * 4, b.data(), n * 4, n * 4, k, 0));
    KhMatrixDim dA = {m, k, k}, dB = {k, n, n}, dC = {m, n, n};
    KhCheck(kh_add_mat_mat(1.0f, da, dA, 0, db, dB, 0, 0.0f, dc, dC));
    KhCheck(kh_memcpy_2d(c1.data(), n * 4, dc, n * 4, n * 4, m, 1));
    KhCheck(kh_synchronize());
    Free(da); Free(db); Free(dc);
    double diff = 0.0, norm = 0.0;
    for (size_t i = 0; i < c.size(); i++) { diff += double(c[i] - c1[i]) * (c[i] - c1[i]); norm += double(c[i]) * c[i]; }
    if (!(std::sqrt(diff) <= 0.01 * std::sqrt(norm))) throw std::runtime_error("CheckGpuHealth: device GEMM differs from the host GEMM");
  }
  /// DoublePrecisionSupported() cu-device.cc:407 (device capability >= 1.3 there): the matrix primitives exist for
  /// <double> (csrc/kh_double.hip, the cudaD_* twins); gfx950 has fp64 matrix cores.
  bool DoublePrecisionSupported() const { return true; }
  void SetVerbose(bool verbose) { verbose_ = verbose; }
  /// AccuProfile / PrintProfile / ResetProfile cu-device.cc:379-405
  void AccuProfile(const std::string &key, double time) { profile_map_[key] += time; }
  void PrintProfile() {
    if (!verbose_ || !Enabled()) return;
    fprintf(stderr, "-----\n[cudevice profile]\n");
    for (std::map<std::string, double>::const_iterator it = profile_map_.begin(); it != profile_map_.end(); ++it)
      fprintf(stderr, "%s\t%gs\n", it->first.c_str(), it->second);
    fprintf(stderr, "-----\n");
    PrintMemoryUsage();
  }
  void ResetProfile() { profile_map_.clear(); }

 private:
  CuDevice() : active_gpu_id_(0), verbose_(true) {}
  int32 active_gpu_id_;
  bool verbose_;
  std::map<std::string, double> profile_map_;
};

// ---- float / double dispatch -----------------------------------------------------------
// The classes below are templates on Real like the reference's (cu-matrix.h:62, cu-vector.h, cu-value.h),
// so code written against CuMatrix<BaseFloat> compiles unchanged.  Storage, copies, views and element access
// work for float and double, and so do the matrix primitives of the forward path (khx:: below: AddMatMat on the
// fp64 matrix cores, softmax, CopyRows, GroupPnorm, the element-wise set, cu::Splice).  What exists for float only
// (the fused Normalize, CompObjfAndDeriv - and everything above the matrix classes: KALDI_DOUBLEPRECISION=0 is
// what the decode path runs with) throws on a <double> object as KALDI_ERR would; nothing computes on the host.
inline float *KhF(float *p) { return p; }
inline const float *KhF(const float *p) { return p; }
inline float *KhF(double *) {
  throw std::runtime_error("ERROR (libkaldi_hip) this operation has no double-precision kernel: use CuMatrix<float> / "
                           "CuVector<float>");
}
inline const float *KhF(const double *p) { return KhF(const_cast<double *>(p)); }

// ---- float / double dispatch of the primitives that exist for both (csrc/kh_double.hip holds the <double> kernels:
// cu-matrix.cc:2415-2418 instantiates CuMatrix<double> too).  khx::op(...) picks kh_op or kh_op_d by the element type.
namespace khx {
inline int add_mat_mat(float alpha, const float *A, KhMatrixDim dA, int tA, const float *B, KhMatrixDim dB, int tB, float beta, float *C, KhMatrixDim dC) { return kh_add_mat_mat(alpha, A, dA, tA, B, dB, tB, beta, C, dC); }
inline int add_mat_mat(double alpha, const double *A, KhMatrixDim dA, int tA, const double *B, KhMatrixDim dB, int tB, double beta, double *C, KhMatrixDim dC) { return kh_add_mat_mat_d(alpha, A, dA, tA, B, dB, tB, beta, C, dC); }
inline int softmax_per_row(float *y, const float *x, KhMatrixDim d, int ss) { return kh_softmax_per_row(y, x, d, ss); }
inline int softmax_per_row(double *y, const double *x, KhMatrixDim d, int ss) { return kh_softmax_per_row_d(y, x, d, ss); }
inline int log_softmax_per_row(float *y, const float *x, KhMatrixDim d, int ss) { return kh_log_softmax_per_row(y, x, d, ss); }
inline int log_softmax_per_row(double *y, const double *x, KhMatrixDim d, int ss) { return kh_log_softmax_per_row_d(y, x, d, ss); }
inline int copy_rows(float *dst, KhMatrixDim dd, const float *src, int ss, const int32_t *idx) { return kh_copy_rows(dst, dd, src, ss, idx); }
inline int copy_rows(double *dst, KhMatrixDim dd, const double *src, int ss, const int32_t *idx) { return kh_copy_rows_d(dst, dd, src, ss, idx); }
inline int group_pnorm(float *y, const float *x, KhMatrixDim d, int ss, int g, float p) { return kh_group_pnorm(y, x, d, ss, g, p); }
inline int group_pnorm(double *y, const double *x, KhMatrixDim d, int ss, int g, double p) { return kh_group_pnorm_d(y, x, d, ss, g, p); }
inline int mul_rows_vec(float *M, KhMatrixDim d, const float *v) { return kh_mul_rows_vec(M, d, v); }
inline int mul_rows_vec(double *M, KhMatrixDim d, const double *v) { return kh_mul_rows_vec_d(M, d, v); }
inline int mul_cols_vec(float *M, KhMatrixDim d, const float *v) { return kh_mul_cols_vec(M, d, v); }
inline int mul_cols_vec(double *M, KhMatrixDim d, const double *v) { return kh_mul_cols_vec_d(M, d, v); }
inline int copy_rows_from_vec(float *M, KhMatrixDim d, const float *v) { return kh_copy_rows_from_vec(M, d, v); }
inline int copy_rows_from_vec(double *M, KhMatrixDim d, const double *v) { return kh_copy_rows_from_vec_d(M, d, v); }
inline int add_vec_to_rows(float alpha, const float *v, float beta, float *M, KhMatrixDim d) { return kh_add_vec_to_rows(alpha, v, beta, M, d); }
inline int add_vec_to_rows(double alpha, const double *v, double beta, double *M, KhMatrixDim d) { return kh_add_vec_to_rows_d(alpha, v, beta, M, d); }
inline int apply_floor(float *M, KhMatrixDim d, float f) { return kh_apply_floor(M, d, f); }
inline int apply_floor(double *M, KhMatrixDim d, double f) { return kh_apply_floor_d(M, d, f); }
inline int apply_log(float *M, KhMatrixDim d) { return kh_apply_log(M, d); }
inline int apply_log(double *M, KhMatrixDim d) { return kh_apply_log_d(M, d); }
inline int apply_exp(float *M, KhMatrixDim d) { return kh_apply_exp(M, d); }
inline int apply_exp(double *M, KhMatrixDim d) { return kh_apply_exp_d(M, d); }
inline int apply_pow(float *M, KhMatrixDim d, float p) { return kh_apply_pow(M, d, p); }
inline int apply_pow(double *M, KhMatrixDim d, double p) { return kh_apply_pow_d(M, d, p); }
inline int scale(float *M, KhMatrixDim d, float a) { return kh_scale(M, d, a); }
inline int scale(double *M, KhMatrixDim d, double a) { return kh_scale_d(M, d, a); }
inline int sum_column_ranges(float *y, KhMatrixDim d, const float *x, KhMatrixDim ds, const int32_t *r) { return kh_sum_column_ranges(y, d, x, ds, r); }
inline int sum_column_ranges(double *y, KhMatrixDim d, const double *x, KhMatrixDim ds, const int32_t *r) { return kh_sum_column_ranges_d(y, d, x, ds, r); }
inline int add_diag_mat2(float alpha, const float *M, KhMatrixDim d, float beta, float *v) { return kh_add_diag_mat2(alpha, M, d, beta, v); }
inline int add_diag_mat2(double alpha, const double *M, KhMatrixDim d, double beta, double *v) { return kh_add_diag_mat2_d(alpha, M, d, beta, v); }
inline int matrix_lookup(const float *M, KhMatrixDim d, const int32_t *pairs, int n, float *out) { return kh_matrix_lookup(M, d, pairs, n, out); }
inline int matrix_lookup(const double *M, KhMatrixDim d, const int32_t *pairs, int n, double *out) { return kh_matrix_lookup_d(M, d, pairs, n, out); }
inline int splice(float *y, KhMatrixDim dout, const float *x, KhMatrixDim din, const int32_t *off, int n) { return kh_splice(y, dout, x, din, off, n); }
inline int splice(double *y, KhMatrixDim dout, const double *x, KhMatrixDim din, const int32_t *off, int n) { return kh_splice_d(y, dout, x, din, off, n); }
}  // namespace khx

/// Int32Pair cudamatrix/cu-matrixdim.h:59-62 (the index pairs of SumColumnRanges and Lookup)
struct Int32Pair { int32 first, second; };

// ---- CuValue cu-value.h:33-81: the proxy CuMatrixBase::operator()(r, c) returns -----------------
template <typename Real>
class CuValue {
 public:
  explicit CuValue(Real *data) : data_(data) {}
  CuValue(const CuValue &o) : data_(o.data_) {}
  CuValue operator=(const CuValue<Real> &o) {
    KhCheck(kh_memcpy_2d(data_, sizeof(Real), o.data_, sizeof(Real), sizeof(Real), 1, 2));
    KhCheck(kh_synchronize());
    return *this;
  }
  Real operator=(Real r) {
    KhCheck(kh_memcpy_2d(data_, sizeof(Real), &r, sizeof(Real), sizeof(Real), 1, 0));
    KhCheck(kh_synchronize());
    return r;
  }
  Real operator+=(Real r) { return (*this = r + Real(*this)); }
  operator Real() const {
    Real v;
    KhCheck(kh_memcpy_2d(&v, sizeof(Real), data_, sizeof(Real), sizeof(Real), 1, 1));
    return v;
  }

 private:
  Real *data_;
};

// ---- CuArray cu-array.h:36-105 -------------------------------------------------------
template <typename T>
class CuArray {
 public:
  CuArray() : dim_(0), data_(NULL) {}
  explicit CuArray(MatrixIndexT dim) : dim_(0), data_(NULL) { Resize(dim); }
  explicit CuArray(const std::vector<T> &src) : dim_(0), data_(NULL) { CopyFromVec(src); }
  ~CuArray() { Destroy(); }
  MatrixIndexT Dim() const { return dim_; }
  const T *Data() const { return data_; }
  T *Data() { return data_; }
  void Resize(MatrixIndexT dim, MatrixResizeType t = kSetZero) {
    Destroy();
    if (dim > 0) {
      data_ = static_cast<T *>(CuDevice::Instantiate().Malloc(sizeof(T) * dim));
      if (t == kSetZero) KhCheck(kh_memset(data_, 0, sizeof(T) * dim));
    }
    dim_ = dim;
  }
  void CopyFromVec(const std::vector<T> &src) {
    Resize(static_cast<MatrixIndexT>(src.size()), kUndefined);
    if (dim_) KhCheck(kh_memcpy_2d(data_, sizeof(T) * dim_, src.data(), sizeof(T) * dim_, sizeof(T) * dim_, 1, 0));
  }
  void CopyToVec(std::vector<T> *dst) const {  // cu-array-inl.h:113-131
    dst->resize(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(dst->data(), sizeof(T) * dim_, data_, sizeof(T) * dim_, sizeof(T) * dim_, 1, 1));
  }
  void Destroy() {
    if (data_) kh_free(data_);
    data_ = NULL;
    dim_ = 0;
  }
  void Swap(CuArray<T> *o) { std::swap(data_, o->data_); std::swap(dim_, o->dim_); }

 private:
  CuArray(const CuArray &);
  CuArray &operator=(const CuArray &);
  MatrixIndexT dim_;
  T *data_;
};

// ---- CuVectorBase / CuVector / CuSubVector cu-vector.h ------------------------------------
template <typename Real> class CuSubVector;
template <typename Real> class CuMatrixBase;
template <typename Real>
class CuVectorBase {
 public:
  MatrixIndexT Dim() const { return dim_; }
  Real *Data() { return data_; }
  const Real *Data() const { return data_; }
  /// CopyFromVec(const VectorBase&) cu-vector.cc: same dimension required
  void CopyFromVec(const VectorBase<Real> &h) {
    KALDI_HIP_ASSERT(h.Dim() == dim_);
    const size_t b = sizeof(Real) * static_cast<size_t>(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(data_, b, h.Data(), b, b, 1, 0));
  }
  void CopyFromVec(const std::vector<Real> &h) {
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(h.size()) == dim_);
    const size_t b = sizeof(Real) * static_cast<size_t>(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(data_, b, h.data(), b, b, 1, 0));
  }
  void CopyFromVec(const CuVectorBase<Real> &v) {
    KALDI_HIP_ASSERT(v.dim_ == dim_);
    const size_t b = sizeof(Real) * static_cast<size_t>(dim_);
    if (dim_) { KhCheck(kh_memcpy_2d(data_, b, v.data_, b, b, 1, 2)); KhCheck(kh_synchronize()); }
  }
  void CopyToVec(VectorBase<Real> *h) const {
    KALDI_HIP_ASSERT(h->Dim() == dim_);
    const size_t b = sizeof(Real) * static_cast<size_t>(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(h->Data(), b, data_, b, b, 1, 1));
  }
  void CopyToVec(std::vector<Real> *h) const {
    h->resize(dim_);
    const size_t b = sizeof(Real) * static_cast<size_t>(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(h->data(), b, data_, b, b, 1, 1));
  }
  void SetZero() { if (dim_) KhCheck(kh_memset(data_, 0, sizeof(Real) * dim_)); }
  /// SetRandn cu-vector.cc (CuRand there): test support, drawn on the host and uploaded
  void SetRandn() { if (dim_) { Vector<Real> tmp(dim_, kUndefined); tmp.SetRandn(); CopyFromVec(tmp); } }
  CuValue<Real> operator()(MatrixIndexT i) { KALDI_HIP_ASSERT(i >= 0 && i < dim_); return CuValue<Real>(data_ + i); }
  Real operator()(MatrixIndexT i) const { KALDI_HIP_ASSERT(i >= 0 && i < dim_); return CuValue<Real>(data_ + i); }
  /// AddDiagMat2 cu-vector.cc:517-580: this = beta this + alpha diag(M M^T) (kNoTrans: the kernel NormalizeComponent uses) or
  /// diag(M^T M) (kTrans: through a transposed copy - not on the decode path)
  inline void AddDiagMat2(Real alpha, const CuMatrixBase<Real> &M, MatrixTransposeType trans, Real beta);
  inline CuSubVector<Real> Range(MatrixIndexT origin, MatrixIndexT length) const;  // cu-vector.h Range()

 protected:
  CuVectorBase() : data_(NULL), dim_(0) {}
  ~CuVectorBase() {}
  Real *data_;
  MatrixIndexT dim_;

 private:
  CuVectorBase(const CuVectorBase &);
  CuVectorBase &operator=(const CuVectorBase &);
};

template <typename Real>
class CuVector : public CuVectorBase<Real> {
 public:
  CuVector() {}
  explicit CuVector(MatrixIndexT dim, MatrixResizeType t = kSetZero) { Resize(dim, t); }
  explicit CuVector(const std::vector<Real> &host) { CopyFromVec(host); }
  explicit CuVector(const VectorBase<Real> &host) { Resize(host.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(host); }
  explicit CuVector(const CuVectorBase<Real> &v) { Resize(v.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(v); }
  CuVector(const CuVector<Real> &v) : CuVectorBase<Real>() { Resize(v.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(v); }
  CuVector<Real> &operator=(const CuVectorBase<Real> &v) { Resize(v.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(v); return *this; }
  CuVector<Real> &operator=(const CuVector<Real> &v) { Resize(v.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(v); return *this; }
  CuVector<Real> &operator=(const VectorBase<Real> &v) { Resize(v.Dim(), kUndefined); CuVectorBase<Real>::CopyFromVec(v); return *this; }
  ~CuVector() { if (this->data_) kh_free(this->data_); }
  void Resize(MatrixIndexT dim, MatrixResizeType t = kSetZero) {
    if (this->data_) kh_free(this->data_);
    this->data_ = NULL;
    this->dim_ = dim;
    if (dim > 0) {
      this->data_ = static_cast<Real *>(CuDevice::Instantiate().Malloc(sizeof(Real) * dim));
      if (t == kSetZero) KhCheck(kh_memset(this->data_, 0, sizeof(Real) * dim));
    }
  }
  void CopyFromVec(const std::vector<Real> &h) {  // the owning class resizes (CuVector(const VectorBase&))
    if (static_cast<MatrixIndexT>(h.size()) != this->dim_) Resize(static_cast<MatrixIndexT>(h.size()), kUndefined);
    CuVectorBase<Real>::CopyFromVec(h);
  }
  using CuVectorBase<Real>::CopyFromVec;
  void Swap(CuVector<Real> *o) { std::swap(this->data_, o->data_); std::swap(this->dim_, o->dim_); }
};

/// Non-owning view (cu-vector.h CuSubVector): a range of a vector or one row of a matrix.
template <typename Real>
class CuSubVector : public CuVectorBase<Real> {
 public:
  CuSubVector(const CuVectorBase<Real> &t, MatrixIndexT origin, MatrixIndexT length) {
    KALDI_HIP_ASSERT(origin >= 0 && length >= 0 && origin + length <= t.Dim());
    this->data_ = const_cast<Real *>(t.Data()) + origin;
    this->dim_ = length;
  }
  inline CuSubVector(const CuMatrixBase<Real> &mat, MatrixIndexT row);
  CuSubVector(const CuSubVector<Real> &o) : CuVectorBase<Real>() { this->data_ = o.data_; this->dim_ = o.dim_; }
};
template <typename Real>
inline CuSubVector<Real> CuVectorBase<Real>::Range(MatrixIndexT origin, MatrixIndexT length) const {
  return CuSubVector<Real>(*this, origin, length);
}

/// MatrixElement<Real> matrix/matrix-common.h:77-82 (the supervision labels of CompObjfAndDeriv)
template <typename Real>
struct MatrixElement { int32 row, column; Real weight; };

// ---- CuMatrixBase / CuMatrix / CuSubMatrix cu-matrix.h:62-644 (members data_, num_cols_, num_rows_,
// stride_ as cu-matrix.h:500-510).  Every operation lives in the base class and works on views as well:
// the library takes (pointer, rows, cols, stride). -------------------------------------------------
template <typename Real> class CuSubMatrix;
template <typename Real> class CuMatrix;
template <typename Real>
class CuMatrixBase {
 public:
  MatrixIndexT NumRows() const { return num_rows_; }
  MatrixIndexT NumCols() const { return num_cols_; }
  MatrixIndexT Stride() const { return stride_; }
  Real *Data() { return data_; }
  const Real *Data() const { return data_; }
  Real *RowData(MatrixIndexT r) { return data_ + static_cast<size_t>(r) * stride_; }
  const Real *RowData(MatrixIndexT r) const { return data_ + static_cast<size_t>(r) * stride_; }
  KhMatrixDim Dim() const { KhMatrixDim d = {num_rows_, num_cols_, stride_}; return d; }
  /// operator()(r, c) cu-matrix.h:465-482: a CuValue proxy (one element over the bus per access - for tests)
  CuValue<Real> operator()(MatrixIndexT r, MatrixIndexT c) {
    KALDI_HIP_ASSERT(r >= 0 && r < num_rows_ && c >= 0 && c < num_cols_);
    return CuValue<Real>(data_ + static_cast<size_t>(r) * stride_ + c);
  }
  Real operator()(MatrixIndexT r, MatrixIndexT c) const {
    KALDI_HIP_ASSERT(r >= 0 && r < num_rows_ && c >= 0 && c < num_cols_);
    return CuValue<Real>(data_ + static_cast<size_t>(r) * stride_ + c);
  }
  /// Range / RowRange / ColRange / Row cu-matrix.h:447-463
  inline CuSubMatrix<Real> Range(MatrixIndexT row_offset, MatrixIndexT num_rows, MatrixIndexT col_offset,
                                 MatrixIndexT num_cols) const;
  inline CuSubMatrix<Real> RowRange(MatrixIndexT row_offset, MatrixIndexT num_rows) const;
  inline CuSubMatrix<Real> ColRange(MatrixIndexT col_offset, MatrixIndexT num_cols) const;
  inline CuSubVector<Real> Row(MatrixIndexT r) const { return CuSubVector<Real>(*this, r); }

  /// CopyFromMat(const MatrixBase&) cu-matrix.cc:283-307: same size; the host matrix may be of the other precision
  template <typename Other>
  void CopyFromMat(const MatrixBase<Other> &src, MatrixTransposeType trans = kNoTrans) {
    if (sizeof(Other) == sizeof(Real) && trans == kNoTrans) {
      CopyFromMat(reinterpret_cast<const Real *>(src.Data()), src.NumRows(), src.NumCols(), src.Stride());
    } else {
      Matrix<Real> tmp(src, trans);
      CopyFromMat(tmp.Data(), tmp.NumRows(), tmp.NumCols(), tmp.Stride());
    }
  }
  /// the same from a raw host pointer: row-major, stride in elements
  void CopyFromMat(const Real *host, MatrixIndexT rows, MatrixIndexT cols, MatrixIndexT host_stride) {
    KALDI_HIP_ASSERT(rows == num_rows_ && cols == num_cols_);
    if (rows) KhCheck(kh_memcpy_2d(data_, sizeof(Real) * (size_t)stride_, host, sizeof(Real) * (size_t)host_stride, sizeof(Real) * (size_t)cols, rows, 0));
  }
  /// CopyToMat cu-matrix.cc:387-412
  template <typename Other>
  void CopyToMat(MatrixBase<Other> *dst, MatrixTransposeType trans = kNoTrans) const {
    if (sizeof(Other) == sizeof(Real) && trans == kNoTrans) {
      KALDI_HIP_ASSERT(dst->NumRows() == num_rows_ && dst->NumCols() == num_cols_);
      CopyToMat(reinterpret_cast<Real *>(dst->Data()), dst->Stride());
    } else {
      Matrix<Real> tmp(num_rows_, num_cols_, kUndefined);
      CopyToMat(tmp.Data(), tmp.Stride());
      dst->CopyFromMat(tmp, trans);
    }
  }
  void CopyToMat(Real *host, MatrixIndexT host_stride) const {
    if (num_rows_) KhCheck(kh_memcpy_2d(host, sizeof(Real) * (size_t)host_stride, data_, sizeof(Real) * (size_t)stride_, sizeof(Real) * (size_t)num_cols_, num_rows_, 1));
  }
  void CopyFromMat(const CuMatrixBase<Real> &src) {  // cu-matrix.cc:197-231, same size
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    if (num_rows_) {
      KhCheck(kh_memcpy_2d(data_, sizeof(Real) * (size_t)stride_, src.data_, sizeof(Real) * (size_t)src.stride_, sizeof(Real) * (size_t)num_cols_, num_rows_, 2));
      Sync();
    }
  }
  void SetZero() {
    for (MatrixIndexT r = 0; r < num_rows_ && stride_ != num_cols_; r++)
      KhCheck(kh_memset(data_ + static_cast<size_t>(r) * stride_, 0, sizeof(Real) * num_cols_));
    if (stride_ == num_cols_ && num_rows_) KhCheck(kh_memset(data_, 0, sizeof(Real) * num_cols_ * num_rows_));
  }
  /// SetRandn cu-matrix.cc:1911-1918 (CuRand in the reference).  Test support, not on the decode path: the
  /// normals are drawn on the host and uploaded.
  void SetRandn() {
    if (num_rows_ == 0) return;
    Matrix<Real> tmp(num_rows_, num_cols_, kUndefined);
    tmp.SetRandn();
    CopyFromMat(tmp);
  }
  /// Set / Add / InvertElements / MulElements / Sum (cu-matrix.cc): NOT operations of the decode path - test support so that the
  /// reference's unit tests read unchanged; they run through a host copy.
  void Set(Real v) { if (num_rows_) { Matrix<Real> h(num_rows_, num_cols_, kUndefined); h.Set(v); CopyFromMat(h); } }
  void Add(Real v) { if (num_rows_) { Matrix<Real> h(*this); h.Add(v); CopyFromMat(h); } }
  void InvertElements() { if (num_rows_) { Matrix<Real> h(*this); h.InvertElements(); CopyFromMat(h); } }
  void MulElements(const CuMatrixBase<Real> &A) { if (num_rows_) { Matrix<Real> h(*this), a(A); h.MulElements(a); CopyFromMat(h); } }
  Real Sum() const { Matrix<Real> h(*this); return h.Sum(); }
  /// FrobeniusNorm / ApproxEqual cu-matrix.cc:1606-1611.  Test support: compared on host copies.
  Real FrobeniusNorm() const { Matrix<Real> h(*this); return h.FrobeniusNorm(); }
  bool ApproxEqual(const CuMatrixBase<Real> &other, float tol = 0.01) const {
    Matrix<Real> a(*this), b(other);
    return a.ApproxEqual(b, tol);
  }

  // ---- forward-path operations; each = the reference method of the same name
  void AddMatMat(Real alpha, const CuMatrixBase<Real> &A, MatrixTransposeType transA, const CuMatrixBase<Real> &B,
                 MatrixTransposeType transB, Real beta) {  // cu-matrix.cc:947-982
    KhCheck(khx::add_mat_mat(static_cast<Real>(alpha), A.data_, A.Dim(), transA == kTrans, B.data_, B.Dim(),
                           transB == kTrans, static_cast<Real>(beta), data_, Dim()));
    Sync();
  }
  void ApplySoftMaxPerRow(const CuMatrixBase<Real> &src) {  // :1251-1271
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    KhCheck(khx::softmax_per_row(data_, src.data_, Dim(), src.stride_));
    Sync();
  }
  void ApplyLogSoftMaxPerRow(const CuMatrixBase<Real> &src) {  // :1274-1295
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    KhCheck(khx::log_softmax_per_row(data_, src.data_, Dim(), src.stride_));
    Sync();
  }
  void CopyRows(const CuMatrixBase<Real> &src, const std::vector<MatrixIndexT> &indices) {  // :1965-1990
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(indices.size()) == num_rows_ && src.num_cols_ == num_cols_);
    CuArray<MatrixIndexT> idx(indices);  // the reference uploads the index vector per call too (:1976)
    KhCheck(khx::copy_rows(data_, Dim(), src.data_, src.stride_, idx.Data()));
    Sync();
  }
  void GroupPnorm(const CuMatrixBase<Real> &src, Real power) {  // :1147-1164
    KALDI_HIP_ASSERT(num_cols_ > 0 && src.num_cols_ % num_cols_ == 0 && src.num_rows_ == num_rows_);
    KhCheck(khx::group_pnorm(data_, src.data_, Dim(), src.stride_, src.num_cols_ / num_cols_, static_cast<Real>(power)));
    Sync();
  }
  void MulRowsVec(const CuVectorBase<Real> &scale) {  // :693-713
    KALDI_HIP_ASSERT(scale.Dim() == num_rows_);
    KhCheck(khx::mul_rows_vec(data_, Dim(), scale.Data()));
    Sync();
  }
  void MulColsVec(const CuVectorBase<Real> &scale) {  // :668
    KALDI_HIP_ASSERT(scale.Dim() == num_cols_);
    KhCheck(khx::mul_cols_vec(data_, Dim(), scale.Data()));
    Sync();
  }
  void CopyRowsFromVec(const CuVectorBase<Real> &v) {  // :1673-1745: NumCols() entries -> every row; NumRows() * NumCols() -> the matrix
    if (v.Dim() == num_cols_) {
      KhCheck(khx::copy_rows_from_vec(data_, Dim(), v.Data()));
    } else {
      KALDI_HIP_ASSERT(v.Dim() == num_rows_ * num_cols_);
      if (num_rows_) KhCheck(kh_memcpy_2d(data_, sizeof(Real) * (size_t)stride_, v.Data(), sizeof(Real) * (size_t)num_cols_, sizeof(Real) * (size_t)num_cols_, num_rows_, 2));
    }
    Sync();
  }
  void AddVecToRows(Real alpha, const CuVectorBase<Real> &row, Real beta = 1.0) {  // :916-939
    KALDI_HIP_ASSERT(row.Dim() == num_cols_);
    KhCheck(khx::add_vec_to_rows(static_cast<Real>(alpha), row.Data(), static_cast<Real>(beta), data_, Dim()));
    Sync();
  }
  void ApplyFloor(Real f) { KhCheck(khx::apply_floor(data_, Dim(), static_cast<Real>(f))); Sync(); }   // :1845
  void ApplyLog() { KhCheck(khx::apply_log(data_, Dim())); Sync(); }                                     // :600
  void ApplyExp() { KhCheck(khx::apply_exp(data_, Dim())); Sync(); }
  void ApplyPow(Real p) { KhCheck(khx::apply_pow(data_, Dim(), static_cast<Real>(p))); Sync(); }
  void Scale(Real a) { KhCheck(khx::scale(data_, Dim(), static_cast<Real>(a))); Sync(); }               // :579
  void SumColumnRanges(const CuMatrixBase<Real> &src, const CuArray<Int32Pair> &indices) {  // :1994-2028, the reference's signature
    KALDI_HIP_ASSERT(indices.Dim() == num_cols_ && src.num_rows_ == num_rows_);
    KhCheck(khx::sum_column_ranges(data_, Dim(), src.data_, src.Dim(), reinterpret_cast<const int32 *>(indices.Data())));
    Sync();
  }
  void SumColumnRanges(const CuMatrixBase<Real> &src, const std::vector<int32> &start_end_pairs) {
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(start_end_pairs.size()) == 2 * num_cols_ && src.num_rows_ == num_rows_);
    CuArray<int32> r(start_end_pairs);
    KhCheck(khx::sum_column_ranges(data_, Dim(), src.data_, src.Dim(), r.Data()));
    Sync();
  }
  /// this <- NormalizeComponent::Propagate(src) (nnet2/nnet-component.cc:576-588) in one kernel
  void NormalizePerRow(const CuMatrixBase<Real> &src) {
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    KhCheck(kh_normalize(KhF(data_), KhF(src.data_), Dim(), src.stride_));
    Sync();
  }
  /// v.AddDiagMat2(alpha, *this, kNoTrans, beta) cu-vector.cc:517-580: v = beta v + alpha diag(M M^T)
  void AddDiagMat2To(CuVectorBase<Real> *v, Real alpha, Real beta) const {
    KALDI_HIP_ASSERT(v->Dim() == num_rows_);
    KhCheck(khx::add_diag_mat2(static_cast<Real>(alpha), data_, Dim(), static_cast<Real>(beta), v->Data()));
    Sync();
  }
  /// CompObjfAndDeriv cu-matrix.cc:1198-1248 on *this = the derivative; labels = (row, column, weight)
  void CompObjfAndDeriv(const std::vector<MatrixElement<Real> > &sv_labels, const CuMatrixBase<Real> &output, Real *tot_objf,
                        Real *tot_weight) {
    std::vector<int32> r(sv_labels.size()), c(sv_labels.size());
    std::vector<float> w(sv_labels.size());
    for (size_t i = 0; i < sv_labels.size(); i++) { r[i] = sv_labels[i].row; c[i] = sv_labels[i].column; w[i] = static_cast<float>(sv_labels[i].weight); }
    float objf = 0.f, weight = 0.f;
    KhCheck(kh_comp_objf_and_deriv(static_cast<int>(r.size()), r.data(), c.data(), w.data(), KhF(output.data_), output.Dim(),
                                   KhF(data_), Dim(), &objf, &weight));
    *tot_objf = objf;
    *tot_weight = weight;
  }
  void Lookup(const std::vector<Int32Pair> &indices, std::vector<Real> *output) const {  // :2327, the reference's signature
    std::vector<int32> flat(2 * indices.size());
    for (size_t i = 0; i < indices.size(); i++) {
      KALDI_HIP_ASSERT(indices[i].first >= 0 && indices[i].first < num_rows_ && indices[i].second >= 0 && indices[i].second < num_cols_);
      flat[2 * i] = indices[i].first;
      flat[2 * i + 1] = indices[i].second;
    }
    Lookup(flat, output);
  }
  void Lookup(const std::vector<int32> &row_col_pairs, std::vector<Real> *output) const {
    const int n = static_cast<int>(row_col_pairs.size() / 2);
    output->resize(n);
    if (!n) return;
    CuArray<int32> idx(row_col_pairs);
    CuVector<Real> out(n);
    KhCheck(khx::matrix_lookup(data_, Dim(), idx.Data(), n, out.Data()));
    out.CopyToVec(output);
  }

 protected:
  CuMatrixBase() : data_(NULL), num_cols_(0), num_rows_(0), stride_(0) {}
  CuMatrixBase(Real *data, MatrixIndexT rows, MatrixIndexT cols, MatrixIndexT stride)
      : data_(data), num_cols_(cols), num_rows_(rows), stride_(stride) {}
  ~CuMatrixBase() {}
  static void Sync() { KhCheck(kh_synchronize()); }
  Real *data_;
  MatrixIndexT num_cols_, num_rows_, stride_;

 private:
  CuMatrixBase(const CuMatrixBase &);
  CuMatrixBase &operator=(const CuMatrixBase &);
};

template <typename Real>
class CuMatrix : public CuMatrixBase<Real> {
 public:
  CuMatrix() {}
  CuMatrix(MatrixIndexT rows, MatrixIndexT cols, MatrixResizeType t = kSetZero) { Resize(rows, cols, t); }
  /// copy constructors cu-matrix.h:536-551: from a device matrix, from a host matrix (either precision)
  CuMatrix(const CuMatrix<Real> &o) : CuMatrixBase<Real>() { Resize(o.NumRows(), o.NumCols(), kUndefined); CuMatrixBase<Real>::CopyFromMat(o); }
  explicit CuMatrix(const CuMatrixBase<Real> &o, MatrixTransposeType trans = kNoTrans) {
    if (trans == kNoTrans) {
      Resize(o.NumRows(), o.NumCols(), kUndefined);
      CuMatrixBase<Real>::CopyFromMat(o);
    } else {   // (a transposed copy is not an operation of the decode path: through the host)
      Matrix<Real> h(o, kTrans);
      Resize(h.NumRows(), h.NumCols(), kUndefined);
      CuMatrixBase<Real>::CopyFromMat(h);
    }
  }
  template <typename Other>
  explicit CuMatrix(const MatrixBase<Other> &o, MatrixTransposeType trans = kNoTrans) {
    if (trans == kNoTrans) Resize(o.NumRows(), o.NumCols(), kUndefined); else Resize(o.NumCols(), o.NumRows(), kUndefined);
    CuMatrixBase<Real>::CopyFromMat(o, trans);
  }
  CuMatrix<Real> &operator=(const CuMatrixBase<Real> &o) { Resize(o.NumRows(), o.NumCols(), kUndefined); CuMatrixBase<Real>::CopyFromMat(o); return *this; }
  CuMatrix<Real> &operator=(const CuMatrix<Real> &o) { Resize(o.NumRows(), o.NumCols(), kUndefined); CuMatrixBase<Real>::CopyFromMat(o); return *this; }
  CuMatrix<Real> &operator=(const MatrixBase<Real> &o) { Resize(o.NumRows(), o.NumCols(), kUndefined); CuMatrixBase<Real>::CopyFromMat(o); return *this; }
  ~CuMatrix() { Destroy(); }
  void Resize(MatrixIndexT rows, MatrixIndexT cols, MatrixResizeType t = kSetZero) {  // cu-matrix.cc:47-100
    KALDI_HIP_ASSERT(rows >= 0 && cols >= 0);
    if (rows == this->num_rows_ && cols == this->num_cols_ && this->data_ != NULL) {  // :56-59 "nothing to do"
      if (t == kSetZero) this->SetZero();
      return;
    }
    Destroy();
    if (rows == 0 || cols == 0) return;
    size_t pitch;
    this->data_ = static_cast<Real *>(CuDevice::Instantiate().MallocPitch(sizeof(Real) * cols, rows, &pitch));
    this->num_rows_ = rows;
    this->num_cols_ = cols;
    this->stride_ = static_cast<MatrixIndexT>(pitch / sizeof(Real));
    if (t == kSetZero) KhCheck(kh_memset(this->data_, 0, pitch * rows));
  }
  void Destroy() {
    if (this->data_) kh_free(this->data_);
    this->data_ = NULL;
    this->num_rows_ = this->num_cols_ = this->stride_ = 0;
  }
  /// the owning class resizes to the source (raw-pointer form; CuMatrix(const MatrixBase&) above)
  void CopyFromMat(const Real *host, MatrixIndexT rows, MatrixIndexT cols, MatrixIndexT host_stride) {
    if (rows != this->num_rows_ || cols != this->num_cols_) Resize(rows, cols, kUndefined);
    CuMatrixBase<Real>::CopyFromMat(host, rows, cols, host_stride);
  }
  using CuMatrixBase<Real>::CopyFromMat;
  void Swap(CuMatrix<Real> *o) {
    std::swap(this->data_, o->data_); std::swap(this->num_cols_, o->num_cols_);
    std::swap(this->num_rows_, o->num_rows_); std::swap(this->stride_, o->stride_);
  }
  /// Swap(Matrix<Real>*) cu-matrix.cc:128-152: exchanges contents with a host matrix
  void Swap(Matrix<Real> *mat) {
    Matrix<Real> mine(this->num_rows_, this->num_cols_, kUndefined);
    if (this->num_rows_) this->CopyToMat(&mine);
    if (mat->NumRows()) { Resize(mat->NumRows(), mat->NumCols(), kUndefined); CuMatrixBase<Real>::CopyFromMat(*mat); } else Destroy();
    mine.Swap(mat);
  }
};

/// Non-owning view of a block of another matrix (cu-matrix.h:620-644).
template <typename Real>
class CuSubMatrix : public CuMatrixBase<Real> {
 public:
  CuSubMatrix(const CuMatrixBase<Real> &mat, MatrixIndexT row_offset, MatrixIndexT num_rows, MatrixIndexT col_offset,
              MatrixIndexT num_cols)
      : CuMatrixBase<Real>(const_cast<Real *>(mat.Data()) + static_cast<size_t>(row_offset) * mat.Stride() + col_offset,
                           num_rows, num_cols, mat.Stride()) {
    KALDI_HIP_ASSERT(row_offset >= 0 && col_offset >= 0 && num_rows >= 0 && num_cols >= 0 &&
                     row_offset + num_rows <= mat.NumRows() && col_offset + num_cols <= mat.NumCols());
  }
  CuSubMatrix(const CuSubMatrix<Real> &o) : CuMatrixBase<Real>(o.data_, o.num_rows_, o.num_cols_, o.stride_) {}
};
template <typename Real>
inline CuSubMatrix<Real> CuMatrixBase<Real>::Range(MatrixIndexT ro, MatrixIndexT nr, MatrixIndexT co, MatrixIndexT nc) const {
  return CuSubMatrix<Real>(*this, ro, nr, co, nc);
}
template <typename Real>
inline CuSubMatrix<Real> CuMatrixBase<Real>::RowRange(MatrixIndexT ro, MatrixIndexT nr) const { return CuSubMatrix<Real>(*this, ro, nr, 0, num_cols_); }
template <typename Real>
inline CuSubMatrix<Real> CuMatrixBase<Real>::ColRange(MatrixIndexT co, MatrixIndexT nc) const { return CuSubMatrix<Real>(*this, 0, num_rows_, co, nc); }
template <typename Real>
inline CuSubVector<Real>::CuSubVector(const CuMatrixBase<Real> &mat, MatrixIndexT row) {
  KALDI_HIP_ASSERT(row >= 0 && row < mat.NumRows());
  this->data_ = const_cast<Real *>(mat.Data()) + static_cast<size_t>(row) * mat.Stride();
  this->dim_ = mat.NumCols();
}

template <typename Real>
inline void CuVectorBase<Real>::AddDiagMat2(Real alpha, const CuMatrixBase<Real> &M, MatrixTransposeType trans, Real beta) {
  if (trans == kNoTrans) {
    M.AddDiagMat2To(this, alpha, beta);
  } else {
    CuMatrix<Real> Mt(M, kTrans);
    Mt.AddDiagMat2To(this, alpha, beta);
  }
}
template <typename Real>
inline void AssertEqual(const CuVectorBase<Real> &a, const CuVectorBase<Real> &b, float tol = 0.01) {
  Vector<Real> ha(a), hb(b);
  KALDI_HIP_ASSERT(ha.ApproxEqual(hb, tol));
}

/// TraceMatMat(A, B, trans) cu-matrix.cc:1613-1650 and ApproxEqual of device matrices: test support through host copies
template <typename Real>
inline Real TraceMatMat(const CuMatrixBase<Real> &A, const CuMatrixBase<Real> &B, MatrixTransposeType trans = kNoTrans) {
  Matrix<Real> a(A), b(B);
  return TraceMatMat(a, b, trans);
}
template <typename Real>
inline bool ApproxEqual(const CuMatrixBase<Real> &A, const CuMatrixBase<Real> &B, float tol = 0.01) { return A.ApproxEqual(B, tol); }

/// AssertEqual / SameDim cu-matrix.h:653-668
template <typename Real>
inline void AssertEqual(CuMatrixBase<Real> &A, CuMatrixBase<Real> &B, float tol = 0.01) { KALDI_HIP_ASSERT(A.ApproxEqual(B, tol)); }
template <typename Real>
inline bool SameDim(const CuMatrixBase<Real> &M, const CuMatrixBase<Real> &N) { return M.NumRows() == N.NumRows() && M.NumCols() == N.NumCols(); }
template <typename Real>
inline bool SameDimAndStride(const CuMatrixBase<Real> &M, const CuMatrixBase<Real> &N) { return SameDim(M, N) && M.Stride() == N.Stride(); }

namespace cu {
/// cu::Splice cudamatrix/cu-math.cc:130-165 with the reference's signature (the offsets already on the device)
template <typename Real>
inline void Splice(const CuMatrixBase<Real> &src, const CuArray<int32> &frame_offsets, CuMatrixBase<Real> *tgt) {
  KALDI_HIP_ASSERT(src.NumCols() * frame_offsets.Dim() == tgt->NumCols() && src.NumRows() == tgt->NumRows());
  KhCheck(khx::splice(tgt->Data(), tgt->Dim(), src.Data(), src.Dim(), frame_offsets.Data(), frame_offsets.Dim()));
  KhCheck(kh_synchronize());
}
/// the same from a host vector of offsets
template <typename Real>
inline void Splice(const CuMatrixBase<Real> &src, const std::vector<int32> &frame_offsets, CuMatrixBase<Real> *tgt) {
  KALDI_HIP_ASSERT(src.NumCols() * static_cast<int>(frame_offsets.size()) == tgt->NumCols() &&
                   src.NumRows() == tgt->NumRows());
  CuArray<int32> off(frame_offsets);
  KhCheck(khx::splice(tgt->Data(), tgt->Dim(), src.Data(), src.Dim(), off.Data(), off.Dim()));
  KhCheck(kh_synchronize());
}
}  // namespace cu

// ---- nnet2::Nnet (forward only) + NnetComputation + DecodableAmNnet ----------------------
/// Components are described by KhComponentDesc (host parameter pointers, copied at Add).
class Nnet {
 public:
  Nnet() : nnet_(kh_nnet_create()) { if (!nnet_) KhCheck(KH_ENOMEM); }
  ~Nnet() { kh_nnet_destroy(nnet_); }
  void AddComponent(const KhComponentDesc &desc) { KhCheck(kh_nnet_add_component(nnet_, &desc)); }
  void SetPriors(const std::vector<BaseFloat> &priors) {  // AmNnet::SetPriors am-nnet.h
    KhCheck(kh_nnet_set_priors(nnet_, priors.data(), static_cast<int>(priors.size())));
  }
  int32 NumComponents() const { return kh_nnet_num_components(nnet_); }
  int32 InputDim() const { return kh_nnet_input_dim(nnet_); }
  int32 OutputDim() const { return kh_nnet_output_dim(nnet_); }
  int32 LeftContext() const { return kh_nnet_left_context(nnet_); }    // nnet-nnet.cc:45
  int32 RightContext() const { return kh_nnet_right_context(nnet_); }  // :55
  KhNnet *Handle() const { return nnet_; }

 private:
  Nnet(const Nnet &);
  Nnet &operator=(const Nnet &);
  KhNnet *nnet_;
};

/// NnetComputation(nnet, input, pad_input, &output) nnet2/nnet-compute.cc:159-166 for one
/// utterance (utt_row_offsets = {0, T}) or a batch stacked by rows.
inline void NnetComputation(const Nnet &nnet, const CuMatrixBase<BaseFloat> &input, bool pad_input, CuMatrix<BaseFloat> *output,
                            const std::vector<int32> *utt_row_offsets = NULL) {
  std::vector<int32> one;
  if (!utt_row_offsets) { one.push_back(0); one.push_back(input.NumRows()); utt_row_offsets = &one; }
  const int n_utts = static_cast<int>(utt_row_offsets->size()) - 1;
  const int rows = pad_input ? input.NumRows() : input.NumRows() - n_utts * (nnet.LeftContext() + nnet.RightContext());
  KALDI_HIP_ASSERT(rows > 0);
  output->Resize(rows, nnet.OutputDim(), kUndefined);
  KhCheck(kh_nnet_compute(nnet.Handle(), input.Data(), input.Stride(), utt_row_offsets->data(), n_utts, pad_input, 0,
                          1.0f, output->Data(), output->Stride(), NULL));
}

/// DecodableAmNnet's matrix (nnet2/decodable-am-nnet.h:47-69): floor, log, -log prior, scale.
inline void ComputeScaledLogLikes(const Nnet &nnet, const CuMatrixBase<BaseFloat> &feats, bool pad_input, BaseFloat prob_scale,
                                  CuMatrix<BaseFloat> *log_probs, const std::vector<int32> *utt_row_offsets = NULL) {
  std::vector<int32> one;
  if (!utt_row_offsets) { one.push_back(0); one.push_back(feats.NumRows()); utt_row_offsets = &one; }
  const int n_utts = static_cast<int>(utt_row_offsets->size()) - 1;
  const int rows = pad_input ? feats.NumRows() : feats.NumRows() - n_utts * (nnet.LeftContext() + nnet.RightContext());
  KALDI_HIP_ASSERT(rows > 0);
  log_probs->Resize(rows, nnet.OutputDim(), kUndefined);
  KhCheck(kh_nnet_compute(nnet.Handle(), feats.Data(), feats.Stride(), utt_row_offsets->data(), n_utts, pad_input, 1,
                          prob_scale, log_probs->Data(), log_probs->Stride(), NULL));
}

// ---- DiagGmm gmm/diag-gmm.h:83-135 (likelihood side) -------------------------------------
class DiagGmm {
 public:
  /// weights [M], means_invvars / inv_vars [M x D] row-major host arrays; ComputeGconsts :114-152
  DiagGmm(const std::vector<BaseFloat> &weights, const std::vector<BaseFloat> &means_invvars,
          const std::vector<BaseFloat> &inv_vars, int32 dim)
      : num_mix_(static_cast<int32>(weights.size())), dim_(dim) {
    std::vector<BaseFloat> g(num_mix_);
    int bad = kh_gmm_compute_gconsts(weights.data(), means_invvars.data(), inv_vars.data(), num_mix_, dim, g.data());
    if (bad < 0) KhCheck(bad);
    gconsts_.CopyFromVec(g);
    means_invvars_.CopyFromVec(means_invvars);  // packed [M x D], as the C-ABI takes them
    inv_vars_.CopyFromVec(inv_vars);
  }
  int32 NumGauss() const { return num_mix_; }
  int32 Dim() const { return dim_; }
  /// LogLikelihoods(const MatrixBase &data, Matrix *loglikes) diag-gmm.cc:546-562
  void LogLikelihoods(const CuMatrixBase<BaseFloat> &data, CuMatrix<BaseFloat> *loglikes) const {
    KALDI_HIP_ASSERT(data.NumCols() == dim_);
    loglikes->Resize(data.NumRows(), num_mix_, kUndefined);
    KhCheck(kh_diag_gmm_loglikes(data.Data(), data.Dim(), gconsts_.Data(), means_invvars_.Data(), inv_vars_.Data(),
                                 num_mix_, loglikes->Data(), loglikes->Stride()));
    KhCheck(kh_synchronize());
  }

 private:
  int32 num_mix_, dim_;
  CuVector<BaseFloat> gconsts_, means_invvars_, inv_vars_;
};

// ---- lattice functions lat/lattice-functions.cc ---------------------------------------------
/// A top-sorted Lattice flattened to CSR (state 0 = start): what ArcIterator yields.
struct LatticeCsr {
  std::vector<int64_t> arc_offsets;                       // num_states + 1
  std::vector<int32> arc_ilabel, arc_nextstate;
  std::vector<BaseFloat> arc_graph, arc_acoustic, state_final;  // final = +inf for non-final states
  int32 NumStates() const { return static_cast<int32>(state_final.size()); }
};
/// LatticeForwardBackward :272-354: arc posteriors, returns tot_backward_prob.
inline double LatticeForwardBackward(const LatticeCsr &lat, std::vector<BaseFloat> *arc_post,
                                     double *acoustic_like_sum = NULL, std::vector<int32> *state_times = NULL) {
  const int32 soff[2] = {0, lat.NumStates()};
  arc_post->resize(lat.arc_ilabel.size());
  if (state_times) state_times->resize(lat.NumStates());
  double tot = 0.0;
  KhCheck(kh_lattice_forward_backward(1, soff, lat.arc_offsets.data(), lat.arc_ilabel.data(), lat.arc_nextstate.data(),
                                      lat.arc_graph.data(), lat.arc_acoustic.data(), lat.state_final.data(),
                                      arc_post->data(), &tot, acoustic_like_sum, state_times ? state_times->data() : NULL));
  return tot;
}
/// LatticeForwardBackwardMpeVariants :740-919; returns tot_forward_score.
inline double LatticeForwardBackwardMpeVariants(const std::vector<int32> &tid2phone, const std::vector<int32> &tid2pdf,
                                                const std::vector<int32> &silence_phones, const LatticeCsr &lat,
                                                const std::vector<int32> &num_ali, const std::string &criterion,
                                                bool one_silence_class, std::vector<BaseFloat> *arc_post) {
  KALDI_HIP_ASSERT(criterion == "mpfe" || criterion == "smbr");
  const int32 soff[2] = {0, lat.NumStates()}, aoff[2] = {0, static_cast<int32>(num_ali.size())};
  arc_post->resize(lat.arc_ilabel.size());
  double score = 0.0;
  KhCheck(kh_lattice_forward_backward_mpe(1, soff, lat.arc_offsets.data(), lat.arc_ilabel.data(),
                                          lat.arc_nextstate.data(), lat.arc_graph.data(), lat.arc_acoustic.data(),
                                          lat.state_final.data(), tid2phone.data(), tid2pdf.data(),
                                          static_cast<int>(tid2phone.size()) - 1, silence_phones.data(),
                                          static_cast<int>(silence_phones.size()), num_ali.data(), aoff,
                                          criterion == "mpfe", one_silence_class, arc_post->data(), &score, NULL));
  return score;
}

// ---- feature front-end feat/feature-mfcc.h, feature-functions.h, transform/cmvn.h ---------
struct MfccOptions {  // feature-mfcc.h:37-78 + FrameExtractionOptions + MelBanksOptions (the supported subset)
  BaseFloat samp_freq, frame_length_ms, frame_shift_ms, preemph_coeff;
  bool remove_dc_offset;
  std::string window_type;
  int32 num_bins, num_ceps;
  BaseFloat low_freq, high_freq, cepstral_lifter;
  // defaults here are the recipes' (conf/mfcc.conf: --use-energy=false, no dither); the reference's struct
  // defaults are use_energy = true, dither = 1.0 (feature-mfcc.h:54, feature-functions.h:91)
  bool snip_edges, use_energy, raw_energy, htk_compat;
  BaseFloat energy_floor, dither;
  uint64_t dither_seed;  // the reference draws from rand(); here a counter-based generator
  MfccOptions()
      : samp_freq(16000), frame_length_ms(25.0), frame_shift_ms(10.0), preemph_coeff(0.97), remove_dc_offset(true),
        window_type("povey"), num_bins(23), num_ceps(13), low_freq(20), high_freq(0), cepstral_lifter(22.0),
        snip_edges(true), use_energy(false), raw_energy(true), htk_compat(false), energy_floor(0.0), dither(0.0),
        dither_seed(0) {}
};

/// Mfcc: the constructor builds the window function, the mel filters, the DCT rows and the
/// lifter as the reference's does.
class Mfcc {
 public:
  explicit Mfcc(const MfccOptions &opts) : opts_(opts) {
    frame_shift_ = static_cast<int32>(opts.samp_freq * 0.001f * opts.frame_shift_ms);
    frame_length_ = static_cast<int32>(opts.samp_freq * 0.001f * opts.frame_length_ms);
    padded_ = 1;
    while (padded_ < frame_length_) padded_ <<= 1;
    const double two_pi = 6.283185307179586476925286766559, pi = 3.1415926535897932384626433832795;
    window_.resize(frame_length_);
    for (int32 i = 0; i < frame_length_; i++) {
      const double a = two_pi * static_cast<BaseFloat>(i) / (frame_length_ - 1);
      if (opts.window_type == "hanning") window_[i] = 0.5 - 0.5 * std::cos(a);
      else if (opts.window_type == "hamming") window_[i] = 0.54 - 0.46 * std::cos(a);
      else if (opts.window_type == "povey") window_[i] = std::pow(0.5 - 0.5 * std::cos(a), 0.85);
      else if (opts.window_type == "rectangular") window_[i] = 1.0;
      else throw std::runtime_error("Invalid window type " + opts.window_type);
    }
    const BaseFloat nyquist = 0.5f * opts.samp_freq;
    const BaseFloat high = opts.high_freq > 0.0f ? opts.high_freq : nyquist + opts.high_freq;
    if (opts.low_freq < 0.0f || opts.low_freq >= nyquist || high <= 0.0f || high > nyquist || high <= opts.low_freq)
      throw std::runtime_error("Bad values in options: low-freq and high-freq vs. nyquist");
    const int32 num_fft_bins = padded_ / 2;
    const BaseFloat bin_width = opts.samp_freq / padded_;
    const BaseFloat mel_low = MelScale(opts.low_freq), mel_high = MelScale(high);
    const BaseFloat delta = (mel_high - mel_low) / (opts.num_bins + 1);
    mel_off_.assign(1, 0);
    for (int32 b = 0; b < opts.num_bins; b++) {
      const BaseFloat left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
      int32 first = -1, last = -1;
      std::vector<BaseFloat> w(num_fft_bins, 0.0f);
      for (int32 i = 0; i < num_fft_bins; i++) {
        const BaseFloat mel = MelScale(bin_width * i);
        if (mel > left && mel < right) {
          w[i] = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
          if (first < 0) first = i;
          last = i;
        }
      }
      if (first < 0) throw std::runtime_error("You may have set --num-mel-bins too large.");
      mel_first_.push_back(first);
      mel_weights_.insert(mel_weights_.end(), w.begin() + first, w.begin() + last + 1);
      mel_off_.push_back(static_cast<int32>(mel_weights_.size()));
    }
    const int32 N = opts.num_bins;
    dct_.resize(static_cast<size_t>(opts.num_ceps) * N);
    BaseFloat norm = std::sqrt(1.0 / static_cast<BaseFloat>(N));
    for (int32 n = 0; n < N; n++) dct_[n] = norm;
    norm = std::sqrt(2.0 / static_cast<BaseFloat>(N));
    for (int32 k = 1; k < opts.num_ceps; k++)
      for (int32 n = 0; n < N; n++) dct_[static_cast<size_t>(k) * N + n] = norm * std::cos(pi / N * (n + 0.5) * k);
    if (opts.cepstral_lifter != 0.0f) {
      lifter_.resize(opts.num_ceps);
      for (int32 i = 0; i < opts.num_ceps; i++)
        lifter_[i] = 1.0 + 0.5 * opts.cepstral_lifter * std::sin(pi * i / opts.cepstral_lifter);
    }
  }
  int32 Dim() const { return opts_.num_ceps; }
  /// Mfcc::Compute(wave, 1.0, &output): wave = n_samples floats on the device
  void Compute(const BaseFloat *wave_dev, int32 n_samples, CuMatrix<BaseFloat> *output) const {
    // NumFrames feature-functions.cc:29-48
    const int32 rows = opts_.snip_edges ? (n_samples < frame_length_ ? 0 : 1 + (n_samples - frame_length_) / frame_shift_)
                                        : static_cast<int32>(n_samples * 1.0f / frame_shift_ + 0.5f);
    output->Resize(rows, opts_.num_ceps, kUndefined);
    int32 got = 0;
    KhMfccOptions o;
    o.snip_edges = opts_.snip_edges; o.use_energy = opts_.use_energy; o.raw_energy = opts_.raw_energy;
    o.htk_compat = opts_.htk_compat; o.energy_floor = opts_.energy_floor; o.dither = opts_.dither;
    o.dither_seed = opts_.dither_seed;
    KhCheck(kh_mfcc_compute_opts(wave_dev, n_samples, frame_shift_, frame_length_, padded_, opts_.preemph_coeff,
                                 opts_.remove_dc_offset, window_.data(), opts_.num_bins, mel_first_.data(), mel_off_.data(),
                                 mel_weights_.data(), opts_.num_ceps, dct_.data(), lifter_.empty() ? NULL : lifter_.data(), &o,
                                 rows ? output->Data() : NULL, output->Stride(), &got));
    KALDI_HIP_ASSERT(got == rows);
  }

 private:
  static BaseFloat MelScale(BaseFloat freq) { return 1127.0f * logf(1.0f + freq / 700.0f); }
  MfccOptions opts_;
  int32 frame_shift_, frame_length_, padded_;
  std::vector<BaseFloat> window_, mel_weights_, dct_, lifter_;
  std::vector<int32> mel_first_, mel_off_;
};

/// ComputeDeltas(DeltaFeaturesOptions(order, window), input, &output) feature-functions.cc:361-372
inline void ComputeDeltas(int32 order, int32 window, const CuMatrixBase<BaseFloat> &input, CuMatrix<BaseFloat> *output) {
  KALDI_HIP_ASSERT(order >= 0 && order < 1000 && window > 0 && window < 1000);
  std::vector<std::vector<BaseFloat> > sc(order + 1);
  sc[0].assign(1, 1.0f);
  for (int32 i = 1; i <= order; i++) {
    const std::vector<BaseFloat> &prev = sc[i - 1];
    const int32 prev_offset = (static_cast<int32>(prev.size()) - 1) / 2;
    std::vector<BaseFloat> &cur = sc[i];
    cur.assign(prev.size() + 2 * window, 0.0f);
    BaseFloat normalizer = 0.0f;
    for (int32 j = -window; j <= window; j++) {
      normalizer += j * j;
      for (int32 k = -prev_offset; k <= prev_offset; k++) cur[j + k + prev_offset + window] += static_cast<BaseFloat>(j) * prev[k + prev_offset];
    }
    const BaseFloat inv = static_cast<BaseFloat>(1.0 / normalizer);
    for (size_t k = 0; k < cur.size(); k++) cur[k] *= inv;
  }
  std::vector<BaseFloat> flat;
  std::vector<int32> lens;
  for (int32 i = 0; i <= order; i++) { lens.push_back(static_cast<int32>(sc[i].size())); flat.insert(flat.end(), sc[i].begin(), sc[i].end()); }
  output->Resize(input.NumRows(), input.NumCols() * (order + 1), kUndefined);
  KhCheck(kh_compute_deltas(input.Data(), input.Dim(), order, flat.data(), lens.data(), output->Data(), output->Stride()));
}

/// AccCmvnStats(feats, NULL, &stats) transform/cmvn.cc:49-62; stats = 2 x (dim + 1) doubles, row-major
inline void AccCmvnStats(const CuMatrixBase<BaseFloat> &feats, std::vector<double> *stats) {
  if (stats->empty()) stats->assign(2 * (feats.NumCols() + 1), 0.0);
  KALDI_HIP_ASSERT(static_cast<int32>(stats->size()) == 2 * (feats.NumCols() + 1));
  KhCheck(kh_acc_cmvn_stats(feats.Data(), feats.Dim(), stats->data()));
}

/// ApplyCmvn(stats, var_norm, &feats) transform/cmvn.cc:64-113
inline void ApplyCmvn(const std::vector<double> &stats, bool var_norm, CuMatrix<BaseFloat> *feats) {
  const int32 dim = feats->NumCols();
  if (static_cast<int32>(stats.size()) != 2 * (dim + 1)) throw std::runtime_error("Dim mismatch: cmvn stats vs feats");
  const double count = stats[dim];
  if (count < 1.0) throw std::runtime_error("Insufficient stats for cepstral mean and variance normalization");
  std::vector<BaseFloat> offset(dim), scale(dim);
  for (int32 d = 0; d < dim; d++) {
    const double mean = stats[d] / count;
    double sc = 1.0, off = -mean;
    if (var_norm) {
      double var = stats[dim + 1 + d] / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      sc = 1.0 / std::sqrt(var);
      off = -(mean * sc);
    }
    offset[d] = static_cast<BaseFloat>(off);
    scale[d] = static_cast<BaseFloat>(sc);
  }
  if (var_norm) { CuVector<BaseFloat> s(scale); feats->MulColsVec(s); }
  CuVector<BaseFloat> o(offset);
  feats->AddVecToRows(1.0, o);
}

// ---- LatticeFasterDecoder lattice-faster-decoder.h:40-205 ------------------------------
struct LatticeFasterDecoderConfig {
  BaseFloat beam;
  int32 max_active, min_active;
  BaseFloat lattice_beam;
  int32 prune_interval;
  BaseFloat beam_delta, hash_ratio, prune_scale;
  LatticeFasterDecoderConfig()
      : beam(16.0), max_active(std::numeric_limits<int32>::max()), min_active(200), lattice_beam(10.0),
        prune_interval(25), beam_delta(0.5), hash_ratio(2.0), prune_scale(0.1) {}
  KhDecoderConfig ToC() const {
    KhDecoderConfig c = {beam, max_active, min_active, lattice_beam, prune_interval, beam_delta, hash_ratio, prune_scale};
    return c;
  }
};

/// DecodableInterface itf/decodable-itf.h:82-120 (same virtuals).
class DecodableInterface {
 public:
  virtual BaseFloat LogLikelihood(int32 frame, int32 index) = 0;
  virtual bool IsLastFrame(int32 frame) const = 0;
  virtual int32 NumFramesReady() const {
    throw std::runtime_error("NumFramesReady() not implemented for this decodable type.");
  }
  virtual int32 NumIndices() const = 0;
  virtual ~DecodableInterface() {}
};

/// DecodableMatrixScaledMapped (decoder/decodable-matrix.h:33-84) on the device: a matrix of
/// log-likelihoods [frames x pdfs] that is ALREADY scaled (DecodableAmNnet / the GMM scorer
/// apply the acoustic scale when they fill it) + the TransitionIdToPdf map (index 0 unused).
/// This is the decodable the device decoder consumes whole; LogLikelihood(frame, tid) reads
/// one element back for host-side callers.  Neither the matrix nor the map is owned.
class DecodableMatrixMapped : public DecodableInterface {
 public:
  DecodableMatrixMapped(const CuMatrixBase<BaseFloat> &likes, const CuArray<int32> &tid2pdf, const std::vector<int32> &tid2pdf_host)
      : likes_(likes), tid2pdf_(tid2pdf), tid2pdf_host_(tid2pdf_host) {
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(tid2pdf_host.size()) == tid2pdf.Dim());
    for (size_t i = 1; i < tid2pdf_host.size(); i++)
      if (tid2pdf_host[i] < 0 || tid2pdf_host[i] >= likes.NumCols())
        throw std::runtime_error("DecodableMatrixMapped: mismatch, matrix columns vs. pdf-ids of the transition model");
  }
  virtual int32 NumFramesReady() const { return likes_.NumRows(); }
  virtual bool IsLastFrame(int32 frame) const {
    KALDI_HIP_ASSERT(frame < NumFramesReady());
    return frame == NumFramesReady() - 1;
  }
  virtual BaseFloat LogLikelihood(int32 frame, int32 tid) {
    KALDI_HIP_ASSERT(frame >= 0 && frame < likes_.NumRows() && tid > 0 && tid < static_cast<int32>(tid2pdf_host_.size()));
    BaseFloat v;
    KhCheck(kh_memcpy_2d(&v, 4, likes_.Data() + static_cast<size_t>(frame) * likes_.Stride() + tid2pdf_host_[tid], 4, 4, 1, 1));
    return v;
  }
  virtual int32 NumIndices() const { return static_cast<int32>(tid2pdf_host_.size()) - 1; }
  const CuMatrixBase<BaseFloat> &Likes() const { return likes_; }
  const int32 *TransitionIdToPdfDevice() const { return tid2pdf_.Data(); }

 private:
  const CuMatrixBase<BaseFloat> &likes_;
  const CuArray<int32> &tid2pdf_;
  const std::vector<int32> &tid2pdf_host_;
};

/// What DecodeUtteranceLatticeFaster (decoder-wrappers.cc:197-293) takes from the decoder.
struct RawLattice {
  std::vector<int32> state_frame, state_hclg, arc_src, arc_dst, arc_ilabel, arc_olabel;
  std::vector<BaseFloat> state_final, arc_graph, arc_acoustic;
};

/// CompactLattice as arrays: arcs sorted by source state, label = word (acceptor), weight =
/// (graph, acoustic) + the transition-id string [arc_string_offsets[a], arc_string_offsets[a + 1]).
struct CompactLatticeArrays {
  int32 num_states;
  std::vector<int32> arc_src, arc_dst, arc_label, arc_string_offsets, arc_strings, final_string_offsets, final_strings;
  std::vector<BaseFloat> arc_graph, arc_acoustic, final_graph, final_acoustic;
};

/// DeterminizeLatticePhonePrunedWrapper as DecodeUtteranceLatticeFaster calls it
/// (decoder-wrappers.cc:264-274; lat/determinize-lattice-pruned.cc:1497-1519, word-level pass).
/// Returns false when the determinization stopped at max_mem ("finished earlier than the beam").
inline bool DeterminizeLatticePruned(const RawLattice &raw, double beam, CompactLatticeArrays *clat,
                                     BaseFloat delta = 0.0009765625f, int32 max_mem = 50000000) {
  KhCompactLattice *c = kh_determinize_lattice_pruned(
      static_cast<int>(raw.state_frame.size()), static_cast<int>(raw.arc_src.size()), raw.arc_src.data(), raw.arc_dst.data(),
      raw.arc_ilabel.data(), raw.arc_olabel.data(), raw.arc_graph.data(), raw.arc_acoustic.data(), raw.state_final.data(),
      beam, delta, max_mem);
  if (c == NULL) throw std::runtime_error(std::string("ERROR (libkaldi_hip) ") + kh_last_error());
  int32 ns = 0, na = 0, nal = 0, nfl = 0, complete = 0;
  KhCheck(kh_compact_lattice_sizes(c, &ns, &na, &nal, &nfl, &complete));
  clat->num_states = ns;
  clat->arc_src.resize(na); clat->arc_dst.resize(na); clat->arc_label.resize(na);
  clat->arc_graph.resize(na); clat->arc_acoustic.resize(na);
  clat->arc_string_offsets.resize(na + 1); clat->arc_strings.resize(nal);
  clat->final_graph.resize(ns); clat->final_acoustic.resize(ns);
  clat->final_string_offsets.resize(ns + 1); clat->final_strings.resize(nfl);
  KhCheck(kh_compact_lattice_get(c, clat->arc_src.data(), clat->arc_dst.data(), clat->arc_label.data(), clat->arc_graph.data(),
                                 clat->arc_acoustic.data(), clat->arc_string_offsets.data(), clat->arc_strings.data(),
                                 clat->final_graph.data(), clat->final_acoustic.data(), clat->final_string_offsets.data(),
                                 clat->final_strings.data()));
  kh_compact_lattice_free(c);
  return complete != 0;
}

/// OnlineIvectorExtractionInfo + OnlineIvectorFeature (online2/online-ivector-feature.h:51-134, :226-330),
/// deterministic mode, for a batch of utterances; the model arrays are the host copies of the
/// objects Init() reads (lda_mat_, global_cmvn_stats_, diag_ubm_, extractor_).
class OnlineIvectorExtractor {
 public:
  OnlineIvectorExtractor(const KhIvectorConfig &cfg, const std::vector<BaseFloat> &lda_mat,
                         const std::vector<double> &global_cmvn_stats, const std::vector<BaseFloat> &ubm_gconsts,
                         const std::vector<BaseFloat> &ubm_means_invvars, const std::vector<BaseFloat> &ubm_inv_vars,
                         const std::vector<double> &M, const std::vector<double> &Sigma_inv)
      : cfg_(cfg) {
    h_ = kh_ivector_extractor_create(&cfg, lda_mat.data(), global_cmvn_stats.data(), ubm_gconsts.data(), ubm_means_invvars.data(),
                                     ubm_inv_vars.data(), M.data(), Sigma_inv.data());
    if (h_ == NULL) throw std::runtime_error(std::string("ERROR (libkaldi_hip) ") + kh_last_error());
  }
  ~OnlineIvectorExtractor() { kh_ivector_extractor_destroy(h_); }
  int32 IvectorDim() const { return cfg_.ivector_dim; }
  /// GetFrame for every frame: feats = device [sum T x base_dim]; ivectors resized to [sum T x ivector_dim]
  void Extract(const CuMatrixBase<BaseFloat> &feats, const std::vector<int32> &utt_row_offsets, CuMatrix<BaseFloat> *ivectors) const {
    KALDI_HIP_ASSERT(!utt_row_offsets.empty() && utt_row_offsets.back() == feats.NumRows());
    ivectors->Resize(feats.NumRows(), cfg_.ivector_dim, kUndefined);
    KhCheck(kh_ivector_extract(h_, feats.Data(), feats.Stride(), utt_row_offsets.data(),
                               static_cast<int>(utt_row_offsets.size()) - 1, ivectors->Data(), ivectors->Stride()));
  }
  /// One OnlineIvectorExtractorAdaptationState per utterance, row-concatenated [n x StateDim()] doubles
  /// (layout: include/kaldi_hip.h kh_ivector_extract_adapt).  FreshStates = the default-constructed state.
  int32 StateDim() const { return kh_ivector_state_dim(h_); }
  std::vector<double> FreshStates(int32 n) const {
    const int32 sd = StateDim(), lo = 2 * (cfg_.base_dim + 1) + 2;
    std::vector<double> st(static_cast<size_t>(n) * sd, 0.0);
    for (int32 u = 0; u < n; u++) {
      st[static_cast<size_t>(u) * sd + lo - 1] = 1.0;                 // the prior's quadratic term: the unit matrix
      st[static_cast<size_t>(u) * sd + lo] = cfg_.prior_offset;       // its linear term: prior_offset * e_0
    }
    return st;
  }
  /// SetAdaptationState(states[u]) -> every GetFrame -> GetAdaptationState into states[u] (before LimitFrames)
  void Extract(const CuMatrixBase<BaseFloat> &feats, const std::vector<int32> &utt_row_offsets, std::vector<double> *states,
               CuMatrix<BaseFloat> *ivectors) const {
    const int32 n = static_cast<int32>(utt_row_offsets.size()) - 1;
    KALDI_HIP_ASSERT(n >= 0 && utt_row_offsets.back() == feats.NumRows() && states->size() == static_cast<size_t>(n) * StateDim());
    ivectors->Resize(feats.NumRows(), cfg_.ivector_dim, kUndefined);
    std::vector<double> out(states->size());
    KhCheck(kh_ivector_extract_adapt(h_, feats.Data(), feats.Stride(), utt_row_offsets.data(), n, states->data(), out.data(),
                                     ivectors->Data(), ivectors->Stride()));
    states->swap(out);
  }
  /// OnlineIvectorExtractorAdaptationState::LimitFrames (online-ivector-feature.cc:99-117) on one state
  void LimitFrames(double *st, BaseFloat max_remembered_frames) const {
    const int32 B = cfg_.base_dim, nc = 2 * (B + 1), lo = nc + 2, sd = StateDim();
    const BaseFloat count = static_cast<BaseFloat>(st[B]);
    if (count > max_remembered_frames) {
      const double f = max_remembered_frames / count;
      for (int32 i = 0; i < nc; i++) st[i] *= f;
    }
    const double lim = static_cast<BaseFloat>(max_remembered_frames * cfg_.posterior_scale), n = st[lo - 2];
    if (n > lim) {                                           // OnlineIvectorEstimationStats::Scale (ivector-extractor.cc:570-592)
      const double scale = lim / n, mc = cfg_.max_count;
      st[lo - 2] = n * scale;
      for (int32 i = lo; i < sd; i++) st[i] *= scale;        // the linear term and the per-Gaussian counts
      const double add = mc == 0.0 ? 1.0 - scale : std::max(n * scale, mc) / mc - scale * std::max(n, mc) / mc;
      st[lo - 1] = st[lo - 1] * scale + add;
      st[lo] += cfg_.prior_offset * add;
    }
  }

 private:
  OnlineIvectorExtractor(const OnlineIvectorExtractor &);
  OnlineIvectorExtractor &operator=(const OnlineIvectorExtractor &);
  KhIvectorConfig cfg_;
  KhIvectorExtractor *h_;
};

class LatticeFasterDecoder {
 public:
  /// fst: HCLG as host CSR (the arrays ReadFstKaldi would yield); not owned.  A decoder for batches of up
  /// to max_batch utterances of up to max_frames frames each.
  LatticeFasterDecoder(KhFst *fst, const LatticeFasterDecoderConfig &config, int max_batch, int max_frames)
      : dec_(NULL), fst_(fst), max_batch_(max_batch), max_frames_(max_frames) {
    KhDecoderConfig c = config.ToC();
    dec_ = kh_decoder_create(fst, &c, max_batch, max_frames);
    if (!dec_) KhCheck(KH_EINVAL);
  }
  /// LatticeFasterDecoder(fst, config) lattice-faster-decoder.h:101-102: the reference's constructor.  One
  /// utterance at a time, any length (the arenas are sized when Decode() sees the decodable); the accessors
  /// without an utterance index below go with it.
  LatticeFasterDecoder(const KhFst &fst, const LatticeFasterDecoderConfig &config)
      : dec_(NULL), fst_(&fst), max_batch_(1), max_frames_(std::numeric_limits<int32>::max()) {
    KhDecoderConfig c = config.ToC();
    dec_ = kh_decoder_create(fst_, &c, max_batch_, max_frames_);
    if (!dec_) KhCheck(KH_EINVAL);
  }
  ~LatticeFasterDecoder() { kh_decoder_destroy(dec_); }
  /// The caller's turn inside Decode(): fn(arg) runs on the calling thread right after the decode kernel's launch (the next
  /// batch's forward pass, enqueued behind it); nullptr = none.  (No counterpart in the reference: its loop is sequential.)
  void SetAfterLaunch(void (*fn)(void *), void *arg) { KhCheck(kh_decoder_set_after_launch(dec_, fn, arg)); }
  /// SetOptions lattice-faster-decoder.h:104-106: takes effect at the next Decode() (the device decoder is
  /// rebuilt: its per-frame capacities derive from max_active)
  void SetOptions(const LatticeFasterDecoderConfig &config) {
    KhDecoderConfig c = config.ToC();
    KhDecoder *d = kh_decoder_create(fst_, &c, max_batch_, max_frames_);
    if (!d) KhCheck(KH_EINVAL);
    kh_decoder_destroy(dec_);
    dec_ = d;
  }
  /// the single-utterance accessors of the reference (lattice-faster-decoder.h:116-140)
  bool ReachedFinal() const { return ReachedFinal(0); }
  BaseFloat FinalRelativeCost() const { return FinalRelativeCost(0); }
  bool GetRawLattice(RawLattice *lat) const { return GetRawLattice(0, lat); }
  bool GetBestPath(std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost, BaseFloat *acoustic_cost) const {
    return GetBestPath(0, alignment, words, graph_cost, acoustic_cost);
  }
  int32 NumFramesDecoded() const {
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, 0, &st));
    return st.num_frames;
  }
  /// Decode(&decodable) for a batch; loglikes = device matrix of scaled log-likelihoods
  /// (rows utt_row_offsets[u]..), tid2pdf = device LUT (TransitionIdToPdf) or NULL.
  bool Decode(const BaseFloat *loglikes, int32 stride, const std::vector<int32> &utt_row_offsets,
              const int32 *tid2pdf) {
    KhCheck(kh_decoder_decode(dec_, loglikes, stride, utt_row_offsets.data(),
                              static_cast<int>(utt_row_offsets.size()) - 1, tid2pdf));
    return true;
  }
  /// Decode(DecodableInterface *decodable) lattice-faster-decoder.h:111-114 for ONE utterance.  The
  /// device decoder consumes the whole matrix, so the decodable must be matrix-backed
  /// (DecodableMatrixMapped); the decodable is not owned.  Returns true if any tokens survived.
  bool Decode(DecodableInterface *decodable) {
    DecodableMatrixMapped *m = dynamic_cast<DecodableMatrixMapped *>(decodable);
    if (m == NULL) throw std::runtime_error("LatticeFasterDecoder::Decode: the device decoder needs a matrix-backed decodable");
    std::vector<int32> off(2, 0);
    off[1] = m->NumFramesReady();
    Decode(m->Likes().Data(), m->Likes().Stride(), off, m->TransitionIdToPdfDevice());
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, 0, &st));
    return st.num_tokens > 0;
  }
  /// Raw lattices + best paths of the whole batch on host threads (0 = all cores);
  /// optional, the per-utterance getters compute on demand otherwise.
  void Prepare(int num_threads = 0) { KhCheck(kh_decoder_prepare(dec_, num_threads)); }
  bool ReachedFinal(int utt) const {
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, utt, &st));
    return st.reached_final != 0;
  }
  BaseFloat FinalRelativeCost(int utt) const {
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, utt, &st));
    return st.final_relative_cost;
  }
  bool GetRawLattice(int utt, RawLattice *lat) const {
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, utt, &st));
    const size_t n = st.num_tokens, m = st.num_links;
    lat->state_frame.resize(n); lat->state_hclg.resize(n); lat->state_final.resize(n);
    lat->arc_src.resize(m); lat->arc_dst.resize(m); lat->arc_ilabel.resize(m); lat->arc_olabel.resize(m);
    lat->arc_graph.resize(m); lat->arc_acoustic.resize(m);
    KhCheck(kh_decoder_get_raw_lattice(dec_, utt, lat->state_frame.data(), lat->state_hclg.data(),
                                       lat->state_final.data(), lat->arc_src.data(), lat->arc_dst.data(),
                                       lat->arc_ilabel.data(), lat->arc_olabel.data(), lat->arc_graph.data(),
                                       lat->arc_acoustic.data()));
    return n > 0;
  }
  /// GetBestPath + GetLinearSymbolSequence
  bool GetBestPath(int utt, std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost,
                   BaseFloat *acoustic_cost) const {
    KhDecodeStats st;
    KhCheck(kh_decoder_get_stats(dec_, utt, &st));
    alignment->resize(st.num_frames + 16);
    words->resize(4 * st.num_frames + 64);
    int32 na = 0, nw = 0;
    KhCheck(kh_decoder_get_best_path(dec_, utt, alignment->data(), static_cast<int>(alignment->size()), &na,
                                     words->data(), static_cast<int>(words->size()), &nw, graph_cost, acoustic_cost));
    alignment->resize(na);
    words->resize(nw);
    return true;
  }
  /// GetBestPath of utterances [first, first + n) in one library call: alignments / words row-concatenated,
  /// *_offsets = n + 1 offsets into them, one cost pair per utterance
  void GetBestPaths(int first, int n, std::vector<int32> *alignments, std::vector<int64_t> *alignment_offsets,
                    std::vector<int32> *words, std::vector<int64_t> *word_offsets, std::vector<BaseFloat> *graph_costs,
                    std::vector<BaseFloat> *acoustic_costs) const {
    int64_t frames = 0;
    for (int u = first; u < first + n; u++) {
      KhDecodeStats st;
      KhCheck(kh_decoder_get_counters(dec_, u, &st));
      frames += st.num_frames + 16;
    }
    alignments->resize(static_cast<size_t>(frames) + 16);
    words->resize(4 * static_cast<size_t>(frames) + 64);
    alignment_offsets->resize(n + 1);
    word_offsets->resize(n + 1);
    graph_costs->resize(n);
    acoustic_costs->resize(n);
    float dummy = 0.f;   // (n == 0: valid pointers for the argument check)
    KhCheck(kh_decoder_get_best_paths(dec_, first, n, alignments->data(), static_cast<int64_t>(alignments->size()),
                                      alignment_offsets->data(), words->data(), static_cast<int64_t>(words->size()),
                                      word_offsets->data(), n ? graph_costs->data() : &dummy, n ? acoustic_costs->data() : &dummy));
    alignments->resize(static_cast<size_t>(alignment_offsets->back()));
    words->resize(static_cast<size_t>(word_offsets->back()));
  }

 private:
  LatticeFasterDecoder(const LatticeFasterDecoder &);
  LatticeFasterDecoder &operator=(const LatticeFasterDecoder &);
  KhDecoder *dec_;
  const KhFst *fst_;
  int max_batch_, max_frames_;
};

/// decoder/lattice-faster-online-decoder.h:44-200 for num_streams concurrent
/// utterances (stream = index).  AdvanceDecoding takes, per stream, the device matrix
/// of the next frames' scaled log-likelihoods (what DecodableNnet2Online serves).
class LatticeFasterOnlineDecoder {
 public:
  LatticeFasterOnlineDecoder(KhFst *fst, const LatticeFasterDecoderConfig &config, int num_streams, int max_frames)
      : dec_(NULL) {
    KhDecoderConfig c = config.ToC();
    dec_ = kh_online_decoder_create(fst, &c, num_streams, max_frames);
    if (!dec_) KhCheck(KH_EINVAL);
  }
  ~LatticeFasterOnlineDecoder() { kh_online_decoder_destroy(dec_); }
  /// The reference's own iteration order (kh_online_decoder_set_reference_order); between utterances only.
  void SetReferenceOrder(bool enable) { KhCheck(kh_online_decoder_set_reference_order(dec_, enable ? 1 : 0)); }
  void InitDecoding(const std::vector<int32> &streams) {
    KhCheck(kh_online_decoder_init_decoding(dec_, streams.data(), static_cast<int>(streams.size())));
  }
  /// loglikes[i]: device pointer to num_frames[i] rows of `stride` floats for streams[i].
  void AdvanceDecoding(const std::vector<int32> &streams, const std::vector<const BaseFloat *> &loglikes, int32 stride,
                       const std::vector<int32> &num_frames, const int32 *tid2pdf) {
    KhCheck(kh_online_decoder_advance(dec_, streams.data(), static_cast<int>(streams.size()), loglikes.data(), stride,
                                      num_frames.data(), tid2pdf));
  }
  void FinalizeDecoding(const std::vector<int32> &streams) {
    KhCheck(kh_online_decoder_finalize(dec_, streams.data(), static_cast<int>(streams.size())));
  }
  int32 NumFramesDecoded(int stream) const {
    int32 n = 0;
    KhCheck(kh_online_decoder_num_frames_decoded(dec_, stream, &n));
    return n;
  }
  bool GetRawLattice(int stream, RawLattice *lat, bool use_final_probs = true) const {
    KhDecodeStats st;
    KhCheck(kh_online_decoder_get_stats(dec_, stream, use_final_probs, &st));
    const size_t n = st.num_tokens, m = st.num_links;
    lat->state_frame.resize(n); lat->state_hclg.resize(n); lat->state_final.resize(n);
    lat->arc_src.resize(m); lat->arc_dst.resize(m); lat->arc_ilabel.resize(m); lat->arc_olabel.resize(m);
    lat->arc_graph.resize(m); lat->arc_acoustic.resize(m);
    KhCheck(kh_online_decoder_get_raw_lattice(dec_, stream, use_final_probs, lat->state_frame.data(),
                                              lat->state_hclg.data(), lat->state_final.data(), lat->arc_src.data(),
                                              lat->arc_dst.data(), lat->arc_ilabel.data(), lat->arc_olabel.data(),
                                              lat->arc_graph.data(), lat->arc_acoustic.data()));
    return n > 0;
  }
  bool GetBestPath(int stream, std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost,
                   BaseFloat *acoustic_cost, bool use_final_probs = true) const {
    const int32 T = NumFramesDecoded(stream);
    alignment->resize(T + 16);
    words->resize(4 * T + 64);
    int32 na = 0, nw = 0;
    KhCheck(kh_online_decoder_get_best_path(dec_, stream, use_final_probs, alignment->data(),
                                            static_cast<int>(alignment->size()), &na, words->data(),
                                            static_cast<int>(words->size()), &nw, graph_cost, acoustic_cost));
    alignment->resize(na);
    words->resize(nw);
    return true;
  }

 private:
  LatticeFasterOnlineDecoder(const LatticeFasterOnlineDecoder &);
  LatticeFasterOnlineDecoder &operator=(const LatticeFasterOnlineDecoder &);
  KhOnlineDecoder *dec_;
};

}  // namespace kaldi
#endif  // KALDI_HIP_HOST_H_
