// kaldi-hip.h — C++ host layer above the C-ABI (include/kaldi_hip.h).
//
// Mirrors the reference's class interface for the hot path — same names, argument
// meaning and error behaviour — so that code written against cudamatrix/cu-matrix.h,
// cu-vector.h, cu-array.h, cu-device.h, cu-math.h, decoder/lattice-faster-decoder.h
// reads the same here:
//   kaldi::CuDevice            cudamatrix/cu-device.h:41-143
//   kaldi::CuMatrix<float>     cudamatrix/cu-matrix.h:62-644  (forward-path subset)
//   kaldi::CuVector<float>     cudamatrix/cu-vector.h
//   kaldi::CuArray<T>          cudamatrix/cu-array.h:36-105
//   kaldi::cu::Splice          cudamatrix/cu-math.h
//   kaldi::LatticeFasterDecoderConfig / LatticeFasterDecoder
//                              decoder/lattice-faster-decoder.h:40-205
// Errors throw std::runtime_error exactly as KALDI_ERR does
// (base/kaldi-error.cc:143,179-182); all operations are synchronous at the API
// (results visible on return), like the reference's CU_SAFE_CALL
// (cudamatrix/cu-common.h:37-44).  Header-only; link with libkaldi_hip.so.
#ifndef KALDI_HIP_HOST_H_
#define KALDI_HIP_HOST_H_

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/kaldi_hip.h"

namespace kaldi {

typedef float BaseFloat;
typedef int32_t int32;
typedef int32_t MatrixIndexT;
enum MatrixTransposeType { kTrans = 112, kNoTrans = 111 };  // matrix/matrix-common.h:32-35
enum MatrixResizeType { kSetZero, kUndefined, kCopyData };

inline void KhCheck(int rc) {
  if (rc != KH_OK) throw std::runtime_error(std::string("ERROR (libkaldi_hip) ") + kh_last_error());
}
#define KALDI_HIP_ASSERT(cond)                                                        \
  do {                                                                                \
    if (!(cond)) throw std::runtime_error(std::string("KALDI_ASSERT: failed: ") + #cond); \
  } while (0)

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
  void SelectGpuId(int32 ordinal) { KhCheck(kh_select_gpu(ordinal)); }
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

 private:
  CuDevice() {}
};

// ---- CuArray cu-array.h:36-105 -------------------------------------------------------
template <typename T>
class CuArray {
 public:
  CuArray() : dim_(0), data_(NULL) {}
  explicit CuArray(const std::vector<T> &src) : dim_(0), data_(NULL) { CopyFromVec(src); }
  ~CuArray() { Destroy(); }
  MatrixIndexT Dim() const { return dim_; }
  const T *Data() const { return data_; }
  void Resize(MatrixIndexT dim) {
    Destroy();
    if (dim > 0) data_ = static_cast<T *>(CuDevice::Instantiate().Malloc(sizeof(T) * dim));
    dim_ = dim;
  }
  void CopyFromVec(const std::vector<T> &src) {
    Resize(static_cast<MatrixIndexT>(src.size()));
    if (dim_) KhCheck(kh_memcpy_2d(data_, sizeof(T) * dim_, src.data(), sizeof(T) * dim_, sizeof(T) * dim_, 1, 0));
  }
  void Destroy() {
    if (data_) kh_free(data_);
    data_ = NULL;
    dim_ = 0;
  }

 private:
  CuArray(const CuArray &);
  CuArray &operator=(const CuArray &);
  MatrixIndexT dim_;
  T *data_;
};

// ---- CuVector cu-vector.h ---------------------------------------------------------------
class CuVector {
 public:
  CuVector() : data_(NULL), dim_(0) {}
  explicit CuVector(MatrixIndexT dim) : data_(NULL), dim_(0) { Resize(dim); }
  explicit CuVector(const std::vector<BaseFloat> &host) : data_(NULL), dim_(0) { CopyFromVec(host); }
  ~CuVector() { if (data_) kh_free(data_); }
  MatrixIndexT Dim() const { return dim_; }
  BaseFloat *Data() { return data_; }
  const BaseFloat *Data() const { return data_; }
  void Resize(MatrixIndexT dim) {
    if (data_) kh_free(data_);
    data_ = NULL;
    dim_ = dim;
    if (dim > 0) {
      data_ = static_cast<BaseFloat *>(CuDevice::Instantiate().Malloc(sizeof(BaseFloat) * dim));
      KhCheck(kh_memset(data_, 0, sizeof(BaseFloat) * dim));
    }
  }
  void CopyFromVec(const std::vector<BaseFloat> &h) {
    Resize(static_cast<MatrixIndexT>(h.size()));
    if (dim_) KhCheck(kh_memcpy_2d(data_, 4 * dim_, h.data(), 4 * dim_, 4 * dim_, 1, 0));
  }
  void CopyToVec(std::vector<BaseFloat> *h) const {
    h->resize(dim_);
    if (dim_) KhCheck(kh_memcpy_2d(h->data(), 4 * dim_, data_, 4 * dim_, 4 * dim_, 1, 1));
  }

 private:
  CuVector(const CuVector &);
  CuVector &operator=(const CuVector &);
  BaseFloat *data_;
  MatrixIndexT dim_;
};

// ---- CuMatrix cu-matrix.h:62-644 (float; members data_, num_cols_, num_rows_,
// stride_ as cu-matrix.h:500-510) ---------------------------------------------------------
class CuMatrix {
 public:
  CuMatrix() : data_(NULL), num_cols_(0), num_rows_(0), stride_(0) {}
  CuMatrix(MatrixIndexT rows, MatrixIndexT cols, MatrixResizeType t = kSetZero)
      : data_(NULL), num_cols_(0), num_rows_(0), stride_(0) { Resize(rows, cols, t); }
  ~CuMatrix() { Destroy(); }
  MatrixIndexT NumRows() const { return num_rows_; }
  MatrixIndexT NumCols() const { return num_cols_; }
  MatrixIndexT Stride() const { return stride_; }
  BaseFloat *Data() { return data_; }
  const BaseFloat *Data() const { return data_; }
  KhMatrixDim Dim() const { KhMatrixDim d = {num_rows_, num_cols_, stride_}; return d; }

  void Resize(MatrixIndexT rows, MatrixIndexT cols, MatrixResizeType t = kSetZero) {  // cu-matrix.cc:47-100
    KALDI_HIP_ASSERT(rows >= 0 && cols >= 0);
    Destroy();
    if (rows == 0 || cols == 0) return;
    size_t pitch;
    data_ = static_cast<BaseFloat *>(CuDevice::Instantiate().MallocPitch(sizeof(BaseFloat) * cols, rows, &pitch));
    num_rows_ = rows;
    num_cols_ = cols;
    stride_ = static_cast<MatrixIndexT>(pitch / sizeof(BaseFloat));
    if (t == kSetZero) KhCheck(kh_memset(data_, 0, pitch * rows));
  }
  void Destroy() {
    if (data_) kh_free(data_);
    data_ = NULL;
    num_rows_ = num_cols_ = stride_ = 0;
  }
  /// CopyFromMat(const MatrixBase&) cu-matrix.cc:283-307: host row-major, stride in elements.
  void CopyFromMat(const BaseFloat *host, MatrixIndexT rows, MatrixIndexT cols, MatrixIndexT host_stride) {
    if (rows != num_rows_ || cols != num_cols_) Resize(rows, cols, kUndefined);
    if (rows) KhCheck(kh_memcpy_2d(data_, 4 * (size_t)stride_, host, 4 * (size_t)host_stride, 4 * (size_t)cols, rows, 0));
  }
  /// CopyToMat cu-matrix.cc:387-412
  void CopyToMat(BaseFloat *host, MatrixIndexT host_stride) const {
    if (num_rows_) KhCheck(kh_memcpy_2d(host, 4 * (size_t)host_stride, data_, 4 * (size_t)stride_, 4 * (size_t)num_cols_, num_rows_, 1));
  }
  void CopyFromMat(const CuMatrix &src) {
    if (src.num_rows_ != num_rows_ || src.num_cols_ != num_cols_) Resize(src.num_rows_, src.num_cols_, kUndefined);
    if (num_rows_) KhCheck(kh_memcpy_2d(data_, 4 * (size_t)stride_, src.data_, 4 * (size_t)src.stride_, 4 * (size_t)num_cols_, num_rows_, 2));
  }
  void Swap(CuMatrix *o) {
    std::swap(data_, o->data_); std::swap(num_cols_, o->num_cols_);
    std::swap(num_rows_, o->num_rows_); std::swap(stride_, o->stride_);
  }

  // ---- forward-path operations; each = the reference method of the same name
  void AddMatMat(BaseFloat alpha, const CuMatrix &A, MatrixTransposeType transA, const CuMatrix &B,
                 MatrixTransposeType transB, BaseFloat beta) {  // cu-matrix.cc:947-982
    KhCheck(kh_add_mat_mat(alpha, A.data_, A.Dim(), transA == kTrans, B.data_, B.Dim(), transB == kTrans, beta, data_, Dim()));
    Sync();
  }
  void ApplySoftMaxPerRow(const CuMatrix &src) {  // :1251-1271
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    KhCheck(kh_softmax_per_row(data_, src.data_, Dim(), src.stride_));
    Sync();
  }
  void ApplyLogSoftMaxPerRow(const CuMatrix &src) {  // :1274-1295
    KALDI_HIP_ASSERT(src.num_rows_ == num_rows_ && src.num_cols_ == num_cols_);
    KhCheck(kh_log_softmax_per_row(data_, src.data_, Dim(), src.stride_));
    Sync();
  }
  void CopyRows(const CuMatrix &src, const std::vector<MatrixIndexT> &indices) {  // :1965-1990
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(indices.size()) == num_rows_ && src.num_cols_ == num_cols_);
    CuArray<MatrixIndexT> idx(indices);  // the reference uploads the index vector per call too (:1976)
    KhCheck(kh_copy_rows(data_, Dim(), src.data_, src.stride_, idx.Data()));
    Sync();
  }
  void GroupPnorm(const CuMatrix &src, BaseFloat power) {  // :1147-1164
    KALDI_HIP_ASSERT(num_cols_ > 0 && src.num_cols_ % num_cols_ == 0 && src.num_rows_ == num_rows_);
    KhCheck(kh_group_pnorm(data_, src.data_, Dim(), src.stride_, src.num_cols_ / num_cols_, power));
    Sync();
  }
  void MulRowsVec(const CuVector &scale) {  // :693-713
    KALDI_HIP_ASSERT(scale.Dim() == num_rows_);
    KhCheck(kh_mul_rows_vec(data_, Dim(), scale.Data()));
    Sync();
  }
  void MulColsVec(const CuVector &scale) {  // :668
    KALDI_HIP_ASSERT(scale.Dim() == num_cols_);
    KhCheck(kh_mul_cols_vec(data_, Dim(), scale.Data()));
    Sync();
  }
  void CopyRowsFromVec(const CuVector &v) {  // :1673-1745
    KALDI_HIP_ASSERT(v.Dim() == num_cols_);
    KhCheck(kh_copy_rows_from_vec(data_, Dim(), v.Data()));
    Sync();
  }
  void AddVecToRows(BaseFloat alpha, const CuVector &row, BaseFloat beta = 1.0) {  // :916-939
    KALDI_HIP_ASSERT(row.Dim() == num_cols_);
    KhCheck(kh_add_vec_to_rows(alpha, row.Data(), beta, data_, Dim()));
    Sync();
  }
  void ApplyFloor(BaseFloat f) { KhCheck(kh_apply_floor(data_, Dim(), f)); Sync(); }   // :1845
  void ApplyLog() { KhCheck(kh_apply_log(data_, Dim())); Sync(); }                     // :600
  void ApplyExp() { KhCheck(kh_apply_exp(data_, Dim())); Sync(); }
  void ApplyPow(BaseFloat p) { KhCheck(kh_apply_pow(data_, Dim(), p)); Sync(); }
  void Scale(BaseFloat a) { KhCheck(kh_scale(data_, Dim(), a)); Sync(); }               // :579
  void SumColumnRanges(const CuMatrix &src, const std::vector<int32> &start_end_pairs) {  // :1994-2028
    KALDI_HIP_ASSERT(static_cast<MatrixIndexT>(start_end_pairs.size()) == 2 * num_cols_ && src.num_rows_ == num_rows_);
    CuArray<int32> r(start_end_pairs);
    KhCheck(kh_sum_column_ranges(data_, Dim(), src.data_, src.Dim(), r.Data()));
    Sync();
  }
  void Lookup(const std::vector<int32> &row_col_pairs, std::vector<BaseFloat> *output) const {  // :2327
    const int n = static_cast<int>(row_col_pairs.size() / 2);
    output->resize(n);
    if (!n) return;
    CuArray<int32> idx(row_col_pairs);
    CuVector out(n);
    KhCheck(kh_matrix_lookup(data_, Dim(), idx.Data(), n, out.Data()));
    out.CopyToVec(output);
  }

 private:
  static void Sync() { KhCheck(kh_synchronize()); }
  CuMatrix(const CuMatrix &);
  CuMatrix &operator=(const CuMatrix &);
  BaseFloat *data_;
  MatrixIndexT num_cols_, num_rows_, stride_;
};

namespace cu {
/// cu::Splice cudamatrix/cu-math.cc:130-165
inline void Splice(const CuMatrix &src, const std::vector<int32> &frame_offsets, CuMatrix *tgt) {
  KALDI_HIP_ASSERT(src.NumCols() * static_cast<int>(frame_offsets.size()) == tgt->NumCols() &&
                   src.NumRows() == tgt->NumRows());
  CuArray<int32> off(frame_offsets);
  KhCheck(kh_splice(tgt->Data(), tgt->Dim(), src.Data(), src.Dim(), off.Data(), off.Dim()));
  KhCheck(kh_synchronize());
}
}  // namespace cu

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

/// What DecodeUtteranceLatticeFaster (decoder-wrappers.cc:197-293) takes from the decoder.
struct RawLattice {
  std::vector<int32> state_frame, state_hclg, arc_src, arc_dst, arc_ilabel, arc_olabel;
  std::vector<BaseFloat> state_final, arc_graph, arc_acoustic;
};

class LatticeFasterDecoder {
 public:
  /// fst: HCLG as host CSR (the arrays ReadFstKaldi would yield); not owned.
  LatticeFasterDecoder(KhFst *fst, const LatticeFasterDecoderConfig &config, int max_batch, int max_frames)
      : dec_(NULL) {
    KhDecoderConfig c = config.ToC();
    dec_ = kh_decoder_create(fst, &c, max_batch, max_frames);
    if (!dec_) KhCheck(KH_EINVAL);
  }
  ~LatticeFasterDecoder() { kh_decoder_destroy(dec_); }
  /// Decode(&decodable) for a batch; loglikes = device matrix of scaled log-likelihoods
  /// (rows utt_row_offsets[u]..), tid2pdf = device LUT (TransitionIdToPdf) or NULL.
  bool Decode(const BaseFloat *loglikes, int32 stride, const std::vector<int32> &utt_row_offsets,
              const int32 *tid2pdf) {
    KhCheck(kh_decoder_decode(dec_, loglikes, stride, utt_row_offsets.data(),
                              static_cast<int>(utt_row_offsets.size()) - 1, tid2pdf));
    return true;
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

 private:
  LatticeFasterDecoder(const LatticeFasterDecoder &);
  LatticeFasterDecoder &operator=(const LatticeFasterDecoder &);
  KhDecoder *dec_;
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
