/* kaldi_hip.h — C-ABI of libkaldi_hip.so: the MI355X (gfx950) implementation of
 * Kaldi's acoustic-scoring + lattice-decoding hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no torch / C++
 * types.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference's src/).  Unless stated otherwise pointers
 * are DEVICE pointers, matrices are row-major float32 described by KhMatrixDim
 * (the reference's MatrixDim, cudamatrix/cu-matrixdim.h:49-53, stride in
 * elements), work is enqueued on the library's current HIP stream
 * (kh_set_stream) and — like the reference, whose CU_SAFE_CALL synchronises
 * after every call (cudamatrix/cu-common.h:37-44) — results are visible to the
 * host after kh_synchronize() or any kh_* call documented as synchronous.
 *
 * Error convention: functions return 0 on success and a negative KH_E* code on
 * failure; kh_last_error() returns the message.  The C++ host layer
 * (old-kaldi-git_amd/host/) turns a failure into std::runtime_error exactly as
 * KALDI_ERR does (base/kaldi-error.cc:143,179-182).  There is NO CPU fallback
 * in this library: if no gfx950 device is usable every compute call fails.
 */
#ifndef KALDI_HIP_H_
#define KALDI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_OK 0
#define KH_EINVAL (-1)  /* dimension / argument violation (KALDI_ASSERT) */
#define KH_EDEVICE (-2) /* HIP runtime failure (CU_SAFE_CALL) */
#define KH_ENOMEM (-3)
#define KH_ESTATE (-4)   /* call sequence violation */
#define KH_ECAPACITY (-5) /* a decoder arena overflowed; see message */
#define KH_ETIMEOUT (-6)  /* a wait for the device ran past its deadline (serving kernel); kh_last_error() holds the streams' state */

/* cudamatrix/cu-matrixdim.h:49-53 */
typedef struct KhMatrixDim {
  int32_t rows;
  int32_t cols;
  int32_t stride;
} KhMatrixDim;

/* ------------------------------------------------------------------ runtime
 * Replaces CuDevice (cudamatrix/cu-device.h:41-143, cu-device.cc). */
const char *kh_last_error(void);
/* cudaGetDeviceCount in SelectGpuId, cu-device.cc:93-192. */
int kh_device_count(void);
/* CuDevice::SelectGpuId with an explicit ordinal (needed for
 * one-process-per-GPU sharding).  ordinal < 0: keep the current device. */
int kh_select_gpu(int ordinal);
/* CuDevice::Enabled() cu-device.h:87 */
int kh_enabled(void);
/* CuDevice::DeviceGetName / GetFreeMemory cu-device.cc:391-397,195-225 */
int kh_device_name(char *buf, size_t len);
int kh_mem_info(size_t *free_bytes, size_t *total_bytes);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all
 * subsequent launches; NULL = the library's own stream. */
int kh_set_stream(void *hip_stream);
void *kh_get_stream(void);
int kh_synchronize(void);
/* CuDevice::Malloc / MallocPitch / Free cu-device.cc:530-555.  Served from a
 * caching pool (the reference allocates on every Resize).  Pitch is a multiple
 * of 256 bytes. */
void *kh_malloc(size_t bytes);
void *kh_malloc_pitch(size_t row_bytes, size_t num_rows, size_t *pitch_bytes);
int kh_free(void *ptr);
int kh_pool_release(void); /* return cached blocks to the driver */
/* cudaMemcpy2D H2D / D2H / D2D in CuMatrix::CopyFromMat / CopyToMat / Swap
 * cu-matrix.cc:112-147,197-231,283-307,387-412.  Synchronous. kind: 0 H2D,
 * 1 D2H, 2 D2D. Pitches and width in bytes. */
int kh_memcpy_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch,
                 size_t width_bytes, size_t height, int kind);
int kh_memset(void *dst, int value, size_t bytes);

/* ------------------------------------------------------------------ a1
 * CuMatrixBase::AddMatMat cu-matrix.cc:947-982 (cublas_gemm
 * cublas-wrappers.h:28-33): C = alpha*op(A)*op(B) + beta*C, FP32 MFMA.
 * transX != 0 means kTrans.  dA/dB describe A/B as stored. */
int kh_add_mat_mat(float alpha, const float *A, KhMatrixDim dA, int transA,
                   const float *B, KhMatrixDim dB, int transB, float beta,
                   float *C, KhMatrixDim dC);
/* Fused affine layer: C = A * W^T + bias (row broadcast).  Equals
 * CopyRowsFromVec + AddMatMat(beta=1) (AffineComponent::Propagate
 * nnet2/nnet-component.cc:1221-1223) and AddMatMat(beta=0) + AddVecToRows
 * (FixedAffineComponent::Propagate :3340-3342). */
int kh_affine(const float *A, KhMatrixDim dA, const float *W, KhMatrixDim dW,
              const float *bias, float *C, KhMatrixDim dC);
/* AffineComponent::Propagate (nnet-component.cc:1219-1224) followed by PnormComponent::Propagate with p = 2
 * (:386-391, CuMatrixBase::GroupPnorm cu-matrix.cc:1037-1057) in one kernel: Y[r][c] = sqrt(sum_j x_j^2), x = row r of
 * A * W^T + bias, j over columns c*group_size .. (c+1)*group_size-1 in order; bit-identical to kh_affine followed by
 * kh_group_pnorm(power = 2).  dW.rows = dY.cols * group_size; kh_affine_pnorm_supported(group_size) says whether
 * the group size is one the kernel tiles (a divisor of 160); KH_EINVAL otherwise. */
int kh_affine_pnorm(const float *A, KhMatrixDim dA, const float *W, KhMatrixDim dW, const float *bias, float *Y,
                    KhMatrixDim dY, int group_size);
int kh_affine_pnorm_supported(int group_size);

/* ------------------------------------------------------------------ a2
 * cudaF_softmax_reduce / CuMatrixBase::ApplySoftMaxPerRow cu-matrix.cc:1251-1271.
 * d describes y; x has stride src_stride. */
int kh_softmax_per_row(float *y, const float *x, KhMatrixDim d, int src_stride);
/* cudaF_log_softmax_reduce / ApplyLogSoftMaxPerRow cu-matrix.cc:1274-1295 */
int kh_log_softmax_per_row(float *y, const float *x, KhMatrixDim d,
                           int src_stride);

/* ------------------------------------------------------------------ a3
 * cudaF_copy_rows / CuMatrixBase::CopyRows cu-matrix.cc:1965-1990:
 * dst[i,:] = idx[i] < 0 ? 0 : src[idx[i],:].  indices is a DEVICE array of
 * dst_dim.rows int32. */
int kh_copy_rows(float *dst, KhMatrixDim dst_dim, const float *src,
                 int src_stride, const int32_t *indices);

/* ------------------------------------------------------------------ a4
 * cudaF_splice / cu::Splice cudamatrix/cu-math.cc:130-165:
 * y[r, k*D + c] = x[clamp(r + off[k], 0, R-1), c].  frame_offsets: DEVICE int32. */
int kh_splice(float *y, KhMatrixDim d_out, const float *x, KhMatrixDim d_in,
              const int32_t *frame_offsets, int n_offsets);

/* ------------------------------------------------------------------ a5
 * cudaF_group_pnorm / CuMatrixBase::GroupPnorm cu-matrix.cc:1147-1164.
 * d describes y (rows x cols); x has cols*group_size columns. */
int kh_group_pnorm(float *y, const float *x, KhMatrixDim d, int src_stride,
                   int group_size, float power);

/* ------------------------------------------------------------------ a6
 * NormalizeComponent::Propagate nnet2/nnet-component.cc:576-588 fused:
 * y = x / sqrt(max(mean(x^2), 2^-66)).  */
int kh_normalize(float *y, const float *x, KhMatrixDim d, int src_stride);
/* cudaF_add_diag_mat_mat via CuVectorBase::AddDiagMat2 cu-vector.cc:517-580
 * (kNoTrans): v[i] = beta*v[i] + alpha * sum_j M[i,j]^2 */
int kh_add_diag_mat2(float alpha, const float *M, KhMatrixDim d, float beta,
                     float *v);
/* cudaF_mul_rows_vec / MulRowsVec cu-matrix.cc:693-713 */
int kh_mul_rows_vec(float *M, KhMatrixDim d, const float *scale);
/* cudaF_mul_cols_vec / MulColsVec cu-matrix.cc:668 */
int kh_mul_cols_vec(float *M, KhMatrixDim d, const float *scale);

/* ------------------------------------------------------------------ a7 */
/* cudaF_copy_rows_from_vec / CopyRowsFromVec cu-matrix.cc:1673-1745 */
int kh_copy_rows_from_vec(float *M, KhMatrixDim d, const float *v);
/* cudaF_add_vec_to_rows / AddVecToRows cu-matrix.cc:916-939 */
int kh_add_vec_to_rows(float alpha, const float *v, float beta, float *M,
                       KhMatrixDim d);
/* cudaF_apply_floor :1845, _apply_log :600, _apply_exp, _apply_pow, _scale :579 */
int kh_apply_floor(float *M, KhMatrixDim d, float floor_val);
int kh_apply_log(float *M, KhMatrixDim d);
int kh_apply_exp(float *M, KhMatrixDim d);
int kh_apply_pow(float *M, KhMatrixDim d, float power);
int kh_scale(float *M, KhMatrixDim d, float alpha);
/* cudaF_sum_column_ranges / SumColumnRanges cu-matrix.cc:1994-2028; ranges:
 * DEVICE array of 2*d.cols int32 ([start,end) per output column). */
int kh_sum_column_ranges(float *y, KhMatrixDim d, const float *x,
                         KhMatrixDim d_src, const int32_t *ranges);
/* cudaF_matrix_lookup / CuMatrixBase::Lookup cu-matrix.cc:2327: out[k] =
 * M[pairs[2k], pairs[2k+1]]; pairs/out DEVICE arrays. */
int kh_matrix_lookup(const float *M, KhMatrixDim d, const int32_t *pairs, int n,
                     float *out);
/* DecodableAmNnet epilogue decodable-am-nnet.h:60-69 fused:
 * M = (log(max(M, 1e-20)) - log_priors[c]) * prob_scale */
int kh_log_prior_scale(float *M, KhMatrixDim d, const float *log_priors,
                       float prob_scale);

/* ------------------------------------------------------------------ a1-a7, <double>
 * The CuMatrix<double> / CuVector<double> instantiation (cu-matrix.cc:2415-2418, cu-vector.cc) of the primitives above:
 * same argument meaning, double elements (strides in elements).  AddMatMat runs on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64).  csrc/kh_double.hip; the cudaD_* launchers of include/cu_kernels_ansi_hip.h forward here. */
int kh_add_mat_mat_d(double alpha, const double *A, KhMatrixDim dA, int transA, const double *B, KhMatrixDim dB,
                     int transB, double beta, double *C, KhMatrixDim dC);
int kh_softmax_per_row_d(double *y, const double *x, KhMatrixDim d, int src_stride);
int kh_log_softmax_per_row_d(double *y, const double *x, KhMatrixDim d, int src_stride);
int kh_copy_rows_d(double *dst, KhMatrixDim dst_dim, const double *src, int src_stride, const int32_t *indices);
int kh_splice_d(double *y, KhMatrixDim d_out, const double *x, KhMatrixDim d_in, const int32_t *frame_offsets,
                int n_offsets);
int kh_group_pnorm_d(double *y, const double *x, KhMatrixDim d, int src_stride, int group_size, double power);
int kh_add_diag_mat2_d(double alpha, const double *M, KhMatrixDim d, double beta, double *v);
int kh_mul_rows_vec_d(double *M, KhMatrixDim d, const double *scale);
int kh_mul_cols_vec_d(double *M, KhMatrixDim d, const double *scale);
int kh_copy_rows_from_vec_d(double *M, KhMatrixDim d, const double *v);
int kh_add_vec_to_rows_d(double alpha, const double *v, double beta, double *M, KhMatrixDim d);
int kh_apply_floor_d(double *M, KhMatrixDim d, double floor_val);
int kh_apply_log_d(double *M, KhMatrixDim d);
int kh_apply_exp_d(double *M, KhMatrixDim d);
int kh_apply_pow_d(double *M, KhMatrixDim d, double power);
int kh_scale_d(double *M, KhMatrixDim d, double alpha);
int kh_sum_column_ranges_d(double *y, KhMatrixDim d, const double *x, KhMatrixDim d_src, const int32_t *ranges);
int kh_matrix_lookup_d(const double *M, KhMatrixDim d, const int32_t *pairs, int n, double *out);

/* ------------------------------------------------------------------ a8
 * nnet2 forward: Nnet + NnetComputer + DecodableAmNnet
 * (nnet2/nnet-nnet.h, nnet2/nnet-compute.cc:63-108,159-166,
 * nnet2/decodable-am-nnet.h:37-98, online-nnet2-decodable.cc:81-142). */
enum KhComponentType {
  KH_SPLICE = 1,
  KH_FIXED_AFFINE = 2,
  KH_AFFINE = 3,
  KH_PNORM = 4,
  KH_NORMALIZE = 5,
  KH_SOFTMAX = 6,
  KH_SUM_GROUP = 7,
  KH_FIXED_SCALE = 8,
  KH_FIXED_BIAS = 9
};
/* HOST-side description of one component; parameters are HOST pointers and are
 * copied to the device by kh_nnet_add_component. */
typedef struct KhComponentDesc {
  int32_t type;
  int32_t input_dim;
  int32_t output_dim;
  const float *linear;    /* [output_dim x input_dim] row-major */
  const float *bias;      /* [output_dim]; scales/bias for FixedScale/FixedBias */
  const int32_t *context; /* splice context */
  int32_t n_context;
  int32_t const_dim;
  float p;
  const int32_t *sizes; /* sum-group sizes */
  int32_t n_sizes;
} KhComponentDesc;

typedef struct KhNnet KhNnet;
KhNnet *kh_nnet_create(void);
void kh_nnet_destroy(KhNnet *nnet);
int kh_nnet_add_component(KhNnet *nnet, const KhComponentDesc *desc);
int kh_nnet_set_priors(KhNnet *nnet, const float *priors_host, int n);
int kh_nnet_num_components(const KhNnet *nnet);
int kh_nnet_input_dim(const KhNnet *nnet);
int kh_nnet_output_dim(const KhNnet *nnet);
int kh_nnet_left_context(const KhNnet *nnet);  /* Nnet::LeftContext nnet-nnet.cc:45 */
int kh_nnet_right_context(const KhNnet *nnet); /* Nnet::RightContext :55 */
/* NnetComputation over a BATCH of utterances stacked by rows.
 * feats: [utt_row_offsets[n_utts] x input_dim]; utterance u owns rows
 * [utt_row_offsets[u], utt_row_offsets[u+1]) (HOST array, n_utts+1).
 * pad_input != 0: each utterance is padded by edge-frame duplication
 * (nnet-compute.cc:75-89) so it yields as many output rows as input rows and
 * out uses the same row offsets; pad_input == 0: utterance u yields
 * T_u - left - right rows, packed in order (out_row_offsets_host, if non-NULL,
 * receives n_utts+1 offsets).
 * epilogue: 0 = raw network output (NnetComputation); 1 = DecodableAmNnet's
 * floor, log, minus log prior, times prob_scale (requires kh_nnet_set_priors). */
int kh_nnet_compute(KhNnet *nnet, const float *feats, int feat_stride,
                    const int32_t *utt_row_offsets_host, int n_utts,
                    int pad_input, int epilogue, float prob_scale, float *out,
                    int out_stride, int32_t *out_row_offsets_host);
/* The same call returning as soon as its work is QUEUED on the library's stream: `out` is valid for work queued on that
 * stream behind it (or after a synchronisation), the handle keeps the call's activation buffers until its next call or its
 * destruction.  What lets a caller prepare the consumer of the output while the forward pass runs
 * (kh_discriminative_lattice_computations_parts; a binary that reads the next archive entry while the GPU works). */
int kh_nnet_compute_async(KhNnet *nnet, const float *feats, int feat_stride,
                    const int32_t *utt_row_offsets_host, int n_utts,
                    int pad_input, int epilogue, float prob_scale, float *out,
                    int out_stride, int32_t *out_row_offsets_host);

/* ------------------------------------------------------------------ a9
 * DiagGmm (gmm/diag-gmm.h:83-135). */
/* DiagGmm::ComputeGconsts gmm/diag-gmm.cc:114-152 — HOST arrays (model-load
 * time, as in the reference).  Returns number of bad gconsts (>= 0). */
int kh_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                           const float *inv_vars, int num_mix, int dim,
                           float *gconsts);
/* DiagGmm::LogLikelihoods(const MatrixBase&, Matrix*) diag-gmm.cc:546-562:
 * loglikes[T x num_mix]. */
int kh_diag_gmm_loglikes(const float *data, KhMatrixDim d_data,
                         const float *gconsts, const float *means_invvars,
                         const float *inv_vars, int num_mix, float *loglikes,
                         int ll_stride);
/* Frame x pdf matrix (gmm-compute-likes.cc:70-77; DecodableAmDiagGmmUnmapped::
 * LogLikelihoodZeroBased decodable-am-diag-gmm.cc:28-71 for every (t,pdf)):
 * out[t,j] = LogSumExp_{m in pdf j}(loglike[t,m], prune), LogSumExp as
 * matrix/kaldi-vector.cc:745-763.  All pdfs' Gaussians concatenated;
 * pdf_offsets: DEVICE int32 [num_pdfs+1]. */
int kh_am_gmm_loglikes(const float *data, KhMatrixDim d_data,
                       const float *gconsts, const float *means_invvars,
                       const float *inv_vars, const int32_t *pdf_offsets,
                       int num_pdfs, int num_mix, float log_sum_exp_prune,
                       float *out, int out_stride);

/* ------------------------------------------------------------------ a10-a14
 * LatticeFasterDecoder (decoder/lattice-faster-decoder.{h,cc}) over a
 * device-resident HCLG, decoding a batch of utterances per call. */
typedef struct KhFst KhFst;
/* fst::Fst<StdArc> as read by ReadFstKaldi (nnet-latgen-faster.cc:108):
 * HOST CSR arrays; arc_offsets has num_states+1 entries; final[s] = +inf for
 * non-final states (TropicalWeight::Zero()).  Limits (NULL + kh_last_error otherwise):
 * num_states + #arcs with ilabel != 0 < 2^28 and #arcs with ilabel == 0 < 2^28 (the
 * device tables are addressed with 32-bit byte offsets, 16 bytes per state / arc);
 * device memory: 20 bytes per state and per emitting arc, 16 per epsilon arc, and
 * 16 more per state and emitting arc in every decoder created on the graph. */
KhFst *kh_fst_create(int32_t num_states, int32_t start,
                     const int64_t *arc_offsets, const int32_t *ilabel,
                     const int32_t *olabel, const float *weight,
                     const int32_t *nextstate, const float *final_cost);
void kh_fst_destroy(KhFst *fst);
int64_t kh_fst_num_arcs(const KhFst *fst);
/* Validates once per (graph, pdf map, model) what the reference asserts on every
 * DecodableAmNnet::LogLikelihood call (nnet2/decodable-am-nnet.h:76-78) /
 * TransitionModel::TransitionIdToPdf (hmm/transition-model.h:312): every ilabel of the
 * graph has an entry in the HOST copy of tid2pdf (NULL: identity minus one) and maps to a
 * column < num_cols of the log-likelihood matrix.  The decode kernels gather unchecked. */
int kh_fst_check_pdf_map(const KhFst *fst, const int32_t *tid2pdf_host, int n_tid2pdf, int num_cols);

/* LatticeFasterDecoderConfig lattice-faster-decoder.h:40-95 (same defaults). */
typedef struct KhDecoderConfig {
  float beam;             /* 16.0 */
  int32_t max_active;     /* INT32_MAX */
  int32_t min_active;     /* 200 */
  float lattice_beam;     /* 10.0 */
  int32_t prune_interval; /* 25 */
  float beam_delta;       /* 0.5 */
  float hash_ratio;       /* 2.0 (sizes the device token table) */
  float prune_scale;      /* 0.1 */
} KhDecoderConfig;
void kh_decoder_config_default(KhDecoderConfig *cfg);

typedef struct KhDecoder KhDecoder;
/* max_batch: utterances decoded concurrently per kh_decoder_decode call
 * (one workgroup each).  max_frames: longest utterance.  max_tokens_per_frame /
 * arena sizes derive from max_active; see DESIGN.md "Decoder memory". */
KhDecoder *kh_decoder_create(const KhFst *fst, const KhDecoderConfig *cfg,
                             int max_batch, int max_frames);
void kh_decoder_destroy(KhDecoder *dec);
/* LatticeFasterDecoder::Decode (lattice-faster-decoder.cc:77-95) for a batch:
 * loglikes is what DecodableAmNnet::LogLikelihood / DecodableMatrixScaledMapped
 * (decoder/decodable-matrix.h:33-84) would return: row t of utterance u at
 * utt_row_offsets[u] + t, column tid2pdf[ilabel] (tid2pdf: DEVICE int32 LUT
 * indexed by transition-id, TransitionModel::TransitionIdToPdf
 * hmm/transition-model.h:312; NULL = identity minus one, ilabel-1).
 * Includes FinalizeDecoding.  Synchronous. */
int kh_decoder_decode(KhDecoder *dec, const float *loglikes, int ll_stride,
                      const int32_t *utt_row_offsets_host, int n_utts,
                      const int32_t *tid2pdf);
/* Per-utterance statistics of the last kh_decoder_decode call. */
typedef struct KhDecodeStats {
  int32_t num_frames;
  int32_t reached_final;      /* ReachedFinal() lattice-faster-decoder.h:143 */
  float final_relative_cost;  /* FinalRelativeCost() */
  float final_best_cost;
  int32_t num_tokens;         /* surviving tokens (lattice states) */
  int32_t num_links;          /* surviving forward links (lattice arcs) */
  int64_t arcs_expanded;      /* emitting+epsilon arcs visited (roofline unit) */
  int64_t tokens_created;
  int32_t status;             /* 0 ok; else the kernel's code: 1-5 a token / link arena or a per-frame cap overflowed, 6 the lattice
                               * did not fit the pool (retried with the exact size), 7 survivor lists full, 8-9 the reference order's
                               * list could not be built, 10 internal inconsistency (a frame's epsilon links are not a DAG: the
                               * backward pruning's fixed point did not stop) - the getters return KH_ECAPACITY with the text */
  int32_t max_tokens_frame;
} KhDecodeStats;
int kh_decoder_get_stats(const KhDecoder *dec, int utt, KhDecodeStats *stats);
/* Same counters without building the lattice (num_tokens / num_links are then the
 * arena slots in use).  Measurement aid, no reference counterpart. */
int kh_decoder_get_counters(const KhDecoder *dec, int utt, KhDecodeStats *stats);
/* How the pruning schedule of the last kh_decoder_decode call treated utterance `utt` (measurement / test aid,
 * no reference counterpart): counters[0] garbage collections (PruneActiveTokens + full compaction when the
 * slot's arenas filled up), [1] frames FinalizeDecoding visited with the frame's extra_costs in LDS, [2] frames
 * it visited through the general routines (more than 12288 tokens), [3] frames whose extra_costs went to the
 * next visit through memory.  All zero under KH_DECODER_PRUNE_SCHEDULE=interval (PruneActiveTokens every
 * prune_interval frames, lattice-faster-decoder.cc:88-89); the lattice is the same either way. */
int kh_decoder_get_schedule_counters(const KhDecoder *dec, int utt, int32_t *counters);
/* enable != 0 (THE DEFAULT since round 6): kh_decoder_decode reproduces the reference's own ITERATION ORDER, so that tokens and
 * forward links are the ones LatticeFasterDecoder itself creates: ProcessEmitting prunes against the running next_cutoff
 * (lattice-faster-decoder.cc:728-733) with the tokens in HashList order (util/hash-list-inl.h:118-147: buckets
 * state % hash_size in order of first occupation, insertion order inside a bucket; hash_size as :37, :219-225), the best
 * token of GetCutoff is the first minimum in that order (:599, :611), and the epsilon closure inserts in the order of its
 * LIFO queue (:766-811).  Each utterance is decoded as by a freshly constructed decoder (hash_size 1000 at its start).
 * enable == 0 (opt-in, "canonical"): the order-independent resolution of those three places (DESIGN.md "Decoder parity":
 * accept against the FINAL next_cutoff, ties to the smallest state id), which is what the reference computes whenever no
 * token lies between the final and the running cutoff - a cheaper kernel, NOT the reference's lattices in general.
 * The environment variable KH_DECODER_ORDER=reference|canonical overrides both. */
int kh_decoder_set_reference_order(KhDecoder *dec, int enable);
/* Search counters of utterance `utt` (-1: summed over the batch) in the last kh_decoder_decode call (measurement aid): counters[0] = emitting
 * candidates that were materialised (given a link slot: the rest of arcs_expanded were read and rejected),
 * counters[1] = 1 if the call ran in reference order. */
int kh_decoder_get_search_counters(const KhDecoder *dec, int utt, int64_t *counters);
/* Duration of the decode kernel of the last kh_decoder_decode call, from HIP
 * events recorded on the launch stream (measurement aid; the reference wraps
 * every CuMatrix op in a Timer, cu-device.cc:384-389). */
int kh_decoder_last_kernel_ms(const KhDecoder *dec, float *ms);
/* Wall time the host threads of the last kh_decoder_decode call (raw lattices, best paths, determinization when
 * kh_decoder_set_determinize is on) still needed after the decode kernel had finished: what the overlap with the
 * kernel did not hide (measurement aid). */
int kh_decoder_last_host_tail_ms(const KhDecoder *dec, float *ms);
/* fn(arg) is called by kh_decoder_decode on the calling thread, once per call, when the LAST decode kernel of the call has
 * finished — i.e. no utterance is waiting to be decoded again after an arena / lattice-pool overflow, and nothing on the
 * device will read the score matrix of this batch any more (offline decoding re-evaluates the acoustic cost of an exported
 * link from it inside the kernel) — and BEFORE the call waits for its completion threads (raw lattices, best paths,
 * determinization): the caller's turn while the host finishes the batch.  What a binary's main loop does there —
 * nnet-latgen-faster.cc:139-160 would compute the NEXT batch's log-likelihoods (kh_nnet_compute on the library's stream),
 * into the same score buffer if it likes.  The hook should return quickly (start the work, do not wait for it): the call's
 * host tail is measured from before it.  Nothing of kh_decoder_decode waits for work the hook enqueues.  (Until round 4
 * the hook fired right after the FIRST launch, which was wrong for a caller that reuses the score buffer when an utterance
 * had to be decoded again.)  fn = NULL: none (default). */
int kh_decoder_set_after_launch(KhDecoder *dec, void (*fn)(void *), void *arg);
/* GetRawLattice (lattice-faster-decoder.cc:109-191), use_final_probs = true,
 * in canonical form: states are the surviving tokens sorted by
 * (frame, hclg_state); arcs sorted by (src, ilabel, olabel, dst, graph, ac).
 * HOST output arrays sized from kh_decoder_get_stats (num_tokens/num_links):
 *   state_frame[num_tokens], state_hclg[num_tokens], state_final[num_tokens]
 *   (LatticeWeight(final,0) value1; +inf = not final),
 *   arc_src/arc_dst/arc_ilabel/arc_olabel[num_links], arc_graph/arc_acoustic
 *   (acoustic_cost - cost_offset[frame], :172). Any pointer may be NULL. */
int kh_decoder_get_raw_lattice(const KhDecoder *dec, int utt,
                               int32_t *state_frame, int32_t *state_hclg,
                               float *state_final, int32_t *arc_src,
                               int32_t *arc_dst, int32_t *arc_ilabel,
                               int32_t *arc_olabel, float *arc_graph,
                               float *arc_acoustic);
/* GetBestPath (lattice-faster-decoder.cc:99-105) + GetLinearSymbolSequence as
 * DecodeUtteranceLatticeFaster uses it (decoder-wrappers.cc:232-246):
 * alignment = ilabels != 0 along the best path, words = olabels != 0.
 * Returns lengths via *n_ali / *n_words (buffers of capacity cap_*), the total
 * weight as (graph, acoustic). */
int kh_decoder_get_best_path(const KhDecoder *dec, int utt, int32_t *alignment,
                             int cap_ali, int32_t *n_ali, int32_t *words,
                             int cap_words, int32_t *n_words,
                             float *graph_cost, float *acoustic_cost);
/* kh_decoder_get_best_path for utterances [first, first + n) in one call (the per-utterance loop of
 * nnet-latgen-faster.cc:163-186 on the library's side of a foreign-function interface): alignments and
 * word sequences row-concatenated, ali_off / words_off = n + 1 offsets into them, one cost pair per
 * utterance.  KH_EINVAL if a buffer is too small. */
int kh_decoder_get_best_paths(const KhDecoder *d, int first, int n, int32_t *alignment, int64_t cap_ali,
                              int64_t *ali_off, int32_t *words, int64_t cap_words, int64_t *words_off,
                              float *graph_cost, float *acoustic_cost);
/* kh_decoder_get_counters / kh_decoder_get_stats for utterances [first, first + n); either array may be NULL. */
int kh_decoder_get_stats_batch(const KhDecoder *d, int first, int n, KhDecodeStats *counters, KhDecodeStats *stats);
/* Host post-pass of the whole batch on num_threads host threads (<= 0: all
 * cores): GetBestPath of every utterance that has none yet, i.e. what
 * DecodeUtteranceLatticeFaster (decoder-wrappers.cc:215-262) does one utterance
 * at a time after Decode().  The per-utterance getters above then return the
 * cached results.  Optional: the getters compute on demand otherwise, and
 * kh_decoder_decode's own completion threads have normally done it already.
 * Since round 4 the best path (and kh_decoder_get_stats' lattice sizes) come
 * straight from the exported token / link arrays; the CANONICAL raw lattice of an
 * utterance is built on first access (kh_decoder_get_raw_lattice, determinization)
 * - the sort of every utterance's lattice was 2 ms of host CPU per utterance that a
 * rank with few host threads does not have. */
int kh_decoder_prepare(KhDecoder *dec, int num_threads);
/* DeterminizeLatticePhonePrunedWrapper behind the decoder, as DecodeUtteranceLatticeFaster runs it when
 * determinize_lattice is set (decoder/decoder-wrappers.cc:264-274; LatticeFasterDecoderConfig::det_opts,
 * lattice-faster-decoder.h:75-91).  With enable != 0 the host threads of kh_decoder_decode that build an utterance's
 * raw lattice as soon as the kernel has exported it also determinize it (beam = the lattice beam; delta, max_mem,
 * phone_determinize, word_determinize, minimize = DeterminizeLatticePhonePrunedOptions, tid_phone as
 * kh_determinize_lattice_phone_pruned takes it - copied), overlapped with the kernel that is still decoding the rest
 * of the batch.
 * kh_decoder_get_compact_lattice returns the utterance's CompactLattice (owned by the decoder, valid until the next
 * kh_decoder_decode / kh_decoder_destroy; read it with kh_compact_lattice_sizes / _get below), NULL on error. */
typedef struct KhCompactLattice KhCompactLattice;
int kh_decoder_set_determinize(KhDecoder *dec, int enable, double beam, float delta, int64_t max_mem, const int32_t *tid_phone,
                               int n_tid, int phone_determinize, int word_determinize, int minimize);
const KhCompactLattice *kh_decoder_get_compact_lattice(KhDecoder *dec, int utt);
/* totals[4] over the batch's CompactLattices: states, arcs, transition-ids on arcs and final weights, lattices whose
 * determinization stopped at max_mem (measurement aid: the size of what the binary would write). */
int kh_decoder_compact_lattice_totals(KhDecoder *dec, int64_t *totals);

/* ------------------------------------------------------------------ (f)1
 * LatticeFasterOnlineDecoder (decoder/lattice-faster-online-decoder.h:44-200) for
 * num_streams concurrent utterances, advanced a chunk of frames at a time (what
 * online2/ feeds from DecodableNnet2Online).  The search, its pruning schedule
 * (every prune_interval frames, :762-764) and therefore the lattice are those of
 * kh_decoder_decode on the concatenated chunks.  A stream is an index in
 * [0, num_streams); every call takes a list of DISTINCT streams and processes them
 * in one kernel launch (one workgroup per stream). */
typedef struct KhOnlineDecoder KhOnlineDecoder;
KhOnlineDecoder *kh_online_decoder_create(const KhFst *hclg, const KhDecoderConfig *config,
                                          int num_streams, int max_frames);
void kh_online_decoder_destroy(KhOnlineDecoder *dec);
/* InitDecoding() :160 (.cc:55-72). */
int kh_online_decoder_init_decoding(KhOnlineDecoder *dec, const int32_t *streams, int n);
/* AdvanceDecoding(decodable, max_num_frames) :166 (.cc:747-769): stream streams[i]
 * decodes the next num_frames[i] frames; loglikes[i] = device matrix of their
 * acoustic_scale * log-likelihoods (num_frames[i] rows of ll_stride floats, whole
 * rows allocated), the rows a Decodable would return for frames NumFramesDecoded()...;
 * tid2pdf as in kh_decoder_decode. */
int kh_online_decoder_advance(KhOnlineDecoder *dec, const int32_t *streams, int n,
                              const float *const *loglikes, int ll_stride,
                              const int32_t *num_frames, const int32_t *tid2pdf);
/* NumFramesDecoded() :194. */
int kh_online_decoder_num_frames_decoded(const KhOnlineDecoder *dec, int stream, int32_t *num_frames);
/* The offline kernel's lazy pruning schedule for the streams (default 0 = the reference's: PruneActiveTokens every
 * prune_interval frames, lattice-faster-online-decoder.cc:811-813): nothing is pruned while a stream advances unless its
 * arenas run low, FinalizeDecoding prunes every frame once.  The final lattice, every best path and the endpointing
 * quantities are unchanged; a raw lattice asked for BEFORE FinalizeDecoding is pruned as of the current frame, not as of
 * the last multiple of prune_interval.  Re-carves the arenas (as much memory as is free, less 48 GB): every stream must be idle.
 * Round 6, experimental: with KH_SERVE_LAZY_SPAN=n (default 0 = only when its arenas run low, the offline kernel's rule) a
 * stream also collects its garbage (PruneActiveTokens + compaction) once n frames have gone unpruned, so that no chunk of a
 * long utterance waits for the collection of a backlog of thousands of frames (chunk latency max 113 -> 28 ms at n = 128);
 * off by default: the serving stress harness saw rare GPU faults with it on (csrc/kh_decoder.hip OnlineLazySpan). */
int kh_online_decoder_set_lazy_prune(KhOnlineDecoder *dec, int enable);
/* kh_decoder_set_reference_order for the streams: LatticeFasterOnlineDecoder::ProcessEmitting (decoder/lattice-faster-online-
 * decoder.cc:864-951) walks the same HashList against the same running next_cutoff as the offline decoder, and with this set
 * the launch-per-job calls (kh_online_decoder_init_decoding / _advance / _finalize) and the persistent serving kernel
 * (kh_online_decoder_serve_*, started afterwards) reproduce it: chunked decoding, the offline kernel in reference order and
 * the line-by-line oracle (mode 0) give the same lattices bit for bit.  Between utterances only (KH_ESTATE while a stream is
 * in a decoding run or the serving kernel is running).  On by default (as kh_decoder_set_reference_order); KH_DECODER_ORDER=
 * reference|canonical in the environment overrides at creation. */
int kh_online_decoder_set_reference_order(KhOnlineDecoder *dec, int enable);
/* The same three calls without a kernel launch per chunk: a PERSISTENT serving kernel, one resident workgroup per stream
 * (num_streams <= 2 x the CU count), which waits on a control block in pinned host memory (online2-wav-nnet2-latgen-faster's
 * per-chunk loop :213-262 for many connections; a stream that is pruning - AdvanceDecoding prunes every prune_interval
 * frames, lattice-faster-online-decoder.cc:811-813 - no longer holds up the others, which a lockstep launch cannot avoid).
 * kh_online_decoder_serve_start: loglikes = the streams' score buffers (DEVICE, stream s owns rows
 * [s * rows_per_stream, (s + 1) * rows_per_stream), row t = frame t of its current utterance; rows_per_stream >= max_frames),
 * tid2pdf as kh_online_decoder_advance; the streams continue from where kh_online_decoder_* calls left them.
 * kh_online_decoder_serve_init / _finalize: InitDecoding / FinalizeDecoding requests (asynchronous; FinalizeDecoding
 * runs after the frames published so far).  kh_online_decoder_serve_publish: frames [0, avail[i]) of stream streams[i]
 * have their scores in the buffer - the kernels that wrote them must have COMPLETED - and the stream decodes up to
 * there.  kh_online_decoder_serve_poll: NumFramesDecoded() so far and whether a request is still in flight (either
 * may be NULL); kh_online_decoder_serve_wait: blocks until the listed streams have caught up (timeout_ms <= 0:
 * KH_SERVE_TIMEOUT_MS, default 30 s);
 * afterwards the getters (kh_online_decoder_get_best_path, _get_raw_lattice, _get_stats) may be used on them while the
 * kernel keeps serving the others.  kh_online_decoder_serve_stop: the kernel leaves (the grid also leaves by itself - as a
 * whole, never one stream's workgroup alone - once EVERY stream has been without work for 2 s, KH_SERVE_IDLE_MS, and is
 * launched again by the next request or poll that finds work: a device-wide synchronisation never waits longer than that);
 * the launch-per-job calls work again.  Results are those of the launch-per-job calls.
 * No call blocks for ever: every wait for the device has a deadline (the timeout argument of _serve_wait;
 * KH_SERVE_TIMEOUT_MS for _serve_stop and the relaunch), after which it returns KH_ETIMEOUT and kh_last_error() holds the
 * control blocks of the streams concerned - frames published / decoded, command and acknowledgement numbers, whether the
 * stream's workgroup is resident and what it was doing (InitDecoding / AdvanceDecoding to which frame / FinalizeDecoding)
 * when it last reported. */
int kh_online_decoder_serve_start(KhOnlineDecoder *dec, const float *loglikes, int ll_stride, int64_t rows_per_stream,
                                  const int32_t *tid2pdf);
int kh_online_decoder_serve_stop(KhOnlineDecoder *dec);
int kh_online_decoder_serve_init(KhOnlineDecoder *dec, const int32_t *streams, int n);
int kh_online_decoder_serve_publish(KhOnlineDecoder *dec, const int32_t *streams, int n, const int32_t *avail);
int kh_online_decoder_serve_finalize(KhOnlineDecoder *dec, const int32_t *streams, int n);
int kh_online_decoder_serve_poll(KhOnlineDecoder *dec, const int32_t *streams, int n, int32_t *decoded, int32_t *in_flight);
int kh_online_decoder_serve_wait(KhOnlineDecoder *dec, const int32_t *streams, int n, int timeout_ms);
/* Serving loops call kh_online_decoder_advance once per chunk with the same transition-id -> pdf map (DEVICE, or NULL) and
 * the same number of score columns: this builds the decoder's arc records for that pair ONCE (a pass over the whole graph
 * and a synchronisation otherwise repeated by every advance call) and validates the map as kh_online_decoder_advance
 * does.  Later advance calls with the same pointer and ll_stride == num_cols reuse the records; the caller must not change
 * the map's contents in between (any other pointer / stride rebuilds as before). */
int kh_online_decoder_set_pdf_map(KhOnlineDecoder *dec, const int32_t *tid2pdf, int num_cols);
/* FinalizeDecoding() :180 (.cc:775-790). */
int kh_online_decoder_finalize(KhOnlineDecoder *dec, const int32_t *streams, int n);
/* GetRawLattice(ofst, use_final_probs) :143 (.cc:143-233), GetBestPath :107 and the
 * counters, valid at any point after the first frame: before FinalizeDecoding the
 * lattice holds every token not yet pruned and, with use_final_probs, the final
 * costs computed on the fly (.cc:160-165).  use_final_probs == 0 after
 * FinalizeDecoding is an error as in the reference (.cc:156-158).  Layout as
 * kh_decoder_get_raw_lattice / kh_decoder_get_best_path (sizes via get_stats). */
int kh_online_decoder_get_stats(KhOnlineDecoder *dec, int stream, int use_final_probs, KhDecodeStats *stats);
int kh_online_decoder_get_raw_lattice(KhOnlineDecoder *dec, int stream, int use_final_probs,
                                      int32_t *state_frame, int32_t *state_hclg, float *state_final,
                                      int32_t *arc_src, int32_t *arc_dst, int32_t *arc_ilabel,
                                      int32_t *arc_olabel, float *arc_graph, float *arc_acoustic);
int kh_online_decoder_get_best_path(KhOnlineDecoder *dec, int stream, int use_final_probs,
                                    int32_t *alignment, int cap_ali, int32_t *n_ali, int32_t *words,
                                    int cap_words, int32_t *n_words, float *graph_cost,
                                    float *acoustic_cost);

/* ------------------------------------------------------------------ (f)3
 * Feature front-end, the step right before the path (feat/, transform/cmvn.cc).
 * kh_mfcc_compute = Mfcc::ComputeInternal (feat/feature-mfcc.cc:119-184) with
 * use_energy = false, dither = 0, snip_edges = true, for a waveform on the DEVICE.  The
 * tables are the ones the reference's constructors build (HOST arrays): window =
 * FeatureWindowFunction (feature-functions.cc:74-92, frame_length floats), mel_first /
 * mel_off / mel_weights = MelBanks::bins_ (mel-computations.cc:33-152: first FFT bin,
 * offsets into the concatenated weights), dct = the first num_ceps rows of
 * ComputeDctMatrix (matrix-functions.cc:592-608, num_ceps x num_bins), lifter =
 * ComputeLifterCoeffs (mel-computations.cc:248-254) or NULL.  padded = the power-of-two
 * FFT size (<= 2048).  *num_frames = NumFrames (feature-functions.cc:29-48); out must
 * hold that many rows of num_ceps floats. */
int kh_mfcc_compute(const float *wave, int n_samples, int frame_shift, int frame_length, int padded,
                    float preemph_coeff, int remove_dc_offset, const float *window_host, int num_bins,
                    const int32_t *mel_first_host, const int32_t *mel_off_host,
                    const float *mel_weights_host, int num_ceps, const float *dct_host,
                    const float *lifter_host, float *out, int out_stride, int *num_frames);
/* The same with the remaining MfccOptions / FrameExtractionOptions (feat/feature-mfcc.h:41-56,
 * feature-functions.h:77-96): snip_edges = 0: NumFrames = round(n / shift), frame r centred on
 * shift * (r + 0.5), the signal extended by reflection (ExtractWindow feature-functions.cc:107-135);
 * use_energy: C0 := log energy of the frame before pre-emphasis and windowing (raw_energy) or of
 * the windowed frame, floored at log(energy_floor) when energy_floor > 0 (feature-mfcc.cc:138-141,
 * :167-171); htk_compat: energy / C0 * sqrt(2) moved to the last column (:173-182); dither > 0:
 * Gaussian noise * dither added to every sample of every window (Dither :51-54) from a
 * counter-based generator seeded with dither_seed (the reference draws from rand(): same
 * distribution, not the same numbers).  kh_mfcc_compute = {snip_edges 1, raw_energy 1, rest 0}. */
typedef struct KhMfccOptions {
  int32_t snip_edges, use_energy, raw_energy, htk_compat;
  float energy_floor, dither;
  uint64_t dither_seed;
} KhMfccOptions;
int kh_mfcc_compute_opts(const float *wave, int n_samples, int frame_shift, int frame_length, int padded,
                         float preemph_coeff, int remove_dc_offset, const float *window_host, int num_bins,
                         const int32_t *mel_first_host, const int32_t *mel_off_host,
                         const float *mel_weights_host, int num_ceps, const float *dct_host,
                         const float *lifter_host, const KhMfccOptions *options, float *out, int out_stride,
                         int *num_frames);
/* ComputeDeltas (feat/feature-functions.cc:361-372): scales of DeltaFeatures
 * (:210-242) for orders 0..order back to back (HOST), their lengths in lens_host. */
int kh_compute_deltas(const float *in, KhMatrixDim d_in, int order, const float *scales_host,
                      const int32_t *lens_host, float *out, int out_stride);
/* AccCmvnStats (transform/cmvn.cc:49-62, no weights): adds the sums of feats (DEVICE) to
 * stats_host [2 x (cols + 1)] doubles (row 0: sum x and the count, row 1: sum x^2).
 * ApplyCmvn (:64-113) is kh_mul_cols_vec + kh_add_vec_to_rows with the offset / scale
 * vectors the caller derives from the stats as the reference does. */
int kh_acc_cmvn_stats(const float *feats, KhMatrixDim d, double *stats_host);

/* ------------------------------------------------------------------ (f)1, serving
 * The per-chunk loop of online2-wav-nnet2-latgen-faster (online2bin/online2-wav-nnet2-latgen-faster.cc:213-262) for many
 * concurrent streams in ONE call per step: the chunk's feature rows enter DecodableNnet2Online
 * (nnet2/online-nnet2-decodable.{h,cc}: NumFramesReady :75-89, ComputeForFrame :91-143 - the context-padded rows of every
 * advancing stream gathered into one matrix, one forward pass, the floor / log / -log prior / acoustic-scale epilogue) and
 * LatticeFasterOnlineDecoder::AdvanceDecoding consumes what became ready.  nnet (with priors set) and dec stay owned by
 * the caller and must outlive the object; results are read from dec (kh_online_decoder_get_*; kh_online_decoder_finalize at
 * the end of an utterance).  kh_online_nnet2_reset: a new utterance on the listed streams (InitDecoding).
 * kh_online_nnet2_step: stream streams[i] receives rows [src_rows[i], src_rows[i] + counts[i]) of the DEVICE matrix src
 * (counts[i] >= 0), finished[i] != 0 = InputFinished() after them; every listed stream then advances by the frames that
 * became ready (at most max_nnet_batch_size per call); frames_decoded (may be NULL) = NumFramesDecoded() afterwards.
 * tid2pdf as kh_online_decoder_advance (pin it with kh_online_decoder_set_pdf_map).  Synchronous. */
typedef struct KhOnlineNnet2 KhOnlineNnet2;
KhOnlineNnet2 *kh_online_nnet2_create(KhNnet *nnet, KhOnlineDecoder *dec, int num_streams, int max_frames, float acoustic_scale,
                                      int pad_input, int max_nnet_batch_size);
void kh_online_nnet2_destroy(KhOnlineNnet2 *h);
int kh_online_nnet2_reset(KhOnlineNnet2 *h, const int32_t *streams, int n);
int kh_online_nnet2_step(KhOnlineNnet2 *h, const int32_t *streams, int n, const float *src, int src_stride,
                         const int32_t *src_rows, const int32_t *counts, const int32_t *finished, const int32_t *tid2pdf,
                         int32_t *frames_decoded);
int kh_online_nnet2_num_frames_ready(const KhOnlineNnet2 *h, int stream, int32_t *ready);
/* Serving through the decoder's persistent kernel (kh_online_decoder_serve_*): after kh_online_nnet2_serve_start,
 * kh_online_nnet2_step returns as soon as the chunk's scores are in the streams' score buffers and published (its
 * frames_decoded = how far the decoder has got, which runs behind), kh_online_nnet2_reset requests InitDecoding,
 * kh_online_nnet2_serve_finalize requests FinalizeDecoding (after the frames submitted so far); _serve_poll / _serve_wait
 * as the decoder's calls.  The score buffers take num_streams x max_frames x output-dim floats of device memory.
 * kh_online_nnet2_serve_stop: back to one launch per step.  Same lattices either way. */
int kh_online_nnet2_serve_start(KhOnlineNnet2 *h, const int32_t *tid2pdf);
int kh_online_nnet2_serve_stop(KhOnlineNnet2 *h);
int kh_online_nnet2_serve_finalize(KhOnlineNnet2 *h, const int32_t *streams, int n);
int kh_online_nnet2_serve_poll(KhOnlineNnet2 *h, const int32_t *streams, int n, int32_t *decoded, int32_t *in_flight);
int kh_online_nnet2_serve_wait(KhOnlineNnet2 *h, const int32_t *streams, int n, int timeout_ms);

/* LatticeStateTimes (lat/lattice-functions.cc:36-67) for a batch of top-sorted lattices
 * (layout as kh_lattice_forward_backward): time of every state (-1: unreachable) and, per
 * lattice, the number of frames (max_times may be NULL).  KH_EINVAL if a lattice is not
 * top-sorted or its state times are inconsistent (the KALDI_ASSERTs of :38-39,:55,:61). */
int kh_lattice_state_times(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                           const int32_t *arc_ilabel, const int32_t *arc_nextstate, const float *state_final,
                           int32_t *state_times, int32_t *max_times);

/* ------------------------------------------------------------------ a15
 * Lattice forward-backward (lat/lattice-functions.cc:36-67,272-354) for a batch
 * of top-sorted lattices given as HOST CSR (state s owns arcs
 * [arc_offsets[s], arc_offsets[s+1])); lattices are concatenated, lattice l
 * owns states [lat_state_offsets[l], lat_state_offsets[l+1]) and its arcs'
 * nextstate values are lattice-local.  Arc weight = (graph, acoustic)
 * LatticeWeight (fstext/lattice-weight.h:47), final = value1+value2 sum
 * (+inf = Zero).  Outputs (HOST): per-arc posterior (float), per-lattice total
 * log-prob (double, tot_backward_prob) and acoustic_like_sum (double), per-state
 * time (LatticeStateTimes).  */
int kh_lattice_forward_backward(int n_lats, const int32_t *lat_state_offsets,
                                const int64_t *arc_offsets,
                                const int32_t *arc_ilabel,
                                const int32_t *arc_nextstate,
                                const float *arc_graph, const float *arc_acoustic,
                                const float *state_final, float *arc_post,
                                double *tot_like, double *acoustic_like_sum,
                                int32_t *state_times);

/* A batch of lattices KEPT ON THE DEVICE: the caller's arrays are uploaded and prepared (validation, LatticeStateTimes,
 * dependency levels, incoming-arc lists) once, and every computation on the batch — the numerator's and the denominator's
 * forward-backward, rescoring with a new score matrix, the forward-backward after it — runs from there
 * (lat/lattice-functions.cc:272-354, :1307-1358; nnet-compute-discriminative.cc:178-321 does exactly this sequence on one
 * lattice).  kh_lattice_batch_create takes the arguments of kh_lattice_forward_backward and returns NULL on an error
 * (kh_last_error).  kh_lattice_batch_forward_backward: outputs as kh_lattice_forward_backward, any may be NULL.
 * kh_lattice_batch_rescore: RescoreLattice with a DEVICE score matrix (lattice l uses rows ll_row_offsets[l]..., HOST
 * offsets; tid2pdf DEVICE or NULL = ilabel - 1); the acoustic costs change on the device, arc_acoustic_out (HOST, may be
 * NULL) receives them.  kh_lattice_last_timings: milliseconds the last lattice call of this thread spent in
 * { upload, device preparation, sweeps, download } (HIP events on the library's stream; measurement aid). */
typedef struct KhLatticeBatch KhLatticeBatch;
KhLatticeBatch *kh_lattice_batch_create(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                                        const int32_t *arc_ilabel, const int32_t *arc_nextstate, const float *arc_graph,
                                        const float *arc_acoustic, const float *state_final);
void kh_lattice_batch_destroy(KhLatticeBatch *batch);
int kh_lattice_batch_sizes(const KhLatticeBatch *batch, int32_t *n_lats, int32_t *total_states, int64_t *total_arcs);
int kh_lattice_batch_forward_backward(KhLatticeBatch *batch, float *arc_post, double *tot_like, double *acoustic_like_sum,
                                      int32_t *state_times);
/* ... with the arc posteriors left on the DEVICE (arc_post_dev: total_arcs floats in the batch's arc order; tot_like /
 * acoustic_like_sum HOST, may be NULL): no 4 bytes per arc over PCIe when the next consumer is a kernel. */
int kh_lattice_batch_forward_backward_dev(KhLatticeBatch *batch, float *arc_post_dev, double *tot_like, double *acoustic_like_sum);
int kh_lattice_batch_rescore(KhLatticeBatch *batch, const float *loglikes, int ll_stride, const int32_t *ll_row_offsets,
                             const int32_t *tid2pdf, float *arc_acoustic_out);
int kh_lattice_last_timings(float *ms4);

/* ComputeLatticeAlphasAndBetas (lat/lattice-functions.cc:412-463) for a batch (same CSR
 * layout as above): alpha / beta per state (log-semiring, or the tropical one when
 * viterbi != 0: LogAddOrMax :395-410), tot[l] = 0.5 * (forward + backward total). */
int kh_lattice_alphas_betas(int n_lats, const int32_t *lat_state_offsets,
                            const int64_t *arc_offsets, const int32_t *arc_ilabel,
                            const int32_t *arc_nextstate, const float *arc_graph,
                            const float *arc_acoustic, const float *state_final,
                            int viterbi, double *alpha, double *beta, double *tot);

/* LatticeForwardBackwardMpeVariants (lat/lattice-functions.cc:740-919): criterion
 * "smbr" (is_mpfe = 0) or "mpfe".  tid2phone / tid2pdf = TransitionIdToPhone /
 * TransitionIdToPdf as arrays of num_tids + 1 entries indexed by transition-id;
 * silence_phones sorted; num_ali = the numerator alignments concatenated
 * (num_ali_offsets[n_lats + 1]; lattice l needs max_time(l) entries, :764).
 * arc_post[a] = posterior_smbr of arc a (0 on epsilon arcs); the Posterior is
 * post[state_times[src(a)]] += (ilabel, arc_post[a]) merged as MergePairVectorSumming
 * (:916-917).  tot_forward_score[l] = the expected frame accuracy (:919); state_times
 * (optional) = LatticeStateTimes of every state.  Fails with
 * KH_ESTATE when one of the reference's forward/backward checks fails (:808, :909). */
int kh_lattice_forward_backward_mpe(int n_lats, const int32_t *lat_state_offsets,
                                    const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                    const int32_t *arc_nextstate, const float *arc_graph,
                                    const float *arc_acoustic, const float *state_final,
                                    const int32_t *tid2phone, const int32_t *tid2pdf, int num_tids,
                                    const int32_t *silence_phones, int n_sil, const int32_t *num_ali,
                                    const int32_t *num_ali_offsets, int is_mpfe, int one_silence_class,
                                    float *arc_post, double *tot_forward_score, int32_t *state_times);

/* RescoreLattice (lat/lattice-functions.cc:1307-1358) with a matrix decodable: every
 * arc with a transition-id gets -loglikes[t][tid2pdf ? tid2pdf[tid] : tid - 1] added to
 * its acoustic cost (host array, updated in place), t = LatticeStateTimes of its source
 * state.  loglikes: DEVICE matrix, lattice l uses rows ll_row_offsets[l]...  Top-sorted
 * input required (the reference sorts first). */
int kh_rescore_lattice(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                       const int32_t *arc_ilabel, const int32_t *arc_nextstate, float *arc_acoustic,
                       const float *loglikes, int ll_stride, const int32_t *ll_row_offsets,
                       const int32_t *tid2pdf);

/* NnetDiscriminativeUpdater::LatticeComputations (nnet2/nnet-compute-discriminative.cc:178-321) for a batch
 * of examples, everything between the network output and the derivative on the device: Lookup + pseudo
 * log-likelihoods log(max(post, 1e-20) / prior) * acoustic_scale written into the denominator lattices
 * (:196-277), LatticeForwardBackwardMmi (criterion 0) or LatticeForwardBackwardMpeVariants ("smbr" 1, "mpfe" 2)
 * (:324-343), the Posterior algebra of hmm/posterior.cc (MergePairVectorSumming, ScalePosterior,
 * ConvertPosteriorToPdfs, AlignmentToPosterior, MergePosteriors with cancellation and drop_frames,
 * ScalePosterior(eg.weight)) and CompObjfAndDeriv (:301-316).
 * Host arrays: the denominator lattices as one top-sorted CSR batch (as kh_lattice_forward_backward), the
 * numerator alignments concatenated (num_ali_offsets[n_lats + 1] = also the row ranges of the examples in the
 * matrices), eg_weights[n_lats], tid2pdf / tid2phone [num_tids + 1] (tid2phone may be NULL for MMI),
 * silence_phones sorted, priors[cols].  DEVICE matrices: posteriors = the network output [sum T x num_pdfs],
 * deriv of the same size (zeroed, then deriv(row, pdf) = w / posteriors(row, pdf) for every merged entry).
 * stats[5] = { tot_num_count, tot_num_objf (MMI), tot_den_objf, CompObjfAndDeriv's tot_objf, tot_weight }.
 * --boost is not supported.  KH_EINVAL where the reference asserts (:194 rows vs frames, :220 lattice length
 * vs alignment length, unsorted lattice); KH_ESTATE when a forward/backward agreement check fails. */
int kh_discriminative_lattice_computations(
    int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets, const int32_t *arc_ilabel,
    const int32_t *arc_nextstate, const float *arc_graph, const float *arc_acoustic, const float *state_final,
    const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights, const int32_t *tid2pdf,
    const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion,
    float acoustic_scale, int drop_frames, int one_silence_class, const float *priors, const float *posteriors,
    KhMatrixDim d_posteriors, float *deriv, KhMatrixDim d_deriv, double *stats);

/* The same call with the denominator lattices as they sit in the examples - one set of host arrays per lattice
 * (NnetDiscriminativeUpdater::Propagate gets its examples one by one, nnet2/nnet-compute-discriminative.cc:150-175):
 * n_states[l], arc_offsets[l][n_states[l] + 1] (starting at 0), and per arc arc_ilabel[l], arc_nextstate[l] (state
 * numbers within the lattice), arc_graph[l], arc_acoustic[l]; state_final[l][n_states[l]].  The library assembles the
 * batch in pinned memory with a few host threads and uploads / prepares it on a stream of its own, beside the forward
 * pass the caller has launched on the library's stream and not waited for; only the steps that read `posteriors` are
 * ordered behind it.  Everything else as kh_discriminative_lattice_computations. */
int kh_discriminative_lattice_computations_parts(
    int n_lats, const int32_t *n_states, const int64_t *const *arc_offsets, const int32_t *const *arc_ilabel,
    const int32_t *const *arc_nextstate, const float *const *arc_graph, const float *const *arc_acoustic,
    const float *const *state_final, const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights,
    const int32_t *tid2pdf, const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion,
    float acoustic_scale, int drop_frames, int one_silence_class, const float *priors, const float *posteriors,
    KhMatrixDim d_posteriors, float *deriv, KhMatrixDim d_deriv, double *stats);

/* ... and in two halves, so that the NEXT batch's forward pass runs beside this batch's lattice work (a trainer's loop:
 * forward(i) queued; begin(i); forward(i + 1) queued; end(i); begin(i + 1); ...).  _begin assembles, uploads and prepares
 * the batch and queues every device step of the call on a stream of the call's own - the steps that read `posteriors`
 * behind an event recorded on the library's stream at the moment of the call, i.e. behind the forward pass queued just
 * before it - and returns without waiting.  _end waits for them, fills stats[5] and destroys the call (also when it
 * returns an error).  posteriors / deriv must stay valid until _end; two calls may be in flight. */
typedef struct KhDiscCall KhDiscCall;
int kh_discriminative_lattice_computations_begin(
    int n_lats, const int32_t *n_states, const int64_t *const *arc_offsets, const int32_t *const *arc_ilabel,
    const int32_t *const *arc_nextstate, const float *const *arc_graph, const float *const *arc_acoustic,
    const float *const *state_final, const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights,
    const int32_t *tid2pdf, const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion,
    float acoustic_scale, int drop_frames, int one_silence_class, const float *priors, const float *posteriors,
    KhMatrixDim d_posteriors, float *deriv, KhMatrixDim d_deriv, KhDiscCall **call);
int kh_discriminative_lattice_computations_end(KhDiscCall *call, double *stats);

/* CuMatrix::CompObjfAndDeriv (cudamatrix/cu-matrix.cc:1198-1248): the supervision
 * labels (row, column, weight) as three host arrays, output / deriv DEVICE matrices of
 * equal size: *tot_objf = sum w log output(r, c), *tot_weight = sum w,
 * deriv(r, c) += w / output(r, c). */
int kh_comp_objf_and_deriv(int n, const int32_t *rows, const int32_t *cols, const float *weights,
                           const float *output, KhMatrixDim d_output, float *deriv, KhMatrixDim d_deriv,
                           float *tot_objf, float *tot_weight);

/* MergePairVectorSumming (util/stl-utils.h:303-322) applied to every frame of a batch at once: the
 * step between the arc posteriors and the Posterior lists of hmm/posterior.cc
 * (lattice-functions.cc:351-352, ConvertPosteriorToPdfs, MergePosteriors).  HOST arrays: entries
 * (row = frame < n_rows, key = transition-id / pdf-id, weight) -> sorted by (row, key), equal keys
 * summed in input order, exact zeros dropped; the outputs hold at most n entries. */
int kh_merge_pair_vector_summing(int64_t n, const int32_t *rows, const int32_t *keys, const float *weights,
                                 int32_t n_rows, int32_t *out_rows, int32_t *out_keys, float *out_weights,
                                 int64_t *n_out);

/* ------------------------------------------------------------------ f2
 * DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1497-1519, called by
 * DecodeUtteranceLatticeFaster, decoder/decoder-wrappers.cc:264-274): the raw state-level
 * lattice (arrays as kh_decoder_get_raw_lattice returns them: state 0 = start, arcs with
 * ilabel = transition-id, olabel = word, weight (graph, acoustic); state_final = +inf for
 * non-final states, else the final graph cost) -> a CompactLattice deterministic on words
 * (acceptor: ilabel = olabel = word; weight = (graph, acoustic) + transition-id string), pruned
 * to `beam` (LatticeFasterDecoderConfig::lattice_beam).  delta / max_mem:
 * DeterminizeLatticePhonePrunedOptions (determinize-lattice-pruned.h:145-162; defaults
 * kDelta = 2^-10, 50000000; max_mem <= 0: unlimited).  Host code (as in the reference).
 * Returns NULL on bad arguments.  `complete` = 0 when the determinization stopped at the
 * memory limit ("Determinization finished earlier than the beam", decoder-wrappers.cc:272). */
typedef struct KhCompactLattice KhCompactLattice;
KhCompactLattice *kh_determinize_lattice_pruned(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                                const int32_t *arc_ilabel, const int32_t *arc_olabel,
                                                const float *arc_graph, const float *arc_acoustic,
                                                const float *state_final, double beam, float delta, int64_t max_mem);
/* The same with every option of DeterminizeLatticePhonePrunedOptions (lat/determinize-lattice-pruned.h:145-175):
 * phone_determinize = the first pass on phone + word labels (:1386-1404), which needs what the reference asks its
 * TransitionModel per transition-id (:1335-1338): tid_phone[tid] = TransitionIdToPhone(tid) when
 * TransitionIdToHmmState(tid) == 0 && !IsSelfLoop(tid), else 0 (n_tid entries, index 0 unused); word_determinize = the
 * pass on words; minimize = PushCompactLatticeStrings + PushCompactLatticeWeights + MinimizeCompactLattice
 * (lat/push-lattice.cc, lat/minimize-lattice.cc).  The reference's defaults are (1, 1, 0);
 * kh_determinize_lattice_pruned is (0, 1, 0). */
KhCompactLattice *kh_determinize_lattice_phone_pruned(int n_states, int n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                                      const int32_t *arc_ilabel, const int32_t *arc_olabel,
                                                      const float *arc_graph, const float *arc_acoustic,
                                                      const float *state_final, const int32_t *tid_phone, int n_tid,
                                                      double beam, float delta, int64_t max_mem, int phone_determinize,
                                                      int word_determinize, int minimize);
int kh_compact_lattice_sizes(const KhCompactLattice *clat, int32_t *n_states, int32_t *n_arcs,
                             int32_t *n_arc_string_labels, int32_t *n_final_string_labels, int32_t *complete);
/* arcs sorted by source state; arc_string_offsets has n_arcs + 1 entries, final_string_offsets
 * n_states + 1; final_graph / final_acoustic = +inf for non-final states.  NULL skips an array. */
int kh_compact_lattice_get(const KhCompactLattice *clat, int32_t *arc_src, int32_t *arc_dst, int32_t *arc_label,
                           float *arc_graph, float *arc_acoustic, int32_t *arc_string_offsets, int32_t *arc_strings,
                           float *final_graph, float *final_acoustic, int32_t *final_string_offsets,
                           int32_t *final_strings);
void kh_compact_lattice_free(KhCompactLattice *clat);

/* ------------------------------------------------------------------ f3
 * OnlineIvectorFeature (online2/online-ivector-feature.{h,cc}) in its deterministic mode: no
 * silence weighting, use_most_recent_ivector = false, a fresh OnlineIvectorExtractorAdaptationState
 * per utterance (no speaker CMVN stats).  The configuration is OnlineIvectorExtractionInfo
 * (online-ivector-feature.h:51-134) with its member models:
 *   lda_mat [feat_dim x lda_cols] (lda_cols = base_dim * (left + right + 1), + 1 for an offset
 *   column), global_cmvn_stats [2 x (base_dim + 1)] double (matrix of OnlineCmvn), diag UBM
 *   (gconsts, means_invvars, inv_vars as kh_diag_gmm_loglikes), IvectorExtractor: M
 *   [num_gauss][feat_dim][ivector_dim], Sigma_inv [num_gauss][feat_dim][feat_dim], prior_offset —
 * all HOST arrays, uploaded once; the derived U_i / Sigma_i^-1 M_i
 * (IvectorExtractor::ComputeDerivedVars, ivector/ivector-extractor.cc:186-217) are computed at
 * creation.  Returns NULL when OnlineIvectorExtractionInfo::Check (online-ivector-feature.cc:70-87)
 * would fail or a limit is exceeded (base_dim <= 64, num_gauss <= 2048, num_gselect <= 16,
 * ivector_dim, feat_dim <= 256). */
typedef struct KhIvectorConfig {
  int32_t base_dim, splice_left, splice_right, feat_dim, num_gauss, ivector_dim, lda_cols;
  int32_t cmn_window, speaker_frames, global_frames, normalize_mean, normalize_variance; /* OnlineCmvnOptions */
  int32_t ivector_period, num_gselect, num_cg_iters;
  float min_post, posterior_scale, max_count;
  double prior_offset;
  /* 1: use_most_recent_ivector + greedy_ivector_extractor (what --online=false sets,
   * online2-wav-nnet2-latgen-faster.cc: one estimate from all the frames of the utterance on every row);
   * 0: use_most_recent_ivector = false (row t: the estimate of frame (t / period) * period) */
  int32_t greedy_most_recent;
} KhIvectorConfig;
typedef struct KhIvectorExtractor KhIvectorExtractor;
KhIvectorExtractor *kh_ivector_extractor_create(const KhIvectorConfig *cfg, const float *lda_mat,
                                                const double *global_cmvn_stats, const float *ubm_gconsts,
                                                const float *ubm_means_invvars, const float *ubm_inv_vars,
                                                const double *M, const double *Sigma_inv);
void kh_ivector_extractor_destroy(KhIvectorExtractor *ext);
/* OnlineIvectorFeature::GetFrame for every frame of a batch of utterances
 * (online-ivector-feature.cc:286-299): feats = DEVICE base features, the utterances row-
 * concatenated, utterance u = rows [utt_row_offsets_host[u], utt_row_offsets_host[u + 1]);
 * ivectors = DEVICE [rows x ivector_dim]: row t holds the iVector estimated from frames
 * 0 .. (t / ivector_period) * ivector_period of its utterance, first dimension minus
 * PriorOffset. */
int kh_ivector_extract(const KhIvectorExtractor *ext, const float *feats, int feat_stride,
                       const int32_t *utt_row_offsets_host, int n_utts, float *ivectors, int ivector_stride);
/* The same with the speaker's OnlineIvectorExtractorAdaptationState (online-ivector-feature.h:138-176:
 * SetAdaptationState before the utterance, GetAdaptationState after it, :151-171): per utterance
 * kh_ivector_state_dim(ext) HOST doubles — CMVN speaker stats [2 x (base_dim + 1)], num_frames, the prior's
 * share of the quadratic diagonal, the linear term [ivector_dim], the per-Gaussian counts [num_gauss] (the
 * quadratic term is diag * I + sum_g count_g U_g).  state_in NULL: fresh speakers; state_out: the state
 * BEFORE LimitFrames (the caller applies it, as GetAdaptationState does, and chains the utterances of a
 * speaker through successive calls). */
int kh_ivector_state_dim(const KhIvectorExtractor *ext);
int kh_ivector_extract_adapt(const KhIvectorExtractor *ext, const float *feats, int feat_stride,
                             const int32_t *utt_row_offsets_host, int n_utts, const double *state_in_host,
                             double *state_out_host, float *ivectors, int ivector_stride);

/* OnlineIvectorFeature WITH FRAME WEIGHTS (online2/online-ivector-feature.h:299-362, .cc:155-254) for n utterances
 * side by side - the feature side of the decoder-traceback silence weighting (--ivector-silence-weighting.*,
 * online2-wav-nnet2-latgen-faster.cc:239-244).  create: the utterances' base features (DEVICE, row-concatenated),
 * the adaptation states they start from (layout of kh_ivector_extract_adapt, NULL = fresh) and the DEVICE matrix
 * [sum T x ivector_dim] whose rows the object fills as their estimation points are reached.  The per-frame inputs
 * that do not depend on the weights (lda_ features, pruned UBM posteriors) are computed here, once.
 *   update_frame_weights(stream, (frame, delta weight)...)  = UpdateFrameWeights() :155-170; num_frames_ready =
 *       NumFramesReady() of the stream at the time of the call (frames >= it are refused as the reference asserts)
 *   get_frames(streams, until_frame)  = what GetFrame(until_frame) triggers: UpdateStatsUntilFrameWeighted()
 *       :215-254 when weights were supplied, UpdateStatsUntilFrame() :191-213 otherwise, for all listed streams in
 *       one launch; rows [0, until_frame] of the streams' iVector matrix are valid afterwards
 *   get_stats(states_out [n x state_dim])  = the OnlineIvectorEstimationStats part of GetAdaptationState() :283-293
 *       (num_frames, quadratic diagonal share, linear term, per-Gaussian counts); the CMVN part of the state vectors
 *       is left untouched (it does not depend on the weights: kh_ivector_extract_adapt over the accepted frames).
 * use_most_recent_ivector / greedy_ivector_extractor are refused.  KH_ESTATE where the reference asserts
 * (a frame asked for beyond the most recent weight, a frame's weight leaving [0, 1]). */
typedef struct KhIvectorStreams KhIvectorStreams;
KhIvectorStreams *kh_ivector_streams_create(const KhIvectorExtractor *x, const float *feats, int feat_stride,
                                            const int32_t *utt_row_offsets, int n_utts, const double *state_in,
                                            float *ivectors, int ivector_stride);
void kh_ivector_streams_destroy(KhIvectorStreams *s);
int kh_ivector_streams_update_frame_weights(KhIvectorStreams *s, int stream, int n, const int32_t *frames,
                                            const float *delta_weights, int num_frames_ready);
int kh_ivector_streams_get_frames(KhIvectorStreams *s, int n, const int32_t *streams, const int32_t *until_frame);
int kh_ivector_streams_get_stats(const KhIvectorStreams *s, double *states_out);

#ifdef __cplusplus
}
#endif
#endif /* KALDI_HIP_H_ */
