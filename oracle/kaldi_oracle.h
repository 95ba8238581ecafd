/* kaldi_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the acoustic-scoring + lattice-decoding hot path
 * of the reference (vimalmanohar/old-kaldi-git).  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/src).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product library (libkaldi_hip.so) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - matrix / nnet2-forward / DiagGmm functions (ko_* in kaldi_oracle.cc):
 *     PINNED against the reference's own CPU code compiled from
 *     /root/reference (oracle/_ref/libkaldi_ref.so) and against golden
 *     vectors generated from it (tests/golden/, tests/golden/make_golden.py).
 *   - decoder (decoder_oracle.cc) and lattice forward-backward
 *     (lattice_oracle.cc): PARITY UNPINNED by reference tests — src/decoder
 *     has 0 unit tests, LatticeForwardBackward has none, and neither compiles
 *     here (needs OpenFst 1.3.4, tools/Makefile:6, absent).  They are pinned
 *     by brute-force path enumeration on small graphs and the reference's own
 *     algebraic self-checks instead.
 *
 * All matrices are row-major float32 with an explicit stride in elements
 * (matrix/kaldi-matrix.h:58-91, cudamatrix/cu-matrixdim.h:49-53).
 */
#ifndef KALDI_ORACLE_H_
#define KALDI_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- a1 ---- */
/* MatrixBase::AddMatMat, matrix/kaldi-matrix.cc:160-175 (cblas_Xgemm):
 * C = alpha * op(A) * op(B) + beta * C.  transX != 0 means kTrans.
 * a_rows/a_cols are the dimensions of A as stored.  Accumulation: one float
 * fmaf chain per output element in increasing k, then
 * c = fl(fl(beta*c) + fl(alpha*acc)) (beta == 0 ignores the old c, as BLAS). */
void ko_add_mat_mat(float alpha, const float *A, int a_rows, int a_cols,
                    int a_stride, int transA, const float *B, int b_rows,
                    int b_cols, int b_stride, int transB, float beta, float *C,
                    int c_rows, int c_cols, int c_stride);

/* ---------------------------------------------------------------- a2 ---- */
/* CuMatrixBase::ApplySoftMaxPerRow CPU branch cu-matrix.cc:1263-1270 ->
 * VectorBase::ApplySoftMax, matrix/kaldi-vector.cc:840-847. */
void ko_softmax_per_row(const float *src, int rows, int cols, int src_stride,
                        float *dst, int dst_stride);
/* ApplyLogSoftMaxPerRow cu-matrix.cc:1274-1295 ->
 * VectorBase::ApplyLogSoftMax kaldi-vector.cc:849-860. */
void ko_log_softmax_per_row(const float *src, int rows, int cols,
                            int src_stride, float *dst, int dst_stride);

/* ---------------------------------------------------------------- a3 ---- */
/* MatrixBase::CopyRows, matrix/kaldi-matrix.cc:2570-2584. */
void ko_copy_rows(float *dst, int rows, int cols, int dst_stride,
                  const float *src, int src_stride, const int32_t *indices);

/* ---------------------------------------------------------------- a4 ---- */
/* cu::Splice CPU branch, cudamatrix/cu-math.cc:147-163:
 * tgt[r, j*D + c] = src[clamp(r + frame_offsets[j], 0, R-1), c]. */
void ko_splice(const float *src, int rows, int cols, int src_stride,
               const int32_t *frame_offsets, int n_offsets, float *tgt,
               int tgt_stride);

/* ---------------------------------------------------------------- a5 ---- */
/* MatrixBase::GroupPnorm kaldi-matrix.cc:2512-2520 + VectorBase::Norm
 * kaldi-vector.cc:508-545. dst is rows x (src_cols / group). */
void ko_group_pnorm(const float *src, int rows, int src_cols, int src_stride,
                    float power, float *dst, int dst_cols, int dst_stride);

/* ---------------------------------------------------------------- a6 ---- */
/* NormalizeComponent::Propagate nnet2/nnet-component.cc:576-588 (CopyFromMat,
 * AddDiagMat2 kaldi-vector.cc:1271-1281, ApplyFloor(2^-66), ApplyPow(-0.5),
 * MulRowsVec). */
void ko_normalize(const float *src, int rows, int cols, int src_stride,
                  float *dst, int dst_stride);
/* VectorBase::AddDiagMat2 (kNoTrans) kaldi-vector.cc:1271-1281. */
void ko_add_diag_mat2(float alpha, const float *M, int rows, int cols,
                      int stride, float beta, float *v);
/* MatrixBase::MulRowsVec kaldi-matrix.cc (scale row i by s[i]). */
void ko_mul_rows_vec(float *M, int rows, int cols, int stride, const float *s);
/* MatrixBase::MulColsVec (scale col j by s[j]). */
void ko_mul_cols_vec(float *M, int rows, int cols, int stride, const float *s);

/* ---------------------------------------------------------------- a7 ---- */
/* CopyRowsFromVec cu-matrix.cc:1673-1745 (vector of dim cols broadcast). */
void ko_copy_rows_from_vec(float *M, int rows, int cols, int stride,
                           const float *v);
/* AddVecToRows cu-matrix.cc:916-939: M = beta*M + alpha*v[c]. */
void ko_add_vec_to_rows(float alpha, const float *v, float beta, float *M,
                        int rows, int cols, int stride);
/* ApplyFloor cu-matrix.cc:1845, ApplyLog :600, Scale :579, ApplyExp,
 * ApplyPow (kaldi-matrix.cc ApplyPow -> per-row VectorBase::ApplyPow
 * kaldi-vector.cc:448-469). */
void ko_apply_floor(float *M, int rows, int cols, int stride, float floor_val);
void ko_apply_log(float *M, int rows, int cols, int stride);
void ko_apply_exp(float *M, int rows, int cols, int stride);
void ko_apply_pow(float *M, int rows, int cols, int stride, float power);
void ko_scale(float *M, int rows, int cols, int stride, float alpha);
/* SumColumnRanges cu-matrix.cc:1994-2028 CPU branch: dst[r,c] =
 * sum_{j in [start_c, end_c)} src[r,j]; ranges = 2*dst_cols int32. */
void ko_sum_column_ranges(float *dst, int rows, int dst_cols, int dst_stride,
                          const float *src, int src_stride,
                          const int32_t *ranges);
/* CuMatrixBase::Lookup cu-matrix.cc:2327-...: out[k] = M[idx[2k], idx[2k+1]]. */
void ko_matrix_lookup(const float *M, int rows, int cols, int stride,
                      const int32_t *row_col_pairs, int n, float *out);

/* ---------------------------------------------------------------- a8 ---- */
/* nnet2 forward: Component::Propagate per component
 * (nnet2/nnet-component.cc: Splice 2628-2708, FixedAffine 3333-3343,
 * Affine 1212-1224, Pnorm 518-527, Normalize 576-588, Softmax 926-943,
 * SumGroup 2491-2499, FixedScale, FixedBias) driven as NnetComputer does
 * (nnet2/nnet-compute.cc:63-108) with Nnet::ComputeChunkInfo
 * (nnet2/nnet-nnet.cc:65-112). */
enum KoComponentType {
  KO_SPLICE = 1,
  KO_FIXED_AFFINE = 2,
  KO_AFFINE = 3, /* Affine / AffineComponentPreconditioned(Online): same fwd */
  KO_PNORM = 4,
  KO_NORMALIZE = 5,
  KO_SOFTMAX = 6,
  KO_SUM_GROUP = 7,
  KO_FIXED_SCALE = 8,
  KO_FIXED_BIAS = 9
};

typedef struct KoComponent {
  int32_t type;
  int32_t input_dim;
  int32_t output_dim;
  const float *linear; /* [output_dim x input_dim], stride = input_dim */
  const float *bias;   /* [output_dim] (also scales / bias for FixedScale/Bias) */
  const int32_t *context; /* splice: context offsets (sorted) */
  int32_t n_context;
  int32_t const_dim;      /* splice: const_component_dim */
  float p;                /* pnorm power */
  const int32_t *sizes;   /* sum-group: group sizes, n_sizes == output_dim */
  int32_t n_sizes;
} KoComponent;

/* NnetComputation(nnet, feats, pad_input, out), nnet-compute.cc:159-166.
 * feats: T x input_dim.  Returns number of output rows (T if pad != 0, else
 * T - left - right), or <0 on error.  out must hold that many rows x
 * output_dim of last component (stride out_stride). */
int ko_nnet_forward(const KoComponent *comps, int n_comps, const float *feats,
                    int T, int feat_stride, int pad_input, float *out,
                    int out_stride);
int ko_nnet_left_context(const KoComponent *comps, int n_comps);
int ko_nnet_right_context(const KoComponent *comps, int n_comps);

/* DecodableAmNnet ctor, nnet2/decodable-am-nnet.h:39-73: forward(pad=true),
 * ApplyFloor(1e-20), ApplyLog, AddVecToRows(-1, log(priors)), Scale(acwt). */
int ko_decodable_am_nnet(const KoComponent *comps, int n_comps,
                         const float *priors, float prob_scale,
                         const float *feats, int T, int feat_stride,
                         float *log_probs, int out_stride);

/* ---------------------------------------------------------------- a9 ---- */
/* DiagGmm::ComputeGconsts gmm/diag-gmm.cc:114-152. Returns num_bad. */
int ko_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                           const float *inv_vars, int num_mix, int dim,
                           float *gconsts);
/* DiagGmm::LogLikelihoods(Matrix) gmm/diag-gmm.cc:546-562: loglikes[T x M]. */
void ko_diag_gmm_loglikes(const float *data, int T, int dim, int data_stride,
                          const float *gconsts, const float *means_invvars,
                          const float *inv_vars, int num_mix, float *loglikes,
                          int ll_stride);
/* VectorBase::LogSumExp(prune) matrix/kaldi-vector.cc:745-763. */
float ko_log_sum_exp(const float *v, int dim, float prune);
/* Dense frame x pdf matrix as gmm-compute-likes builds it
 * (gmmbin/gmm-compute-likes.cc:70-77 -> AmDiagGmm::LogLikelihood ->
 * DiagGmm::LogLikelihood diag-gmm.cc:517-526) with the decodable's prune
 * (gmm/decodable-am-diag-gmm.cc:58-64).  All pdfs' Gaussians concatenated;
 * pdf j owns mixtures [pdf_offsets[j], pdf_offsets[j+1]). */
void ko_am_gmm_loglikes(const float *data, int T, int dim, int data_stride,
                        const float *gconsts, const float *means_invvars,
                        const float *inv_vars, const int32_t *pdf_offsets,
                        int num_pdfs, float log_sum_exp_prune, float *out,
                        int out_stride);

#ifdef __cplusplus
}
#endif
#endif /* KALDI_ORACLE_H_ */
