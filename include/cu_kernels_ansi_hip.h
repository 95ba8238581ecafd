/* The reference's LOWER C-ABI seam, verbatim: the extern "C" launcher set that
 * cudamatrix/cu-matrix.cc, cu-vector.cc and cu-math.cc call
 * (cudamatrix/cu-kernels-ansi.h, 195 prototypes) — here the FP32 subset that is on
 * the nnet2 forward / discriminative hot path, with the reference's
 * names, argument order and argument meaning, plus the legacy cuBLAS entry point its
 * cublas_gemm wrapper forwards to (cudamatrix/cublas-wrappers.h:28-30).
 *
 * A maintainer who keeps cu-matrix.cc as it is links libkaldi_hip.so in place of
 * cu-kernels.o for these symbols.  The launch geometry
 * arguments (Gr, Bl) are accepted and IGNORED: every kernel here chooses its own
 * wave64 geometry.  All pointers are DEVICE pointers; work is enqueued on the
 * library stream (kh_set_stream), exactly like the kh_* entry points of
 * include/kaldi_hip.h that these forward to.
 *
 * Errors: the reference's launchers return void and the caller checks
 * cudaGetLastError().  Here an invalid argument or a failed launch is recorded:
 * kh_cuda_seam_status() returns (and clears) the first non-zero KhStatus since the
 * last call, kh_last_error() the message.
 */
#ifndef KALDI_HIP_CU_KERNELS_ANSI_HIP_H_
#define KALDI_HIP_CU_KERNELS_ANSI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cudamatrix/cu-matrixdim.h:52-56, :84-87, :41-45 — same layouts. */
#ifndef KALDI_HIP_MATRIXDIM_DEFINED
#define KALDI_HIP_MATRIXDIM_DEFINED
typedef struct MatrixDim_ {
  int32_t rows;
  int32_t cols;
  int32_t stride;
} MatrixDim;
typedef struct Int32Pair {
  int32_t first;
  int32_t second;
} Int32Pair;
typedef struct MatrixElementF { /* MatrixElement<float> */
  int32_t row;
  int32_t column;
  float weight;
} MatrixElementF;
typedef struct MatrixElementD { /* MatrixElement<double> */
  int32_t row;
  int32_t column;
  double weight;
} MatrixElementD;
#endif
/* CUDA's dim3 as a C struct (three 32-bit unsigned, passed by value). */
typedef struct KhDim3 {
  unsigned int x, y, z;
} KhDim3;

int kh_cuda_seam_status(void);

/* cu-kernels-ansi.h:131-132 (cu-matrix.cc:1251-1295) */
void cudaF_softmax_reduce(size_t Gr, size_t Bl, float *y, const float *x, MatrixDim d, int src_stride);
void cudaF_log_softmax_reduce(size_t Gr, size_t Bl, float *y, const float *x, MatrixDim d, int src_stride);
/* :65 (cu-matrix.cc:1965-1990) */
void cudaF_copy_rows(KhDim3 Gr, KhDim3 Bl, float *dst, const float *src, const int32_t *reorder,
                     MatrixDim dst_dim, int src_stride);
/* :147 (cu-math.cc:130-165) */
void cudaF_splice(KhDim3 Gr, KhDim3 Bl, float *y, const float *x, const int32_t *off, MatrixDim d_out,
                  MatrixDim d_in);
/* :134 (cu-matrix.cc:1147-1164) */
void cudaF_group_pnorm(KhDim3 Gr, KhDim3 Bl, float *y, const float *x, MatrixDim d, int src_stride,
                       int group_size, float power);
/* :104-106 (cu-vector.cc:517-580): v[i] = beta v[i] + alpha sum_j M(i,j) N(j,i) with
 * explicit row / column strides for both matrices */
void cudaF_add_diag_mat_mat(int Gr, int Bl, float alpha, float *v, int v_dim, const float *M, int M_cols,
                            int M_row_stride, int M_col_stride, const float *N, int N_row_stride,
                            int N_col_stride, int threads_per_element, float beta);
/* :79-80 (cu-matrix.cc:668-713) */
void cudaF_mul_cols_vec(KhDim3 Gr, KhDim3 Bl, float *mat, const float *scale, MatrixDim d);
void cudaF_mul_rows_vec(KhDim3 Gr, KhDim3 Bl, float *mat, const float *scale, MatrixDim d);
/* :144 (cu-matrix.cc:1673-1745), :88 (cu-matrix.cc:916-939) */
void cudaF_copy_rows_from_vec(KhDim3 Gr, KhDim3 Bl, float *mat_out, MatrixDim d_out, const float *v_in);
void cudaF_add_vec_to_rows(KhDim3 Gr, KhDim3 Bl, float alpha, const float *row, float beta, float *dst,
                           MatrixDim d);
/* :59-60, :63, :75-76 (cu-matrix.cc:579-625,1845) */
void cudaF_apply_exp(KhDim3 Gr, KhDim3 Bl, float *mat, MatrixDim d);
void cudaF_apply_pow(KhDim3 Gr, KhDim3 Bl, float *mat, float power, MatrixDim d);
void cudaF_apply_floor(KhDim3 Gr, KhDim3 Bl, float *mat, float floor_val, MatrixDim d);
void cudaF_scale(KhDim3 Gr, KhDim3 Bl, float *mat, float value, MatrixDim d);
void cudaF_apply_log(KhDim3 Gr, KhDim3 Bl, float *mat, MatrixDim d);
/* :159-164 (cu-matrix.cc:1994-2028, :2327) */
void cudaF_sum_column_ranges(KhDim3 Gr, KhDim3 Bl, float *data, MatrixDim dim, const float *src_data,
                             MatrixDim src_dim, const Int32Pair *indices);
void cudaF_matrix_lookup(KhDim3 Gr, KhDim3 Bl, const float *data, MatrixDim dim, const Int32Pair *indices,
                         int indices_size, float *output);
/* :155 (cu-matrix.cc:1198-1248): x = s DEVICE elements; z = output, z2 = deriv;
 * t[0] += sum w log z(r,c), t[1] += sum w (DEVICE, 2 floats) */
void cudaF_comp_obj_deriv(KhDim3 Gr, KhDim3 Bl, MatrixElementF *x, int s, const float *z, MatrixDim d, float *z2,
                          MatrixDim d2, float *t);
/* legacy cuBLAS v1 SGEMM, column-major (cublas-wrappers.h:28-30; caller cu-matrix.cc:947-982) */
void cublasSgemm(char transa, char transb, int m, int n, int k, float alpha, const float *A, int lda,
                 const float *B, int ldb, float beta, float *C, int ldc);

/* ---- the <double> twins of the same subset (cu-kernels-ansi.h:187-308; CuMatrix<double> is instantiated by
 * cu-matrix.cc:2415-2418) and cublasDgemm (cublas-wrappers.h:31-33).  Same argument meaning as the cudaF_* above. */
void cudaD_softmax_reduce(size_t Gr, size_t Bl, double *y, const double *x, MatrixDim d, int src_stride);
void cudaD_log_softmax_reduce(size_t Gr, size_t Bl, double *y, const double *x, MatrixDim d, int src_stride);
void cudaD_copy_rows(KhDim3 Gr, KhDim3 Bl, double *dst, const double *src, const int32_t *reorder, MatrixDim dst_dim,
                     int src_stride);
void cudaD_splice(KhDim3 Gr, KhDim3 Bl, double *y, const double *x, const int32_t *off, MatrixDim d_out, MatrixDim d_in);
void cudaD_group_pnorm(KhDim3 Gr, KhDim3 Bl, double *y, const double *x, MatrixDim d, int src_stride, int group_size,
                       double power);
void cudaD_add_diag_mat_mat(int Gr, int Bl, double alpha, double *v, int v_dim, const double *M, int M_cols,
                            int M_row_stride, int M_col_stride, const double *N, int N_row_stride, int N_col_stride,
                            int threads_per_element, double beta);
void cudaD_mul_cols_vec(KhDim3 Gr, KhDim3 Bl, double *mat, const double *scale, MatrixDim d);
void cudaD_mul_rows_vec(KhDim3 Gr, KhDim3 Bl, double *mat, const double *scale, MatrixDim d);
void cudaD_copy_rows_from_vec(KhDim3 Gr, KhDim3 Bl, double *mat_out, MatrixDim d_out, const double *v_in);
void cudaD_add_vec_to_rows(KhDim3 Gr, KhDim3 Bl, double alpha, const double *row, double beta, double *dst, MatrixDim d);
void cudaD_apply_exp(KhDim3 Gr, KhDim3 Bl, double *mat, MatrixDim d);
void cudaD_apply_pow(KhDim3 Gr, KhDim3 Bl, double *mat, double power, MatrixDim d);
void cudaD_apply_floor(KhDim3 Gr, KhDim3 Bl, double *mat, double floor_val, MatrixDim d);
void cudaD_scale(KhDim3 Gr, KhDim3 Bl, double *mat, double value, MatrixDim d);
void cudaD_apply_log(KhDim3 Gr, KhDim3 Bl, double *mat, MatrixDim d);
void cudaD_sum_column_ranges(KhDim3 Gr, KhDim3 Bl, double *data, MatrixDim dim, const double *src_data, MatrixDim src_dim,
                             const Int32Pair *indices);
void cudaD_matrix_lookup(KhDim3 Gr, KhDim3 Bl, const double *data, MatrixDim dim, const Int32Pair *indices, int indices_size,
                         double *output);
void cudaD_comp_obj_deriv(KhDim3 Gr, KhDim3 Bl, MatrixElementD *x, int s, const double *z, MatrixDim d, double *z2,
                          MatrixDim d2, double *t);
void cublasDgemm(char transa, char transb, int m, int n, int k, double alpha, const double *A, int lda, const double *B,
                 int ldb, double beta, double *C, int ldc);

#ifdef __cplusplus
}
#endif
#endif /* KALDI_HIP_CU_KERNELS_ANSI_HIP_H_ */
