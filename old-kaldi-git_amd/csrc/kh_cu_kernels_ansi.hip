// kh_cu_kernels_ansi.hip — the reference's lower C-ABI seam
// (cudamatrix/cu-kernels-ansi.h launcher names, cublasSgemm) forwarded to the kh_*
// entry points; see include/cu_kernels_ansi_hip.h.
#include "kh_common.h"
#include "../../include/cu_kernels_ansi_hip.h"

#include <atomic>

namespace {

using namespace kh;

std::atomic<int> g_seam_status{0};

inline void Note(int rc) {
  int zero = 0;
  if (rc != KH_OK) g_seam_status.compare_exchange_strong(zero, rc);
}
inline KhMatrixDim Dim(MatrixDim d) { return KhMatrixDim{d.rows, d.cols, d.stride}; }

constexpr int kBlock = 256;

// v[i] = beta v[i] + alpha sum_j M(i, j) N(j, i), arbitrary strides (cu-kernels.cu
// _add_diag_mat_mat; callers AddDiagMatMat / AddDiagMat2 cu-vector.cc:517-580): one wave per i.
__global__ void __launch_bounds__(kBlock)
AddDiagMatMatKernel(float alpha, float *__restrict__ v, int v_dim, const float *__restrict__ M, int m_cols,
                    int m_rs, int m_cs, const float *__restrict__ N, int n_rs, int n_cs, float beta) {
  const int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= v_dim) return;
  float s = 0.0f;
  for (int j = lane; j < m_cols; j += 64)
    s += M[static_cast<size_t>(i) * m_rs + static_cast<size_t>(j) * m_cs] *
         N[static_cast<size_t>(j) * n_rs + static_cast<size_t>(i) * n_cs];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) v[i] = (beta == 0.0f ? 0.0f : beta * v[i]) + alpha * s;
}

// _cuda_comp_obj_deriv (cu-kernels.cu:997-1035): one workgroup, t[0] / t[1] overwritten.
// Duplicated (row, column) pairs are summed (atomicAdd; the reference's kernel races on them).
__global__ void __launch_bounds__(1024)
CompObjDerivKernel(const MatrixElementF *__restrict__ x, int s, const float *__restrict__ z, int z_stride,
                   float *__restrict__ z2, int z2_stride, float *__restrict__ t) {
  __shared__ double red[2][16];
  double objf = 0.0, wsum = 0.0;
  for (int j = threadIdx.x; j < s; j += 1024) {
    const MatrixElementF e = x[j];
    const float p = z[static_cast<size_t>(e.row) * z_stride + e.column];
    objf += static_cast<double>(e.weight * logf(p));
    wsum += static_cast<double>(e.weight);
    atomicAdd(&z2[static_cast<size_t>(e.row) * z2_stride + e.column], e.weight / p);
  }
  objf = kh_wave_sum_d(objf);
  wsum = kh_wave_sum_d(wsum);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = objf; red[1][threadIdx.x >> 6] = wsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < 16; i++) { a += red[0][i]; b += red[1][i]; }
    t[0] = static_cast<float>(a);
    t[1] = static_cast<float>(b);
  }
}

int AddDiagMatMat(float alpha, float *v, int v_dim, const float *M, int M_cols, int M_row_stride, int M_col_stride,
                  const float *N, int N_row_stride, int N_col_stride, float beta) {
  if (int rc = EnsureDevice()) return rc;
  KH_CHECK_ARG(v_dim >= 0 && M_cols >= 0);
  if (v_dim == 0) return KH_OK;
  KH_CHECK_ARG(v != nullptr && M != nullptr && N != nullptr);
  // AddDiagMat2(alpha, M, kNoTrans, beta) (NormalizeComponent, nnet-component.cc:580): N = M^T
  if (N == M && M_col_stride == 1 && N_row_stride == 1 && N_col_stride == M_row_stride)
    return kh_add_diag_mat2(alpha, M, KhMatrixDim{v_dim, M_cols, M_row_stride}, beta, v);
  hipLaunchKernelGGL(AddDiagMatMatKernel, dim3(DivUp(v_dim, kBlock / 64)), dim3(kBlock), 0, Stream(), alpha, v, v_dim,
                     M, M_cols, M_row_stride, M_col_stride, N, N_row_stride, N_col_stride, beta);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int CompObjDeriv(MatrixElementF *x, int s, const float *z, MatrixDim d, float *z2, MatrixDim d2, float *t) {
  if (int rc = EnsureDevice()) return rc;
  KH_CHECK_ARG(s >= 0 && t != nullptr && d.rows == d2.rows && d.cols == d2.cols);
  KH_CHECK_ARG(s == 0 || (x != nullptr && z != nullptr && z2 != nullptr));
  hipLaunchKernelGGL(CompObjDerivKernel, dim3(1), dim3(1024), 0, Stream(), x, s, z, d.stride, z2, d2.stride, t);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int Sgemm(char transa, char transb, int m, int n, int k, float alpha, const float *A, int lda, const float *B, int ldb,
          float beta, float *C, int ldc) {
  const bool ta = transa == 'T' || transa == 't' || transa == 'C' || transa == 'c';
  const bool tb = transb == 'T' || transb == 't' || transb == 'C' || transb == 'c';
  KH_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
  // Column-major C (m x n, ldc) is the row-major matrix C' (n x m, stride ldc), and
  // C' = op(B') op(A') with X' = the same memory read row-major: B' is n x k when
  // transb == 'N' (else k x n, transposed), A' is k x m when transa == 'N'.
  const KhMatrixDim dB{tb ? k : n, tb ? n : k, ldb}, dA{ta ? m : k, ta ? k : m, lda}, dC{n, m, ldc};
  return kh_add_mat_mat(alpha, B, dB, tb ? 1 : 0, A, dA, ta ? 1 : 0, beta, C, dC);
}

// ---- <double> twins (cudaD_*): the same launchers over the kh_*_d entry points (kh_double.hip)
__global__ void __launch_bounds__(kBlock)
AddDiagMatMatDKernel(double alpha, double *__restrict__ v, int v_dim, const double *__restrict__ M, int m_cols,
                     int m_rs, int m_cs, const double *__restrict__ N, int n_rs, int n_cs, double beta) {
  const int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= v_dim) return;
  double s = 0.0;
  for (int j = lane; j < m_cols; j += 64)
    s += M[static_cast<size_t>(i) * m_rs + static_cast<size_t>(j) * m_cs] *
         N[static_cast<size_t>(j) * n_rs + static_cast<size_t>(i) * n_cs];
  s = kh_wave_sum_d(s);
  if (lane == 0) v[i] = (beta == 0.0 ? 0.0 : beta * v[i]) + alpha * s;
}

// _cuda_comp_obj_deriv<double> (cu-kernels.cu:997-1035)
__global__ void __launch_bounds__(1024)
CompObjDerivDKernel(const MatrixElementD *__restrict__ x, int s, const double *__restrict__ z, int z_stride,
                    double *__restrict__ z2, int z2_stride, double *__restrict__ t) {
  __shared__ double red[2][16];
  double objf = 0.0, wsum = 0.0;
  for (int j = threadIdx.x; j < s; j += 1024) {
    const MatrixElementD e = x[j];
    const double p = z[static_cast<size_t>(e.row) * z_stride + e.column];
    objf += e.weight * log(p);
    wsum += e.weight;
    atomicAdd(&z2[static_cast<size_t>(e.row) * z2_stride + e.column], e.weight / p);
  }
  objf = kh_wave_sum_d(objf);
  wsum = kh_wave_sum_d(wsum);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = objf; red[1][threadIdx.x >> 6] = wsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < 16; i++) { a += red[0][i]; b += red[1][i]; }
    t[0] = a;
    t[1] = b;
  }
}

int AddDiagMatMatD(double alpha, double *v, int v_dim, const double *M, int M_cols, int M_row_stride, int M_col_stride,
                   const double *N, int N_row_stride, int N_col_stride, double beta) {
  if (int rc = EnsureDevice()) return rc;
  KH_CHECK_ARG(v_dim >= 0 && M_cols >= 0);
  if (v_dim == 0) return KH_OK;
  KH_CHECK_ARG(v != nullptr && M != nullptr && N != nullptr);
  hipLaunchKernelGGL(AddDiagMatMatDKernel, dim3(DivUp(v_dim, kBlock / 64)), dim3(kBlock), 0, Stream(), alpha, v, v_dim,
                     M, M_cols, M_row_stride, M_col_stride, N, N_row_stride, N_col_stride, beta);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int CompObjDerivD(MatrixElementD *x, int s, const double *z, MatrixDim d, double *z2, MatrixDim d2, double *t) {
  if (int rc = EnsureDevice()) return rc;
  KH_CHECK_ARG(s >= 0 && t != nullptr && d.rows == d2.rows && d.cols == d2.cols);
  KH_CHECK_ARG(s == 0 || (x != nullptr && z != nullptr && z2 != nullptr));
  hipLaunchKernelGGL(CompObjDerivDKernel, dim3(1), dim3(1024), 0, Stream(), x, s, z, d.stride, z2, d2.stride, t);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int Dgemm(char transa, char transb, int m, int n, int k, double alpha, const double *A, int lda, const double *B, int ldb,
          double beta, double *C, int ldc) {
  const bool ta = transa == 'T' || transa == 't' || transa == 'C' || transa == 'c';
  const bool tb = transb == 'T' || transb == 't' || transb == 'C' || transb == 'c';
  KH_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
  const KhMatrixDim dB{tb ? k : n, tb ? n : k, ldb}, dA{ta ? m : k, ta ? k : m, lda}, dC{n, m, ldc};   // (as Sgemm above)
  return kh_add_mat_mat_d(alpha, B, dB, tb ? 1 : 0, A, dA, ta ? 1 : 0, beta, C, dC);
}

}  // namespace

extern "C" {

int kh_cuda_seam_status(void) { return g_seam_status.exchange(0); }

void cudaF_softmax_reduce(size_t, size_t, float *y, const float *x, MatrixDim d, int src_stride) {
  Note(kh_softmax_per_row(y, x, Dim(d), src_stride));
}
void cudaF_log_softmax_reduce(size_t, size_t, float *y, const float *x, MatrixDim d, int src_stride) {
  Note(kh_log_softmax_per_row(y, x, Dim(d), src_stride));
}
void cudaF_copy_rows(KhDim3, KhDim3, float *dst, const float *src, const int32_t *reorder, MatrixDim dst_dim,
                     int src_stride) {
  Note(kh_copy_rows(dst, Dim(dst_dim), src, src_stride, reorder));
}
void cudaF_splice(KhDim3, KhDim3, float *y, const float *x, const int32_t *off, MatrixDim d_out, MatrixDim d_in) {
  Note(kh_splice(y, Dim(d_out), x, Dim(d_in), off, d_in.cols > 0 ? d_out.cols / d_in.cols : 0));
}
void cudaF_group_pnorm(KhDim3, KhDim3, float *y, const float *x, MatrixDim d, int src_stride, int group_size,
                       float power) {
  Note(kh_group_pnorm(y, x, Dim(d), src_stride, group_size, power));
}
void cudaF_add_diag_mat_mat(int, int, float alpha, float *v, int v_dim, const float *M, int M_cols, int M_row_stride,
                            int M_col_stride, const float *N, int N_row_stride, int N_col_stride, int, float beta) {
  Note(AddDiagMatMat(alpha, v, v_dim, M, M_cols, M_row_stride, M_col_stride, N, N_row_stride, N_col_stride, beta));
}
void cudaF_mul_cols_vec(KhDim3, KhDim3, float *mat, const float *scale, MatrixDim d) {
  Note(kh_mul_cols_vec(mat, Dim(d), scale));
}
void cudaF_mul_rows_vec(KhDim3, KhDim3, float *mat, const float *scale, MatrixDim d) {
  Note(kh_mul_rows_vec(mat, Dim(d), scale));
}
void cudaF_copy_rows_from_vec(KhDim3, KhDim3, float *mat_out, MatrixDim d_out, const float *v_in) {
  Note(kh_copy_rows_from_vec(mat_out, Dim(d_out), v_in));
}
void cudaF_add_vec_to_rows(KhDim3, KhDim3, float alpha, const float *row, float beta, float *dst, MatrixDim d) {
  Note(kh_add_vec_to_rows(alpha, row, beta, dst, Dim(d)));
}
void cudaF_apply_exp(KhDim3, KhDim3, float *mat, MatrixDim d) { Note(kh_apply_exp(mat, Dim(d))); }
void cudaF_apply_pow(KhDim3, KhDim3, float *mat, float power, MatrixDim d) { Note(kh_apply_pow(mat, Dim(d), power)); }
void cudaF_apply_floor(KhDim3, KhDim3, float *mat, float floor_val, MatrixDim d) {
  Note(kh_apply_floor(mat, Dim(d), floor_val));
}
void cudaF_scale(KhDim3, KhDim3, float *mat, float value, MatrixDim d) { Note(kh_scale(mat, Dim(d), value)); }
void cudaF_apply_log(KhDim3, KhDim3, float *mat, MatrixDim d) { Note(kh_apply_log(mat, Dim(d))); }
void cudaF_sum_column_ranges(KhDim3, KhDim3, float *data, MatrixDim dim, const float *src_data, MatrixDim src_dim,
                             const Int32Pair *indices) {
  Note(kh_sum_column_ranges(data, Dim(dim), src_data, Dim(src_dim), reinterpret_cast<const int32_t *>(indices)));
}
void cudaF_matrix_lookup(KhDim3, KhDim3, const float *data, MatrixDim dim, const Int32Pair *indices, int indices_size,
                         float *output) {
  Note(kh_matrix_lookup(data, Dim(dim), reinterpret_cast<const int32_t *>(indices), indices_size, output));
}
void cudaF_comp_obj_deriv(KhDim3, KhDim3, MatrixElementF *x, int s, const float *z, MatrixDim d, float *z2,
                          MatrixDim d2, float *t) {
  Note(CompObjDeriv(x, s, z, d, z2, d2, t));
}
void cublasSgemm(char transa, char transb, int m, int n, int k, float alpha, const float *A, int lda, const float *B,
                 int ldb, float beta, float *C, int ldc) {
  Note(Sgemm(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc));
}

// ---- cudaD_* (cu-kernels-ansi.h:187-308) and cublasDgemm (cublas-wrappers.h:31-33)
void cudaD_softmax_reduce(size_t, size_t, double *y, const double *x, MatrixDim d, int src_stride) {
  Note(kh_softmax_per_row_d(y, x, Dim(d), src_stride));
}
void cudaD_log_softmax_reduce(size_t, size_t, double *y, const double *x, MatrixDim d, int src_stride) {
  Note(kh_log_softmax_per_row_d(y, x, Dim(d), src_stride));
}
void cudaD_copy_rows(KhDim3, KhDim3, double *dst, const double *src, const int32_t *reorder, MatrixDim dst_dim, int src_stride) {
  Note(kh_copy_rows_d(dst, Dim(dst_dim), src, src_stride, reorder));
}
void cudaD_splice(KhDim3, KhDim3, double *y, const double *x, const int32_t *off, MatrixDim d_out, MatrixDim d_in) {
  Note(kh_splice_d(y, Dim(d_out), x, Dim(d_in), off, d_in.cols > 0 ? d_out.cols / d_in.cols : 0));
}
void cudaD_group_pnorm(KhDim3, KhDim3, double *y, const double *x, MatrixDim d, int src_stride, int group_size, double power) {
  Note(kh_group_pnorm_d(y, x, Dim(d), src_stride, group_size, power));
}
void cudaD_add_diag_mat_mat(int, int, double alpha, double *v, int v_dim, const double *M, int M_cols, int M_row_stride,
                            int M_col_stride, const double *N, int N_row_stride, int N_col_stride, int, double beta) {
  Note(AddDiagMatMatD(alpha, v, v_dim, M, M_cols, M_row_stride, M_col_stride, N, N_row_stride, N_col_stride, beta));
}
void cudaD_mul_cols_vec(KhDim3, KhDim3, double *mat, const double *scale, MatrixDim d) { Note(kh_mul_cols_vec_d(mat, Dim(d), scale)); }
void cudaD_mul_rows_vec(KhDim3, KhDim3, double *mat, const double *scale, MatrixDim d) { Note(kh_mul_rows_vec_d(mat, Dim(d), scale)); }
void cudaD_copy_rows_from_vec(KhDim3, KhDim3, double *mat_out, MatrixDim d_out, const double *v_in) {
  Note(kh_copy_rows_from_vec_d(mat_out, Dim(d_out), v_in));
}
void cudaD_add_vec_to_rows(KhDim3, KhDim3, double alpha, const double *row, double beta, double *dst, MatrixDim d) {
  Note(kh_add_vec_to_rows_d(alpha, row, beta, dst, Dim(d)));
}
void cudaD_apply_exp(KhDim3, KhDim3, double *mat, MatrixDim d) { Note(kh_apply_exp_d(mat, Dim(d))); }
void cudaD_apply_pow(KhDim3, KhDim3, double *mat, double power, MatrixDim d) { Note(kh_apply_pow_d(mat, Dim(d), power)); }
void cudaD_apply_floor(KhDim3, KhDim3, double *mat, double floor_val, MatrixDim d) { Note(kh_apply_floor_d(mat, Dim(d), floor_val)); }
void cudaD_scale(KhDim3, KhDim3, double *mat, double value, MatrixDim d) { Note(kh_scale_d(mat, Dim(d), value)); }
void cudaD_apply_log(KhDim3, KhDim3, double *mat, MatrixDim d) { Note(kh_apply_log_d(mat, Dim(d))); }
void cudaD_sum_column_ranges(KhDim3, KhDim3, double *data, MatrixDim dim, const double *src_data, MatrixDim src_dim,
                             const Int32Pair *indices) {
  Note(kh_sum_column_ranges_d(data, Dim(dim), src_data, Dim(src_dim), reinterpret_cast<const int32_t *>(indices)));
}
void cudaD_matrix_lookup(KhDim3, KhDim3, const double *data, MatrixDim dim, const Int32Pair *indices, int indices_size,
                         double *output) {
  Note(kh_matrix_lookup_d(data, Dim(dim), reinterpret_cast<const int32_t *>(indices), indices_size, output));
}
void cudaD_comp_obj_deriv(KhDim3, KhDim3, MatrixElementD *x, int s, const double *z, MatrixDim d, double *z2, MatrixDim d2,
                          double *t) {
  Note(CompObjDerivD(x, s, z, d, z2, d2, t));
}
void cublasDgemm(char transa, char transb, int m, int n, int k, double alpha, const double *A, int lda, const double *B,
                 int ldb, double beta, double *C, int ldc) {
  Note(Dgemm(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc));
}

}  // extern "C"
