// kh_double.hip — the <double> instantiation of the hot CuMatrix primitives (SURVEY.md §8b:
// cudamatrix/cu-matrix.cc:2415-2418 instantiates CuMatrix<double> / CuMatrixBase<double>, and
// cu-kernels-ansi.h carries a cudaD_* twin of every cudaF_* launcher; the reference's own unit tests run
// CudaMatrixUnitTest<double>() next to <float>, cu-matrix-test.cc).  The nnet2 decode path itself is FP32
// (KALDI_DOUBLEPRECISION=0); these are the same operations for callers that hold CuMatrix<double>:
//   kh_add_mat_mat_d            AddMatMat       cu-matrix.cc:947-982   fp64 MFMA (v_mfma_f64_16x16x4_f64)
//   kh_[log_]softmax_per_row_d  ApplySoftMaxPerRow / ApplyLogSoftMaxPerRow :1251-1295
//   kh_copy_rows_d, kh_splice_d CopyRows :1965-1990, cu::Splice cu-math.cc:130-165
//   kh_group_pnorm_d            GroupPnorm :1147-1164 (VectorBase::Norm kaldi-vector.cc:508-545)
//   kh_add_diag_mat2_d, kh_mul_rows_vec_d, kh_mul_cols_vec_d, kh_copy_rows_from_vec_d, kh_add_vec_to_rows_d,
//   kh_apply_{floor,log,exp,pow}_d, kh_scale_d, kh_sum_column_ranges_d, kh_matrix_lookup_d
// Layout and argument meaning are those of the float entry points (include/kaldi_hip.h); all HBM-bound
// kernels put the column on the lane (coalesced rows), per-row reductions use wave shuffles.
#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ double WaveMaxD(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

bool DimOk(const KhMatrixDim &d) { return d.rows >= 0 && d.cols >= 0 && d.stride >= d.cols; }

// ---- GEMM: C = alpha op(A) op(B) + beta C, row-major with strides ------------------------------------------------
// 64 x 128 tile per 256-thread workgroup (4 waves x (4 x 2) MFMA tiles of 16 x 16), BK = 16, operands staged
// through LDS as [k][m] / [k][n] images so that the fragment reads are conflict-free whatever the transposes.
// The loaders walk memory along the contiguous dimension of the operand AS STORED.
constexpr int kGM = 64, kGN = 128, kGK = 16;
typedef double KhD4 __attribute__((ext_vector_type(4)));

struct GemmD {
  const double *A, *B;
  double *C;
  int M, N, K, lda, ldb, ldc, ta, tb;
  double alpha, beta;
};

__global__ void __launch_bounds__(256) GemmF64Kernel(GemmD g) {
  __shared__ double sa[kGK][kGM + 1];
  __shared__ double sb[kGK][kGN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bm = blockIdx.y * kGM, bn = blockIdx.x * kGN;
  KhD4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = KhD4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < g.K; k0 += kGK) {
    // A tile: 64 (m) x 16 (k) = 1024 elements, 4 per thread
    if (!g.ta) {   // stored M x K: contiguous along k
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int e = p * 256 + tid, m = e >> 4, k = e & 15;
        const int gm = bm + m, gk = k0 + k;
        sa[k][m] = (gm < g.M && gk < g.K) ? g.A[static_cast<size_t>(gm) * g.lda + gk] : 0.0;
      }
    } else {       // stored K x M: contiguous along m
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int e = p * 256 + tid, k = e >> 6, m = e & 63;
        const int gm = bm + m, gk = k0 + k;
        sa[k][m] = (gm < g.M && gk < g.K) ? g.A[static_cast<size_t>(gk) * g.lda + gm] : 0.0;
      }
    }
    // B tile: 16 (k) x 128 (n) = 2048 elements, 8 per thread
    if (!g.tb) {   // stored K x N: contiguous along n
#pragma unroll
      for (int p = 0; p < 8; p++) {
        const int e = p * 256 + tid, k = e >> 7, n = e & 127;
        const int gn = bn + n, gk = k0 + k;
        sb[k][n] = (gn < g.N && gk < g.K) ? g.B[static_cast<size_t>(gk) * g.ldb + gn] : 0.0;
      }
    } else {       // stored N x K: contiguous along k
#pragma unroll
      for (int p = 0; p < 8; p++) {
        const int e = p * 256 + tid, n = e >> 4, k = e & 15;
        const int gn = bn + n, gk = k0 + k;
        sb[k][n] = (gn < g.N && gk < g.K) ? g.B[static_cast<size_t>(gn) * g.ldb + gk] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 4) {
      double af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = sa[ks + (lane >> 4)][i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; j++) bf[j] = sb[ks + (lane >> 4)][wave * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gm = bm + i * 16 + (lane >> 4) + 4 * r, gn = bn + wave * 32 + j * 16 + (lane & 15);
        if (gm < g.M && gn < g.N) {
          double *c = g.C + static_cast<size_t>(gm) * g.ldc + gn;
          // beta == 0 overwrites (cublas semantics: C may hold NaN / uninitialised memory, cu-matrix.cc:960-975)
          *c = g.alpha * acc[i][j][r] + (g.beta == 0.0 ? 0.0 : g.beta * *c);
        }
      }
}

// ---- softmax / log-softmax: one wave per row (max, exp and sum in double, ApplySoftMax kaldi-vector.cc:840-847) ----
template <bool LOG>
__global__ void __launch_bounds__(kBlock)
SoftmaxDKernel(double *__restrict__ y, const double *__restrict__ x, int rows, int cols, int y_stride, int x_stride) {
  const int r = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const double *xr = x + static_cast<size_t>(r) * x_stride;
  double *yr = y + static_cast<size_t>(r) * y_stride;
  double m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmax(m, xr[c]);
  m = WaveMaxD(m);
  double s = 0.0;
  for (int c = lane; c < cols; c += 64) s += exp(xr[c] - m);
  s = kh_wave_sum_d(s);
  if (LOG) {
    const double ls = -1.0 * log(s);
    for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - m) + ls;
  } else {
    const double inv = 1.0 / s;
    for (int c = lane; c < cols; c += 64) yr[c] = exp(xr[c] - m) * inv;
  }
}

// ---- generic 2-D element-wise launch: column on the lane ---------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(kBlock) Map2DD(int rows, int cols, F f) {
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) f(r, c);
}

template <class F>
int LaunchMap(int rows, int cols, F f) {
  if (rows <= 0 || cols <= 0) return KH_OK;
  const int bx = cols >= 256 ? 256 : 64;
  int gx = DivUp(cols, bx);
  if (gx > 64) gx = 64;
  int gy = rows;
  const int max_blocks = NumCUs() * 16;
  if (static_cast<int64_t>(gx) * gy > max_blocks) gy = max_blocks / gx > 0 ? max_blocks / gx : 1;
  if (gy > 65535) gy = 65535;
  hipLaunchKernelGGL(Map2DD<F>, dim3(gx, gy), dim3(bx), 0, Stream(), rows, cols, f);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

// VectorBase<double>::Norm(p) kaldi-vector.cc:508-545 over one group
__device__ __forceinline__ double GroupNorm(const double *g, int group, double p) {
  if (p == 2.0) {
    double s = 0.0;
    for (int j = 0; j < group; j++) s += g[j] * g[j];
    return sqrt(s);
  }
  if (p == 1.0) {
    double s = 0.0;
    for (int j = 0; j < group; j++) s += fabs(g[j]);
    return s;
  }
  if (p == 0.0) {
    double s = 0.0;
    for (int j = 0; j < group; j++) s += g[j] != 0.0 ? 1.0 : 0.0;
    return s;
  }
  double s = 0.0, mx = 0.0;
  bool ok = true;
  for (int j = 0; j < group; j++) {
    const double a = fabs(g[j]);
    mx = fmax(mx, a);
    const double t = pow(a, p);
    if (t == HUGE_VAL) ok = false;   // :531 "HUGE_VAL is what pow returns on error"
    s += t;
  }
  const double ip = 1.0 / p;
  if (ok) return pow(s, ip);
  // :536-542 rescue: scale by the largest magnitude, take the norm, scale back
  const double sc = 1.0 / mx;
  double s2 = 0.0;
  for (int j = 0; j < group; j++) s2 += pow(fabs(g[j] * sc), p);
  return pow(s2, ip) * mx;
}

__global__ void __launch_bounds__(kBlock)
AddDiagMat2DKernel(double alpha, const double *__restrict__ M, int rows, int cols, int stride, double beta, double *__restrict__ v) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < rows; r += gridDim.x * (kBlock / 64)) {
    const double *xr = M + static_cast<size_t>(r) * stride;
    double s = 0.0;
    for (int c = lane; c < cols; c += 64) s += xr[c] * xr[c];
    s = kh_wave_sum_d(s);
    if (lane == 0) v[r] = (beta == 0.0 ? 0.0 : beta * v[r]) + alpha * s;
  }
}

__global__ void __launch_bounds__(kBlock)
LookupDKernel(const double *__restrict__ M, int rows, int cols, int stride, const int32_t *__restrict__ pairs, int n,
              double *__restrict__ out) {
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
    const int r = pairs[2 * k], c = pairs[2 * k + 1];
    out[k] = (r >= 0 && r < rows && c >= 0 && c < cols) ? M[static_cast<size_t>(r) * stride + c] : __longlong_as_double(0x7ff8000000000000ll);
  }
}

}  // namespace

extern "C" {

int kh_add_mat_mat_d(double alpha, const double *A, KhMatrixDim dA, int transA, const double *B, KhMatrixDim dB, int transB,
                     double beta, double *C, KhMatrixDim dC) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(A && B && C && DimOk(dA) && DimOk(dB) && DimOk(dC));
  const int m = transA ? dA.cols : dA.rows, k = transA ? dA.rows : dA.cols;
  const int kb = transB ? dB.cols : dB.rows, n = transB ? dB.rows : dB.cols;
  // the reference asserts the same (cu-matrix.cc:951-958)
  KH_CHECK_ARG(k == kb && m == dC.rows && n == dC.cols);
  if (m == 0 || n == 0) return KH_OK;
  GemmD g{A, B, C, m, n, k, dA.stride, dB.stride, dC.stride, transA ? 1 : 0, transB ? 1 : 0, alpha, beta};
  hipLaunchKernelGGL(GemmF64Kernel, dim3(DivUp(n, kGN), DivUp(m, kGM)), dim3(256), 0, Stream(), g);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int kh_softmax_per_row_d(double *y, const double *x, KhMatrixDim d, int src_stride) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(DimOk(d) && src_stride >= d.cols && y && x);
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  hipLaunchKernelGGL(SoftmaxDKernel<false>, dim3(DivUp(d.rows, kBlock / 64)), dim3(kBlock), 0, Stream(), y, x, d.rows, d.cols,
                     d.stride, src_stride);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int kh_log_softmax_per_row_d(double *y, const double *x, KhMatrixDim d, int src_stride) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(DimOk(d) && src_stride >= d.cols && y && x);
  if (d.rows == 0 || d.cols == 0) return KH_OK;
  hipLaunchKernelGGL(SoftmaxDKernel<true>, dim3(DivUp(d.rows, kBlock / 64)), dim3(kBlock), 0, Stream(), y, x, d.rows, d.cols,
                     d.stride, src_stride);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

int kh_copy_rows_d(double *dst, KhMatrixDim dd, const double *src, int src_stride, const int32_t *indices) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(dst && src && indices && DimOk(dd) && src_stride >= dd.cols);
  const int ds = dd.stride;
  return LaunchMap(dd.rows, dd.cols, [=] __device__(int r, int c) {
    const int s = indices[r];
    dst[static_cast<size_t>(r) * ds + c] = s < 0 ? 0.0 : src[static_cast<size_t>(s) * src_stride + c];
  });
}

int kh_splice_d(double *y, KhMatrixDim d_out, const double *x, KhMatrixDim d_in, const int32_t *frame_offsets, int n_offsets) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(y && x && frame_offsets && DimOk(d_out) && DimOk(d_in) && n_offsets > 0 && d_out.rows == d_in.rows &&
               d_out.cols == d_in.cols * n_offsets);
  const int D = d_in.cols, R = d_in.rows, ys = d_out.stride, xs = d_in.stride;
  return LaunchMap(d_out.rows, d_out.cols, [=] __device__(int r, int c) {
    const int k = c / D, cc = c - k * D;
    int s = r + frame_offsets[k];
    s = s < 0 ? 0 : (s >= R ? R - 1 : s);
    y[static_cast<size_t>(r) * ys + c] = x[static_cast<size_t>(s) * xs + cc];
  });
}

int kh_group_pnorm_d(double *y, const double *x, KhMatrixDim d, int src_stride, int group_size, double power) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(y && x && DimOk(d) && group_size > 0 && src_stride >= d.cols * group_size && power >= 0.0);
  const int ys = d.stride;
  return LaunchMap(d.rows, d.cols, [=] __device__(int r, int c) {
    y[static_cast<size_t>(r) * ys + c] = GroupNorm(x + static_cast<size_t>(r) * src_stride + c * group_size, group_size, power);
  });
}

int kh_add_diag_mat2_d(double alpha, const double *M, KhMatrixDim d, double beta, double *v) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(M && v && DimOk(d));
  if (d.rows == 0) return KH_OK;
  hipLaunchKernelGGL(AddDiagMat2DKernel, dim3(std::min(DivUp(d.rows, kBlock / 64), NumCUs() * 8)), dim3(kBlock), 0, Stream(), alpha, M,
                     d.rows, d.cols, d.stride, beta, v);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

#define KH_MAP_D(NAME, ARGS, CHECK, BODY)                                            \
  int NAME ARGS {                                                                    \
    int rc = EnsureDevice();                                                         \
    if (rc) return rc;                                                               \
    KH_CHECK_ARG(M && DimOk(d) && (CHECK));                                          \
    const int st = d.stride;                                                         \
    return LaunchMap(d.rows, d.cols, [=] __device__(int r, int c) {                  \
      double &e = M[static_cast<size_t>(r) * st + c];                                \
      BODY;                                                                          \
    });                                                                              \
  }

KH_MAP_D(kh_mul_rows_vec_d, (double *M, KhMatrixDim d, const double *scale), scale != nullptr, e *= scale[r])
KH_MAP_D(kh_mul_cols_vec_d, (double *M, KhMatrixDim d, const double *scale), scale != nullptr, e *= scale[c])
KH_MAP_D(kh_copy_rows_from_vec_d, (double *M, KhMatrixDim d, const double *v), v != nullptr, e = v[c])
KH_MAP_D(kh_add_vec_to_rows_d, (double alpha, const double *v, double beta, double *M, KhMatrixDim d), v != nullptr,
         e = alpha * v[c] + beta * e)
KH_MAP_D(kh_apply_floor_d, (double *M, KhMatrixDim d, double floor_val), true, e = e < floor_val ? floor_val : e)
KH_MAP_D(kh_apply_log_d, (double *M, KhMatrixDim d), true, e = log(e))
KH_MAP_D(kh_apply_exp_d, (double *M, KhMatrixDim d), true, e = exp(e))
KH_MAP_D(kh_scale_d, (double *M, KhMatrixDim d, double alpha), true, e *= alpha)
// ApplyPow: MatrixBase::ApplyPow -> VectorBase::ApplyPow kaldi-vector.cc:483-505: 1 = nothing, 2 = square, 0.5 = sqrt
// (negative inputs are an error there: NaN here), else pow()
KH_MAP_D(kh_apply_pow_d, (double *M, KhMatrixDim d, double power), true,
         e = power == 1.0 ? e : (power == 2.0 ? e * e : (power == 0.5 ? sqrt(e) : pow(e, power))))
#undef KH_MAP_D

int kh_sum_column_ranges_d(double *y, KhMatrixDim d, const double *x, KhMatrixDim d_src, const int32_t *ranges) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(y && x && ranges && DimOk(d) && DimOk(d_src) && d.rows == d_src.rows);
  const int ys = d.stride, xs = d_src.stride;
  return LaunchMap(d.rows, d.cols, [=] __device__(int r, int c) {
    const int s = ranges[2 * c], e = ranges[2 * c + 1];
    const double *xr = x + static_cast<size_t>(r) * xs;
    double sum = 0.0;
    for (int j = s; j < e; j++) sum += xr[j];
    y[static_cast<size_t>(r) * ys + c] = sum;
  });
}

int kh_matrix_lookup_d(const double *M, KhMatrixDim d, const int32_t *pairs, int n, double *out) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(M && DimOk(d) && n >= 0 && (n == 0 || (pairs && out)));
  if (n == 0) return KH_OK;
  hipLaunchKernelGGL(LookupDKernel, dim3(std::min(DivUp(n, kBlock), NumCUs() * 8)), dim3(kBlock), 0, Stream(), M, d.rows, d.cols,
                     d.stride, pairs, n, out);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

}  // extern "C"
