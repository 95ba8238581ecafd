// kh_gmm.hip — DiagGmm::LogLikelihoods over batched frames.
//
// Replaces DiagGmm::LogLikelihoods(const MatrixBase&, Matrix*)
// (gmm/diag-gmm.cc:546-562: two sgemm calls on [x | x^2]), ComputeGconsts
// (:114-152, host), and the per-(frame,pdf) scoring of
// DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased
// (gmm/decodable-am-diag-gmm.cc:28-71) + VectorBase::LogSumExp
// (matrix/kaldi-vector.cc:745-763) evaluated densely for every (frame, pdf).
//
// Kernel 1 (per-Gaussian log-likelihoods): frames on the lanes.  A wave owns 64
// frames whose features [x | x*x] live in registers; the Gaussian parameters
// (means_invvars / inv_vars rows, gconst) are wave-uniform, so they are fetched
// through the scalar cache and used as SGPR operands of the FMAs — no LDS
// traffic in the inner loop.  Each 64-frame x 64-Gaussian result tile is
// transposed through LDS so the T x M matrix is written in full coalesced rows.
// Numerics per element (bit-identical to the CPU oracle):
//   a1 = fmaf-chain_d(x_d, mi_d); a2 = fmaf-chain_d(x_d*x_d, iv_d);
//   ll = (g + a1) + (-0.5f * a2)
// Kernel 2: one thread per (frame, pdf): exact LogSumExp with the reference's
// cutoff and double accumulation.
#include <cfloat>
#include <cmath>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kThreads = 256;
constexpr int kTileM = 64;

template <int DP>
__global__ void __launch_bounds__(kThreads)
GmmLoglikesKernel(const float *__restrict__ data, int T, int D, int data_stride,
                  const float *__restrict__ gconsts, const float *__restrict__ mi,
                  const float *__restrict__ iv, int M, float *__restrict__ out,
                  int out_stride, int m_per_block) {
  __shared__ float tile[kThreads / 64][kTileM][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * (kThreads / 64) + wave) * 64;
  if (t0 >= T) return;  // wave-uniform
  const int t = t0 + lane;
  float x[DP], xx[DP];
#pragma unroll
  for (int d = 0; d < DP; d++) {
    float v = (d < D && t < T) ? data[static_cast<size_t>(t) * data_stride + d] : 0.f;
    x[d] = v;
    xx[d] = v * v;  // data_sq.ApplyPow(2.0)
  }
  const int m_begin = blockIdx.y * m_per_block;
  int m_end = m_begin + m_per_block;
  if (m_end > M) m_end = M;
  float(*tl)[65] = tile[wave];
  for (int mb = m_begin; mb < m_end; mb += kTileM) {
    const int nm = (m_end - mb) < kTileM ? (m_end - mb) : kTileM;
    for (int j = 0; j < nm; j++) {
      const int m = __builtin_amdgcn_readfirstlane(mb + j);
      const float *mir = mi + static_cast<size_t>(m) * D;
      const float *ivr = iv + static_cast<size_t>(m) * D;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int d = 0; d < DP; d++) {
        if (d < D) {
          a1 = fmaf(x[d], mir[d], a1);
          a2 = fmaf(xx[d], ivr[d], a2);
        }
      }
      const float ll = (gconsts[m] + a1) + (-0.5f * a2);
      tl[j][lane] = ll;  // [gaussian][frame]
    }
    // wave-private tile: no block barrier needed, only LDS visibility in-wave
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 64; r++) {
      const int tr = t0 + r;
      if (tr >= T) break;
      if (lane < nm)
        out[static_cast<size_t>(tr) * out_stride + mb + lane] = tl[lane][r];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// VectorBase::LogSumExp(prune) kaldi-vector.cc:745-763 per (frame, pdf).
__global__ void __launch_bounds__(kThreads)
GmmPdfLseKernel(const float *__restrict__ ll, int T, int ll_stride,
                const int32_t *__restrict__ pdf_offsets, int num_pdfs, float prune,
                float min_log_diff, float *__restrict__ out, int out_stride) {
  for (int t = blockIdx.y; t < T; t += gridDim.y) {
    const float *row = ll + static_cast<size_t>(t) * ll_stride;
    for (int j = blockIdx.x * kThreads + threadIdx.x; j < num_pdfs;
         j += gridDim.x * kThreads) {
      const int s = pdf_offsets[j], e = pdf_offsets[j + 1];
      float mx = -INFINITY;
      for (int m = s; m < e; m++) mx = fmaxf(mx, row[m]);
      float cutoff = mx + min_log_diff;
      if (prune > 0.0f && mx - prune > cutoff) cutoff = mx - prune;
      double sum = 0.0;
      for (int m = s; m < e; m++) {
        const float f = row[m];
        if (f >= cutoff) sum += static_cast<double>(expf(f - mx));
      }
      out[static_cast<size_t>(t) * out_stride + j] =
          static_cast<float>(static_cast<double>(mx) + log(sum));
    }
  }
}

// Same, the row of per-Gaussian log-likelihoods staged in LDS first (read once, coalesced;
// above, every lane walks its own pdf's Gaussians with 20-byte strides).
constexpr int kLseLdsFloats = 12288;
__global__ void __launch_bounds__(kThreads)
GmmPdfLseRowKernel(const float *__restrict__ ll, int T, int ll_stride, int num_mix,
                   const int32_t *__restrict__ pdf_offsets, int num_pdfs, float prune,
                   float min_log_diff, float *__restrict__ out, int out_stride) {
  __shared__ float row[kLseLdsFloats];
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    const float *src = ll + static_cast<size_t>(t) * ll_stride;
    for (int m = threadIdx.x; m < num_mix; m += kThreads) row[m] = src[m];
    __syncthreads();
    for (int j = threadIdx.x; j < num_pdfs; j += kThreads) {
      const int s = pdf_offsets[j], e = pdf_offsets[j + 1];
      float mx = -INFINITY;
      for (int m = s; m < e; m++) mx = fmaxf(mx, row[m]);
      float cutoff = mx + min_log_diff;
      if (prune > 0.0f && mx - prune > cutoff) cutoff = mx - prune;
      double sum = 0.0;
      for (int m = s; m < e; m++) {
        const float f = row[m];
        if (f >= cutoff) sum += static_cast<double>(expf(f - mx));
      }
      out[static_cast<size_t>(t) * out_stride + j] = static_cast<float>(static_cast<double>(mx) + log(sum));
    }
    __syncthreads();
  }
}

template <int DP>
int LaunchLoglikes(const float *data, KhMatrixDim dd, const float *g,
                   const float *mi, const float *iv, int M, float *out,
                   int out_stride) {
  const int frame_blocks = DivUp(dd.rows, 64 * (kThreads / 64));
  // split the Gaussians over blockIdx.y until the chip is filled
  int my = 1;
  const int tiles_m = DivUp(M, kTileM);
  while (frame_blocks * my < NumCUs() * 4 && my < tiles_m) my *= 2;
  if (my > tiles_m) my = tiles_m;
  const int m_per_block = DivUp(tiles_m, my) * kTileM;
  my = DivUp(M, m_per_block);
  hipLaunchKernelGGL(GmmLoglikesKernel<DP>, dim3(frame_blocks, my), dim3(kThreads),
                     0, Stream(), data, dd.rows, dd.cols, dd.stride, g, mi, iv, M,
                     out, out_stride, m_per_block);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

}  // namespace

namespace {
// data_sq = data; data_sq.ApplyPow(2.0) (diag-gmm.cc:552-553), packed to a stride of 4 floats
__global__ void SquareKernel(float *__restrict__ y, int y_stride, const float *__restrict__ x, int x_stride, int rows, int cols) {
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
      const float v = x[static_cast<size_t>(r) * x_stride + c];
      y[static_cast<size_t>(r) * y_stride + c] = v * v;
    }
}
}  // namespace

extern "C" {

// HOST: DiagGmm::ComputeGconsts gmm/diag-gmm.cc:114-152
int kh_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                           const float *inv_vars, int num_mix, int dim,
                           float *gconsts) {
  KH_CHECK_ARG(weights && means_invvars && inv_vars && gconsts && num_mix > 0 && dim > 0);
  const double kLog2Pi = 1.8378770664093454835606594728112;
  const float offset = -0.5 * kLog2Pi * dim;
  int num_bad = 0;
  for (int mix = 0; mix < num_mix; mix++) {
    KH_CHECK_ARG(weights[mix] >= 0);  // KALDI_ASSERT :125
    float gc = logf(weights[mix]) + offset;
    for (int d = 0; d < dim; d++) {
      const float ivv = inv_vars[static_cast<size_t>(mix) * dim + d];
      const float miv = means_invvars[static_cast<size_t>(mix) * dim + d];
      gc += 0.5 * logf(ivv) - 0.5 * miv * miv / ivv;
    }
    if (std::isnan(gc)) {
      SetError("At component %d, not a number in gconst computation", mix);
      return KH_EINVAL;
    }
    if (std::isinf(gc)) {
      num_bad++;
      if (gc > 0) gc = -gc;
    }
    gconsts[mix] = gc;
  }
  return num_bad;
}

int kh_diag_gmm_loglikes(const float *data, KhMatrixDim dd, const float *gconsts,
                         const float *means_invvars, const float *inv_vars,
                         int num_mix, float *loglikes, int ll_stride) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(data && gconsts && means_invvars && inv_vars && loglikes);
  KH_CHECK_ARG(dd.rows > 0 && dd.cols > 0 && dd.stride >= dd.cols && num_mix > 0 &&
               ll_stride >= num_mix);  // KALDI_ASSERT(data.NumRows() != 0) :548
  const int D = dd.cols;
  // Large problems: the reference's own formulation (diag-gmm.cc:546-562) as two MFMA
  // GEMMs, loglikes = gconsts + data * means_invvars^T, then += -0.5 * data^2 * inv_vars^T
  // (same k-ordered accumulation as the register kernel below, ~3x its rate).
  if (static_cast<int64_t>(dd.rows) * num_mix >= (1 << 22) && !getenv("KH_GMM_NO_GEMM")) {
    const int sq_stride = (D + 3) & ~3;
    float *sq = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(dd.rows) * sq_stride));
    if (!sq) return KH_ENOMEM;
    hipLaunchKernelGGL(SquareKernel, dim3(1, std::min(dd.rows, NumCUs() * 32)), dim3(64), 0, Stream(), sq, sq_stride, data,
                       dd.stride, dd.rows, D);
    const KhMatrixDim dsq{dd.rows, D, sq_stride}, dpar{num_mix, D, D}, dll{dd.rows, num_mix, ll_stride};
    rc = kh_affine(data, dd, means_invvars, dpar, gconsts, loglikes, dll);
    if (!rc) rc = kh_add_mat_mat(-0.5f, sq, dsq, 0, inv_vars, dpar, 1, 1.0f, loglikes, dll);
    hipError_t e = hipStreamSynchronize(Stream());
    PoolFree(sq);
    if (!rc && e != hipSuccess) { SetError("kh_diag_gmm_loglikes: %s", hipGetErrorString(e)); rc = KH_EDEVICE; }
    return rc;
  }
  if (D <= 16) return LaunchLoglikes<16>(data, dd, gconsts, means_invvars, inv_vars, num_mix, loglikes, ll_stride);
  if (D <= 32) return LaunchLoglikes<32>(data, dd, gconsts, means_invvars, inv_vars, num_mix, loglikes, ll_stride);
  if (D <= 40) return LaunchLoglikes<40>(data, dd, gconsts, means_invvars, inv_vars, num_mix, loglikes, ll_stride);
  if (D <= 64) return LaunchLoglikes<64>(data, dd, gconsts, means_invvars, inv_vars, num_mix, loglikes, ll_stride);
  if (D <= 96) return LaunchLoglikes<96>(data, dd, gconsts, means_invvars, inv_vars, num_mix, loglikes, ll_stride);
  SetError("kh_diag_gmm_loglikes: feature dimension %d > 96 not supported", D);
  return KH_EINVAL;
}

int kh_am_gmm_loglikes(const float *data, KhMatrixDim dd, const float *gconsts,
                       const float *means_invvars, const float *inv_vars,
                       const int32_t *pdf_offsets, int num_pdfs, int num_mix,
                       float log_sum_exp_prune, float *out, int out_stride) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(pdf_offsets && out && num_pdfs > 0 && out_stride >= num_pdfs);
  const int ll_stride = (num_mix + 3) & ~3;
  // Bound the scratch: process frames in slabs of at most ~1 GiB of T x M.
  const int64_t slab_rows = std::max<int64_t>(64, (int64_t(1) << 28) / ll_stride);
  const int T = dd.rows;
  float *scratch = static_cast<float *>(
      PoolMalloc(sizeof(float) * std::min<int64_t>(T, slab_rows) * ll_stride));
  if (!scratch) return KH_ENOMEM;
  const float min_log_diff = logf(FLT_EPSILON);  // kMinLogDiffFloat kaldi-math.h:121
  for (int64_t r0 = 0; r0 < T; r0 += slab_rows) {
    const int rows = static_cast<int>(std::min<int64_t>(slab_rows, T - r0));
    KhMatrixDim ds{rows, dd.cols, dd.stride};
    rc = kh_diag_gmm_loglikes(data + r0 * dd.stride, ds, gconsts, means_invvars,
                              inv_vars, num_mix, scratch, ll_stride);
    if (rc) break;
    int gx = DivUp(num_pdfs, kThreads);
    int gy = rows;
    const int cap = NumCUs() * 16 / gx;
    if (gy > cap) gy = cap > 0 ? cap : 1;
    if (num_mix <= kLseLdsFloats && num_mix >= 1024)
      hipLaunchKernelGGL(GmmPdfLseRowKernel, dim3(std::min(rows, NumCUs() * 12)), dim3(kThreads), 0, Stream(),
                         scratch, rows, ll_stride, num_mix, pdf_offsets, num_pdfs, log_sum_exp_prune, min_log_diff,
                         out + r0 * out_stride, out_stride);
    else
      hipLaunchKernelGGL(GmmPdfLseKernel, dim3(gx, gy), dim3(kThreads), 0, Stream(),
                         scratch, rows, ll_stride, pdf_offsets, num_pdfs,
                         log_sum_exp_prune, min_log_diff, out + r0 * out_stride,
                         out_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      SetError("GmmPdfLseKernel launch failed: %s", hipGetErrorString(e));
      rc = KH_EDEVICE;
      break;
    }
  }
  KH_HIP(hipStreamSynchronize(Stream()));  // scratch returns to the pool
  PoolFree(scratch);
  return rc;
}

}  // extern "C"
