// ref_driver.cc — TEST INFRASTRUCTURE (oracle/_ref build only).
//
// Thin extern "C" driver over the REFERENCE's own classes, compiled against the
// reference headers/sources where they lie under /root/reference/src (see
// oracle/Makefile).  It contains no algorithm: each ref_* function copies its
// dense arguments into the reference's Matrix/CuMatrix (HAVE_CUDA undefined ->
// CPU branch), calls the reference function named in its comment, and copies
// the result back.  Signatures mirror the ko_* functions of kaldi_oracle.h so
// the same Python wrappers can drive both.
//
// Used by tests/golden/make_golden.py (in the build container) to generate the
// golden vectors, by tests/test_oracle_vs_ref.py, and (optionally) as bench.py's
// cpu_baseline kind "reference" for the nnet2 forward / GMM part.

#include <cstring>
#include <string>
#include <vector>

#include "cudamatrix/cu-math.h"
#include "cudamatrix/cu-matrix-lib.h"
#include "feat/feature-functions.h"
#include "feat/feature-mfcc.h"
#include "feat/wave-reader.h"
#include "feat/online-feature.h"
#include "matrix/optimization.h"
#include "gmm/am-diag-gmm.h"
#include "gmm/diag-gmm.h"
#include "hmm/hmm-topology.h"
#include "matrix/matrix-lib.h"
#include "nnet2/nnet-component.h"
#include "nnet2/am-nnet.h"
#include "nnet2/nnet-nnet.h"
#include "transform/cmvn.h"
#include "util/kaldi-io.h"
#include "util/table-types.h"

#include "kaldi_oracle.h"  // KoComponent

using namespace kaldi;
using namespace kaldi::nnet2;

namespace {

Matrix<BaseFloat> In(const float *p, int rows, int cols, int stride) {
  Matrix<BaseFloat> m(rows, cols, kUndefined);
  for (int r = 0; r < rows; r++)
    memcpy(m.RowData(r), p + static_cast<size_t>(r) * stride,
           sizeof(float) * cols);
  return m;
}
void Out(const MatrixBase<BaseFloat> &m, float *p, int stride) {
  for (int r = 0; r < m.NumRows(); r++)
    memcpy(p + static_cast<size_t>(r) * stride, m.RowData(r),
           sizeof(float) * m.NumCols());
}
Vector<BaseFloat> InV(const float *p, int dim) {
  Vector<BaseFloat> v(dim, kUndefined);
  memcpy(v.Data(), p, sizeof(float) * dim);
  return v;
}

}  // namespace

extern "C" {

// MatrixBase::AddMatMat matrix/kaldi-matrix.cc:160-175 via CuMatrix CPU branch
// cudamatrix/cu-matrix.cc:977-981.
void ref_add_mat_mat(float alpha, const float *A, int a_rows, int a_cols,
                     int a_stride, int transA, const float *B, int b_rows,
                     int b_cols, int b_stride, int transB, float beta, float *C,
                     int c_rows, int c_cols, int c_stride) {
  CuMatrix<BaseFloat> a(In(A, a_rows, a_cols, a_stride)),
      b(In(B, b_rows, b_cols, b_stride)), c(In(C, c_rows, c_cols, c_stride));
  c.AddMatMat(alpha, a, transA ? kTrans : kNoTrans, b,
              transB ? kTrans : kNoTrans, beta);
  Out(c.Mat(), C, c_stride);
}

// cu-matrix.cc:1251-1271
void ref_softmax_per_row(const float *src, int rows, int cols, int src_stride,
                         float *dst, int dst_stride) {
  CuMatrix<BaseFloat> s(In(src, rows, cols, src_stride)), d(rows, cols);
  d.ApplySoftMaxPerRow(s);
  Out(d.Mat(), dst, dst_stride);
}

// cu-matrix.cc:1274-1295
void ref_log_softmax_per_row(const float *src, int rows, int cols,
                             int src_stride, float *dst, int dst_stride) {
  CuMatrix<BaseFloat> s(In(src, rows, cols, src_stride)), d(rows, cols);
  d.ApplyLogSoftMaxPerRow(s);
  Out(d.Mat(), dst, dst_stride);
}

// cu-matrix.cc:1965-1990
void ref_copy_rows(float *dst, int rows, int cols, int dst_stride,
                   const float *src, int src_rows, int src_stride,
                   const int32_t *indices) {
  CuMatrix<BaseFloat> s(In(src, src_rows, cols, src_stride)),
      d(In(dst, rows, cols, dst_stride));
  std::vector<int32> idx(indices, indices + rows);
  d.CopyRows(s, idx);
  Out(d.Mat(), dst, dst_stride);
}

// cudamatrix/cu-math.cc:130-165
void ref_splice(const float *src, int rows, int cols, int src_stride,
                const int32_t *frame_offsets, int n_offsets, float *tgt,
                int tgt_stride) {
  CuMatrix<BaseFloat> s(In(src, rows, cols, src_stride)),
      t(rows, cols * n_offsets);
  std::vector<int32> off(frame_offsets, frame_offsets + n_offsets);
  CuArray<int32> cu_off(off);
  cu::Splice(s, cu_off, &t);
  Out(t.Mat(), tgt, tgt_stride);
}

// cu-matrix.cc:1147-1164
void ref_group_pnorm(const float *src, int rows, int src_cols, int src_stride,
                     float power, float *dst, int dst_cols, int dst_stride) {
  CuMatrix<BaseFloat> s(In(src, rows, src_cols, src_stride)), d(rows, dst_cols);
  d.GroupPnorm(s, power);
  Out(d.Mat(), dst, dst_stride);
}

// NormalizeComponent::Propagate nnet2/nnet-component.cc:576-588
void ref_normalize(const float *src, int rows, int cols, int src_stride,
                   float *dst, int dst_stride) {
  CuMatrix<BaseFloat> s(In(src, rows, cols, src_stride)), d(rows, cols);
  NormalizeComponent comp(cols);
  ChunkInfo info(cols, 1, 0, rows - 1);
  comp.Propagate(info, info, s, &d);
  Out(d.Mat(), dst, dst_stride);
}

// cu-vector.cc:517-580 AddDiagMat2
void ref_add_diag_mat2(float alpha, const float *M, int rows, int cols,
                       int stride, float beta, float *v) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  CuVector<BaseFloat> vec(InV(v, rows));
  vec.AddDiagMat2(alpha, m, kNoTrans, beta);
  Vector<BaseFloat> host(rows);
  vec.CopyToVec(&host);
  memcpy(v, host.Data(), sizeof(float) * rows);
}

void ref_mul_rows_vec(float *M, int rows, int cols, int stride, const float *s) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  CuVector<BaseFloat> v(InV(s, rows));
  m.MulRowsVec(v);
  Out(m.Mat(), M, stride);
}

void ref_mul_cols_vec(float *M, int rows, int cols, int stride, const float *s) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  CuVector<BaseFloat> v(InV(s, cols));
  m.MulColsVec(v);
  Out(m.Mat(), M, stride);
}

void ref_copy_rows_from_vec(float *M, int rows, int cols, int stride,
                            const float *v) {
  CuMatrix<BaseFloat> m(rows, cols);
  CuVector<BaseFloat> vec(InV(v, cols));
  m.CopyRowsFromVec(vec);
  Out(m.Mat(), M, stride);
}

void ref_add_vec_to_rows(float alpha, const float *v, float beta, float *M,
                         int rows, int cols, int stride) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  CuVector<BaseFloat> vec(InV(v, cols));
  m.AddVecToRows(alpha, vec, beta);
  Out(m.Mat(), M, stride);
}

void ref_apply_floor(float *M, int rows, int cols, int stride, float f) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  m.ApplyFloor(f);
  Out(m.Mat(), M, stride);
}
void ref_apply_log(float *M, int rows, int cols, int stride) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  m.ApplyLog();
  Out(m.Mat(), M, stride);
}
void ref_apply_exp(float *M, int rows, int cols, int stride) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  m.ApplyExp();
  Out(m.Mat(), M, stride);
}
void ref_apply_pow(float *M, int rows, int cols, int stride, float power) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  m.ApplyPow(power);
  Out(m.Mat(), M, stride);
}
void ref_scale(float *M, int rows, int cols, int stride, float alpha) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  m.Scale(alpha);
  Out(m.Mat(), M, stride);
}

// cu-matrix.cc:1994-2028
void ref_sum_column_ranges(float *dst, int rows, int dst_cols, int dst_stride,
                           const float *src, int src_cols, int src_stride,
                           const int32_t *ranges) {
  CuMatrix<BaseFloat> s(In(src, rows, src_cols, src_stride)), d(rows, dst_cols);
  std::vector<Int32Pair> idx(dst_cols);
  for (int i = 0; i < dst_cols; i++) {
    idx[i].first = ranges[2 * i];
    idx[i].second = ranges[2 * i + 1];
  }
  CuArray<Int32Pair> cu_idx(idx);
  d.SumColumnRanges(s, cu_idx);
  Out(d.Mat(), dst, dst_stride);
}

// cu-matrix.cc:2327-...
void ref_matrix_lookup(const float *M, int rows, int cols, int stride,
                       const int32_t *row_col_pairs, int n, float *out) {
  CuMatrix<BaseFloat> m(In(M, rows, cols, stride));
  std::vector<Int32Pair> idx(n);
  for (int i = 0; i < n; i++) {
    idx[i].first = row_col_pairs[2 * i];
    idx[i].second = row_col_pairs[2 * i + 1];
  }
  std::vector<BaseFloat> o;
  m.Lookup(idx, &o);
  memcpy(out, o.data(), sizeof(float) * n);
}

// ---- nnet2 -------------------------------------------------------------------
static Nnet *BuildNnet(const KoComponent *comps, int n_comps) {
  std::vector<Component *> cs;
  for (int i = 0; i < n_comps; i++) {
    const KoComponent &k = comps[i];
    switch (k.type) {
      case KO_SPLICE: {
        SpliceComponent *c = new SpliceComponent();
        c->Init(k.input_dim,
                std::vector<int32>(k.context, k.context + k.n_context),
                k.const_dim);
        cs.push_back(c);
        break;
      }
      case KO_FIXED_AFFINE: {
        Matrix<BaseFloat> m(k.output_dim, k.input_dim + 1);
        for (int r = 0; r < k.output_dim; r++) {
          memcpy(m.RowData(r), k.linear + static_cast<size_t>(r) * k.input_dim,
                 sizeof(float) * k.input_dim);
          m(r, k.input_dim) = k.bias[r];
        }
        FixedAffineComponent *c = new FixedAffineComponent();
        c->Init(CuMatrix<BaseFloat>(m));
        cs.push_back(c);
        break;
      }
      case KO_AFFINE: {
        CuMatrix<BaseFloat> lin(In(k.linear, k.output_dim, k.input_dim,
                                   k.input_dim));
        CuVector<BaseFloat> b(InV(k.bias, k.output_dim));
        cs.push_back(new AffineComponent(lin, b, 0.001));
        break;
      }
      case KO_PNORM:
        cs.push_back(new PnormComponent(k.input_dim, k.output_dim, k.p));
        break;
      case KO_NORMALIZE:
        cs.push_back(new NormalizeComponent(k.input_dim));
        break;
      case KO_SOFTMAX:
        cs.push_back(new SoftmaxComponent(k.input_dim));
        break;
      case KO_SUM_GROUP: {
        SumGroupComponent *c = new SumGroupComponent();
        c->Init(std::vector<int32>(k.sizes, k.sizes + k.n_sizes));
        cs.push_back(c);
        break;
      }
      case KO_FIXED_SCALE: {
        FixedScaleComponent *c = new FixedScaleComponent();
        c->Init(CuVector<BaseFloat>(InV(k.bias, k.input_dim)));
        cs.push_back(c);
        break;
      }
      case KO_FIXED_BIAS: {
        FixedBiasComponent *c = new FixedBiasComponent();
        c->Init(CuVector<BaseFloat>(InV(k.bias, k.input_dim)));
        cs.push_back(c);
        break;
      }
      default:
        return NULL;
    }
  }
  Nnet *nnet = new Nnet();
  nnet->Init(&cs);
  return nnet;
}

// ---- the <double> instantiation (cu-matrix.cc:2415-2418; the reference's own tests run CudaMatrixUnitTest<double>()):
// ONE entry point over the reference's CuMatrix<double> / CuVector<double> methods, dense operands (stride = cols).
// op: 0 AddMatMat(alpha, A, tA, B, tB, beta) into C | 1 ApplySoftMaxPerRow(A) | 2 ApplyLogSoftMaxPerRow(A) |
// 3 CopyRows(A, idx) | 4 cu::Splice(A, idx) | 5 GroupPnorm(A, alpha) | 6 C(vector, c_rows) .AddDiagMat2(alpha, A, kNoTrans, beta) |
// 7 C.MulRowsVec(B) | 8 C.MulColsVec(B) | 9 C.CopyRowsFromVec(B) | 10 C.AddVecToRows(alpha, B, beta) | 11 C.ApplyFloor(alpha) |
// 12 C.ApplyLog | 13 C.ApplyExp | 14 C.ApplyPow(alpha) | 15 C.Scale(alpha) | 16 C.SumColumnRanges(A, idx pairs) |
// 17 A.Lookup(idx pairs) -> C[n_idx / 2].  B as a vector has b_cols elements.  C holds the in/out matrix.
int ref_op_d(int op, double alpha, double beta, const double *A, int a_rows, int a_cols, int transA, const double *B,
             int b_rows, int b_cols, int transB, const int32_t *idx, int n_idx, double *C, int c_rows, int c_cols) {
  auto in = [](const double *p, int rows, int cols) {
    Matrix<double> m(rows, cols, kUndefined);
    for (int r = 0; r < rows; r++) memcpy(m.RowData(r), p + static_cast<size_t>(r) * cols, sizeof(double) * cols);
    return m;
  };
  auto out = [](const MatrixBase<double> &m, double *p) {
    for (int r = 0; r < m.NumRows(); r++) memcpy(p + static_cast<size_t>(r) * m.NumCols(), m.RowData(r), sizeof(double) * m.NumCols());
  };
  auto vec = [](const double *p, int dim) {
    Vector<double> v(dim, kUndefined);
    memcpy(v.Data(), p, sizeof(double) * dim);
    return v;
  };
  auto pairs = [&]() {
    std::vector<Int32Pair> v(n_idx / 2);
    for (int i = 0; i < n_idx / 2; i++) { v[i].first = idx[2 * i]; v[i].second = idx[2 * i + 1]; }
    return v;
  };
  CuMatrix<double> a, b, c;
  if (A) a = CuMatrix<double>(in(A, a_rows, a_cols));
  if (C && op != 6 && op != 17) c = CuMatrix<double>(in(C, c_rows, c_cols));
  switch (op) {
    case 0:
      b = CuMatrix<double>(in(B, b_rows, b_cols));
      c.AddMatMat(alpha, a, transA ? kTrans : kNoTrans, b, transB ? kTrans : kNoTrans, beta);
      break;
    case 1: c.ApplySoftMaxPerRow(a); break;
    case 2: c.ApplyLogSoftMaxPerRow(a); break;
    case 3: c.CopyRows(a, std::vector<int32>(idx, idx + n_idx)); break;
    case 4: {
      CuArray<int32> off(std::vector<int32>(idx, idx + n_idx));
      cu::Splice(a, off, &c);
      break;
    }
    case 5: c.GroupPnorm(a, alpha); break;
    case 6: {
      CuVector<double> v(vec(C, c_rows));
      v.AddDiagMat2(alpha, a, kNoTrans, beta);
      Vector<double> h(c_rows);
      v.CopyToVec(&h);
      memcpy(C, h.Data(), sizeof(double) * c_rows);
      return 0;
    }
    case 7: c.MulRowsVec(CuVector<double>(vec(B, b_cols))); break;
    case 8: c.MulColsVec(CuVector<double>(vec(B, b_cols))); break;
    case 9: c.CopyRowsFromVec(CuVector<double>(vec(B, b_cols))); break;
    case 10: c.AddVecToRows(alpha, CuVector<double>(vec(B, b_cols)), beta); break;
    case 11: c.ApplyFloor(alpha); break;
    case 12: c.ApplyLog(); break;
    case 13: c.ApplyExp(); break;
    case 14: c.ApplyPow(alpha); break;
    case 15: c.Scale(alpha); break;
    case 16: {
      CuArray<Int32Pair> cu_idx(pairs());
      c.SumColumnRanges(a, cu_idx);
      break;
    }
    case 17: {
      std::vector<double> o;
      a.Lookup(pairs(), &o);
      memcpy(C, o.data(), sizeof(double) * o.size());
      return 0;
    }
    default: return -1;
  }
  out(c.Mat(), C);
  return 0;
}


int ref_nnet_left_context(const KoComponent *comps, int n) {
  Nnet *nnet = BuildNnet(comps, n);
  int ans = nnet->LeftContext();
  delete nnet;
  return ans;
}
int ref_nnet_right_context(const KoComponent *comps, int n) {
  Nnet *nnet = BuildNnet(comps, n);
  int ans = nnet->RightContext();
  delete nnet;
  return ans;
}

// The reference's NnetComputer (nnet2/nnet-compute.cc:63-108) cannot be
// compiled here (it includes hmm/posterior.h -> fst/fst-decl.h, OpenFst absent),
// so this driver performs its two steps with the reference's own pieces:
// the edge-frame padding by Matrix row copies (:75-89), Nnet::ComputeChunkInfo
// (reference code, nnet-nnet.cc:65-112) and a loop of the reference's
// Component::Propagate (nnet-component.h:197-215) — :94-108.
int ref_nnet_forward(const KoComponent *comps, int n_comps, const float *feats,
                     int T, int feat_stride, int pad_input, float *out,
                     int out_stride) {
  Nnet *nnet = BuildNnet(comps, n_comps);
  if (!nnet) return -8;
  int dim = nnet->InputDim();
  int left = pad_input ? nnet->LeftContext() : 0,
      right = pad_input ? nnet->RightContext() : 0;
  int num_rows = left + T + right;
  std::vector<ChunkInfo> chunk_info;
  nnet->ComputeChunkInfo(num_rows, 1, &chunk_info);
  CuMatrix<BaseFloat> input_feats(In(feats, T, dim, feat_stride));
  CuMatrix<BaseFloat> input(num_rows, dim);
  input.Range(left, T, 0, dim).CopyFromMat(input_feats);
  for (int i = 0; i < left; i++) input.Row(i).CopyFromVec(input_feats.Row(0));
  for (int i = 0; i < right; i++)
    input.Row(num_rows - i - 1).CopyFromVec(input_feats.Row(T - 1));
  CuMatrix<BaseFloat> cur(input), next;
  for (int c = 0; c < nnet->NumComponents(); c++) {
    nnet->GetComponent(c).Propagate(chunk_info[c], chunk_info[c + 1], cur, &next);
    cur.Swap(&next);
    next.Resize(0, 0);
  }
  int rows = cur.NumRows();
  Out(cur.Mat(), out, out_stride);
  delete nnet;
  return rows;
}

// nnet2/decodable-am-nnet.h:39-73 (header cannot be included: it pulls
// hmm/transition-model.h -> OpenFst); the five CuMatrix calls of its ctor are
// issued here on the reference's CuMatrix.
int ref_decodable_am_nnet(const KoComponent *comps, int n_comps,
                          const float *priors, float prob_scale,
                          const float *feats, int T, int feat_stride,
                          float *log_probs, int out_stride) {
  int n = comps[n_comps - 1].output_dim;
  Matrix<BaseFloat> tmp(T, n);
  int rows = ref_nnet_forward(comps, n_comps, feats, T, feat_stride, 1,
                              tmp.Data(), tmp.Stride());
  if (rows < 0) return rows;
  CuMatrix<BaseFloat> lp(tmp);
  lp.ApplyFloor(1.0e-20);
  lp.ApplyLog();
  CuVector<BaseFloat> priors_v(InV(priors, n));
  priors_v.ApplyLog();
  lp.AddVecToRows(-1.0, priors_v);
  lp.Scale(prob_scale);
  Out(lp.Mat(), log_probs, out_stride);
  return rows;
}

// ---- DiagGmm -------------------------------------------------------------------
// The model is given in its natural parameters (weights, means, vars); the
// reference derives inv_vars / means_invvars / gconsts itself
// (DiagGmm::SetInvVarsAndMeans + ComputeGconsts, gmm/diag-gmm.cc:114-152) and
// they are exported so the restatement and the HIP kernels consume the same
// stored parameters the reference would read from a model file.
static void FillGmm(DiagGmm *gmm, const float *weights, const float *means,
                    const float *vars, int num_mix, int dim) {
  gmm->Resize(num_mix, dim);
  gmm->SetWeights(InV(weights, num_mix));
  Matrix<BaseFloat> inv_vars(In(vars, num_mix, dim, dim));
  inv_vars.InvertElements();
  gmm->SetInvVarsAndMeans(inv_vars, In(means, num_mix, dim, dim));
}

int ref_diag_gmm_build(const float *weights, const float *means,
                       const float *vars, int num_mix, int dim, float *gconsts,
                       float *means_invvars, float *inv_vars) {
  DiagGmm gmm;
  FillGmm(&gmm, weights, means, vars, num_mix, dim);
  int bad = gmm.ComputeGconsts();
  memcpy(gconsts, gmm.gconsts().Data(), sizeof(float) * num_mix);
  Out(gmm.means_invvars(), means_invvars, dim);
  Out(gmm.inv_vars(), inv_vars, dim);
  return bad;
}

// DiagGmm::LogLikelihoods(const MatrixBase&, Matrix*) gmm/diag-gmm.cc:546-562
void ref_diag_gmm_loglikes(const float *weights, const float *means,
                           const float *vars, int num_mix, int dim,
                           const float *data, int T, int data_stride,
                           float *loglikes, int ll_stride) {
  DiagGmm gmm;
  FillGmm(&gmm, weights, means, vars, num_mix, dim);
  gmm.ComputeGconsts();
  Matrix<BaseFloat> d(In(data, T, dim, data_stride)), ll;
  gmm.LogLikelihoods(d, &ll);
  Out(ll, loglikes, ll_stride);
}

// Per-frame DiagGmm::LogLikelihood(VectorBase) diag-gmm.cc:517-526 (vector
// LogLikelihoods :528-543 + LogSumExp()).
void ref_diag_gmm_loglike_per_frame(const float *weights, const float *means,
                                    const float *vars, int num_mix, int dim,
                                    const float *data, int T, int data_stride,
                                    float *out) {
  DiagGmm gmm;
  FillGmm(&gmm, weights, means, vars, num_mix, dim);
  gmm.ComputeGconsts();
  Matrix<BaseFloat> d(In(data, T, dim, data_stride));
  for (int t = 0; t < T; t++) out[t] = gmm.LogLikelihood(d.Row(t));
}

// VectorBase::LogSumExp matrix/kaldi-vector.cc:745-763
float ref_log_sum_exp(const float *v, int dim, float prune) {
  Vector<BaseFloat> vec(InV(v, dim));
  return vec.LogSumExp(prune);
}


// ---- feature front-end: Mfcc::Compute feat/feature-mfcc.cc:96-184,
// ComputeDeltas feat/feature-functions.cc:361-372, AccCmvnStats / ApplyCmvn
// transform/cmvn.cc:49-113.  dither is forced to 0 (the only random step).
int ref_mfcc_compute_opts(const float *wave, int n_samples, float samp_freq, float frame_length_ms, float frame_shift_ms,
                          float preemph_coeff, int remove_dc_offset, const char *window_type, int snip_edges, int use_energy,
                          int raw_energy, float energy_floor, int htk_compat, int num_bins, float low_freq, float high_freq,
                          int num_ceps, float cepstral_lifter, float *out, int out_stride, int max_rows) {
  MfccOptions opts;
  opts.frame_opts.samp_freq = samp_freq;
  opts.frame_opts.frame_length_ms = frame_length_ms;
  opts.frame_opts.frame_shift_ms = frame_shift_ms;
  opts.frame_opts.dither = 0.0;
  opts.frame_opts.preemph_coeff = preemph_coeff;
  opts.frame_opts.remove_dc_offset = remove_dc_offset != 0;
  opts.frame_opts.window_type = window_type;
  opts.frame_opts.snip_edges = snip_edges != 0;
  opts.mel_opts.num_bins = num_bins;
  opts.mel_opts.low_freq = low_freq;
  opts.mel_opts.high_freq = high_freq;
  opts.num_ceps = num_ceps;
  opts.cepstral_lifter = cepstral_lifter;
  opts.use_energy = use_energy != 0;
  opts.raw_energy = raw_energy != 0;
  opts.energy_floor = energy_floor;
  opts.htk_compat = htk_compat != 0;
  Mfcc mfcc(opts);
  Vector<BaseFloat> w(n_samples);
  memcpy(w.Data(), wave, sizeof(float) * n_samples);
  Matrix<BaseFloat> feats;
  mfcc.Compute(w, 1.0, &feats, NULL);
  if (feats.NumRows() > max_rows) return -1;
  if (feats.NumRows() > 0) Out(feats, out, out_stride);
  return feats.NumRows();
}

int ref_mfcc_compute(const float *wave, int n_samples, float samp_freq, float frame_length_ms, float frame_shift_ms,
                     float preemph_coeff, int remove_dc_offset, const char *window_type, int snip_edges,
                     int num_bins, float low_freq, float high_freq, int num_ceps, float cepstral_lifter,
                     float *out, int out_stride, int max_rows) {
  return ref_mfcc_compute_opts(wave, n_samples, samp_freq, frame_length_ms, frame_shift_ms, preemph_coeff, remove_dc_offset,
                               window_type, snip_edges, 0, 1, 0.0f, 0, num_bins, low_freq, high_freq, num_ceps,
                               cepstral_lifter, out, out_stride, max_rows);
}

void ref_compute_deltas(const float *in, int rows, int cols, int in_stride, int order, int window, float *out,
                        int out_stride) {
  DeltaFeaturesOptions opts(order, window);
  Matrix<BaseFloat> input = In(in, rows, cols, in_stride), output;
  ComputeDeltas(opts, input, &output);
  Out(output, out, out_stride);
}

// stats: [2 x (cols + 1)] doubles, row-major, accumulated over the rows of feats
void ref_acc_cmvn_stats(const float *feats, int rows, int cols, int stride, double *stats) {
  Matrix<BaseFloat> f = In(feats, rows, cols, stride);
  Matrix<double> st(2, cols + 1);
  for (int r = 0; r < 2; r++)
    for (int c = 0; c <= cols; c++) st(r, c) = stats[r * (cols + 1) + c];
  AccCmvnStats(f, NULL, &st);
  for (int r = 0; r < 2; r++)
    for (int c = 0; c <= cols; c++) stats[r * (cols + 1) + c] = st(r, c);
}

void ref_apply_cmvn(const double *stats, int var_norm, float *feats, int rows, int cols, int stride) {
  Matrix<double> st(2, cols + 1);
  for (int r = 0; r < 2; r++)
    for (int c = 0; c <= cols; c++) st(r, c) = stats[r * (cols + 1) + c];
  Matrix<BaseFloat> f = In(feats, rows, cols, stride);
  ApplyCmvn(st, var_norm != 0, &f);
  Out(f, feats, stride);
}


// ---- object / table / model I/O by the reference's own Write / Read -------------
// (fixtures of tests/golden/kaldi_io/, tests/test_kaldi_io.py)

// Matrix<float|double>::Write matrix/kaldi-matrix.cc:1155-1193 through Output (util/kaldi-io.cc)
int ref_write_matrix(const char *path, const float *data, int rows, int cols, int binary, int as_double) {
  try {
    Output ko(path, binary != 0);
    Matrix<BaseFloat> m(In(data, rows, cols, cols));
    if (as_double) { Matrix<double> d(m); d.Write(ko.Stream(), binary != 0); }
    else m.Write(ko.Stream(), binary != 0);
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
int ref_write_vector(const char *path, const float *data, int dim, int binary) {
  try {
    Output ko(path, binary != 0);
    InV(data, dim).Write(ko.Stream(), binary != 0);
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
// CompressedMatrix::Write matrix/compressed-matrix.cc:404-435 (format 1 "CM" or, for <= 8 rows, 2 "CM2")
int ref_write_compressed_matrix(const char *path, const float *data, int rows, int cols) {
  try {
    Output ko(path, true);
    CompressedMatrix cm(In(data, rows, cols, cols));
    cm.Write(ko.Stream(), true);
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
// and what the reference itself decompresses it to (CopyToMat :476-520)
int ref_compress_roundtrip(const float *data, int rows, int cols, float *out) {
  CompressedMatrix cm(In(data, rows, cols, cols));
  Matrix<BaseFloat> m(rows, cols);
  cm.CopyToMat(&m);
  Out(m, out, cols);
  return 0;
}
// WriteIntegerVector base/io-funcs-inl.h:195-224
int ref_write_int_vector(const char *path, const int32_t *data, int n, int binary) {
  try {
    Output ko(path, binary != 0);
    std::vector<int32> v(data, data + n);
    WriteIntegerVector(ko.Stream(), binary != 0, v);
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
// TableWriter<KaldiObjectHolder<Matrix>> util/kaldi-table-inl.h: utterance u = rows
// [offsets[u], offsets[u+1]) of data, key "utt<u>"; compressed != 0: CompressedMatrixWriter
int ref_write_matrix_table(const char *wspecifier, const float *data, int cols, const int32_t *offsets, int n_utts,
                           int compressed) {
  try {
    BaseFloatMatrixWriter w;
    CompressedMatrixWriter cw;
    if (compressed) { if (!cw.Open(wspecifier)) return -1; }
    else if (!w.Open(wspecifier)) return -1;
    for (int u = 0; u < n_utts; u++) {
      char key[32];
      snprintf(key, sizeof(key), "utt%d", u);
      Matrix<BaseFloat> m(In(data + static_cast<size_t>(offsets[u]) * cols, offsets[u + 1] - offsets[u], cols, cols));
      if (compressed) cw.Write(key, CompressedMatrix(m));
      else w.Write(key, m);
    }
    return (compressed ? cw.Close() : w.Close()) ? 0 : -1;
  } catch (...) { return -1; }
}
// SequentialTableReader over any rspecifier: total rows / cols / checksum of what the REFERENCE reads
int ref_read_matrix_table(const char *rspecifier, int *n_utts, int *tot_rows, int *cols, double *sum,
                          char *keys, int keys_cap) {
  try {
    SequentialBaseFloatMatrixReader r(rspecifier);
    *n_utts = 0; *tot_rows = 0; *cols = 0; *sum = 0.0;
    std::string all;
    for (; !r.Done(); r.Next()) {
      const Matrix<BaseFloat> &m = r.Value();
      (*n_utts)++;
      *tot_rows += m.NumRows();
      *cols = m.NumCols();
      *sum += m.Sum();
      all += r.Key() + " ";
    }
    snprintf(keys, keys_cap, "%s", all.c_str());
    return 0;
  } catch (...) { return -1; }
}
// TableWriter<BasicVectorHolder<int32>> (alignments, word sequences) util/kaldi-holder-inl.h:191-224:
// vector u = data[offsets[u], offsets[u+1]), key "utt<u>"
int ref_write_int_vector_table(const char *wspecifier, const int32_t *data, const int32_t *offsets, int n_utts) {
  try {
    Int32VectorWriter w(wspecifier);
    for (int u = 0; u < n_utts; u++) {
      char key[32];
      snprintf(key, sizeof(key), "utt%d", u);
      w.Write(key, std::vector<int32>(data + offsets[u], data + offsets[u + 1]));
    }
    return w.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
int ref_read_int_vector_table(const char *rspecifier, int *n_utts, long long *sum) {
  try {
    SequentialInt32VectorReader r(rspecifier);
    *n_utts = 0; *sum = 0;
    for (; !r.Done(); r.Next()) {
      (*n_utts)++;
      for (size_t i = 0; i < r.Value().size(); i++) *sum += static_cast<long long>(r.Value()[i]) * (i + 1);
    }
    return 0;
  } catch (...) { return -1; }
}
// AmNnet::Write nnet2/am-nnet.cc:31-37 (Nnet::Write nnet-nnet.cc:160-173 + priors).  affine_kind: the class
// the KO_AFFINE layers are written as: 0 AffineComponent, 1 AffineComponentPreconditioned,
// 2 AffineComponentPreconditionedOnline (nnet-component.cc:1288,1520,1829).  with_header: Output's "\0B".
int ref_write_am_nnet(const char *path, const KoComponent *comps, int n_comps, const float *priors, int n_priors,
                      int binary, int affine_kind, int with_header) {
  try {
    Nnet *nnet = BuildNnet(comps, n_comps);
    if (!nnet) return -2;
    if (affine_kind != 0) {
      for (int c = 0; c < nnet->NumComponents(); c++) {
        AffineComponent *ac = dynamic_cast<AffineComponent *>(&nnet->GetComponent(c));
        if (!ac) continue;
        Component *repl;
        if (affine_kind == 2) {
          repl = new AffineComponentPreconditionedOnline(*ac, 20, 40, 2, 2000.0, 4.0);
        } else {
          AffineComponentPreconditioned *p = new AffineComponentPreconditioned();
          std::ostringstream args;
          args << "learning-rate=0.002 input-dim=" << ac->InputDim() << " output-dim=" << ac->OutputDim()
               << " alpha=3.5 max-change=7.0 param-stddev=0.1 bias-stddev=0.1";
          p->InitFromString(args.str());
          p->SetParams(Vector<BaseFloat>(ac->BiasParams()), Matrix<BaseFloat>(ac->LinearParams()));
          repl = p;
        }
        nnet->SetComponent(c, repl);
      }
    }
    AmNnet am(*nnet);
    delete nnet;
    if (n_priors > 0) am.SetPriors(InV(priors, n_priors));
    if (with_header) {
      Output ko(path, binary != 0);
      am.Write(ko.Stream(), binary != 0);
      return ko.Close() ? 0 : -1;
    }
    std::ofstream os(path, std::ios::binary);
    am.Write(os, binary != 0);
    return os.good() ? 0 : -1;
  } catch (...) { return -1; }
}
// AmDiagGmm::Write gmm/am-diag-gmm.cc:163-176 (DiagGmm::Write diag-gmm.cc:705-720): pdf j owns the
// Gaussians [pdf_offsets[j], pdf_offsets[j+1]) of the concatenated arrays; no "\0B" header.
int ref_write_am_diag_gmm(const char *path, const float *weights, const float *means, const float *vars,
                          const int32_t *pdf_offsets, int num_pdfs, int dim, int binary) {
  try {
    AmDiagGmm am;
    for (int j = 0; j < num_pdfs; j++) {
      const int b = pdf_offsets[j], n = pdf_offsets[j + 1] - b;
      DiagGmm gmm;
      FillGmm(&gmm, weights + b, means + static_cast<size_t>(b) * dim, vars + static_cast<size_t>(b) * dim, n, dim);
      gmm.ComputeGconsts();
      am.AddPdf(gmm);
    }
    std::ofstream os(path, std::ios::binary);
    am.Write(os, binary != 0);
    return os.good() ? 0 : -1;
  } catch (...) { return -1; }
}
// One DiagGmm as a Kaldi file ("\0B" header + DiagGmm::Write, diag-gmm.cc:705-720): final.dubm
int ref_write_diag_gmm_file(const char *path, const float *weights, const float *means, const float *vars, int n, int dim,
                            int binary) {
  try {
    DiagGmm gmm;
    FillGmm(&gmm, weights, means, vars, n, dim);
    gmm.ComputeGconsts();
    Output ko(path, binary != 0);
    gmm.Write(ko.Stream(), binary != 0);
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
// The byte sequence of IvectorExtractor::Write (ivector/ivector-extractor.cc:706-724) produced with the
// reference's own primitives (WriteToken, Matrix<double>::Write, SpMatrix<double>::Write,
// WriteBasicType) - ivector-extractor.cc itself needs OpenFst headers.  M: [n][feat_dim][ivector_dim],
// Sigma_inv: [n][feat_dim][feat_dim] full (symmetric), w: [n][ivector_dim] or empty (ivector_dim_w = 0).
int ref_write_ivector_extractor(const char *path, const double *w, int w_rows, int w_cols, const double *w_vec, int n,
                                int feat_dim, int ivector_dim, const double *M, const double *Sigma_inv, double prior_offset,
                                int binary) {
  try {
    Output ko(path, binary != 0);
    std::ostream &os = ko.Stream();
    const bool b = binary != 0;
    WriteToken(os, b, "<IvectorExtractor>");
    WriteToken(os, b, "<w>");
    Matrix<double> wm(w_rows, w_cols);
    for (int r = 0; r < w_rows; r++)
      for (int c = 0; c < w_cols; c++) wm(r, c) = w[static_cast<size_t>(r) * w_cols + c];
    wm.Write(os, b);
    WriteToken(os, b, "<w_vec>");
    Vector<double> wv(n);
    for (int i = 0; i < n; i++) wv(i) = w_vec[i];
    wv.Write(os, b);
    WriteToken(os, b, "<M>");
    int32 size = n;
    WriteBasicType(os, b, size);
    for (int i = 0; i < n; i++) {
      Matrix<double> Mi(feat_dim, ivector_dim);
      for (int r = 0; r < feat_dim; r++)
        for (int c = 0; c < ivector_dim; c++) Mi(r, c) = M[(static_cast<size_t>(i) * feat_dim + r) * ivector_dim + c];
      Mi.Write(os, b);
    }
    WriteToken(os, b, "<SigmaInv>");
    for (int i = 0; i < n; i++) {
      SpMatrix<double> Si(feat_dim);
      for (int r = 0; r < feat_dim; r++)
        for (int c = 0; c <= r; c++) Si(r, c) = Sigma_inv[(static_cast<size_t>(i) * feat_dim + r) * feat_dim + c];
      Si.Write(os, b);
    }
    WriteToken(os, b, "<IvectorOffset>");
    WriteBasicType(os, b, prior_offset);
    WriteToken(os, b, "</IvectorExtractor>");
    return ko.Close() ? 0 : -1;
  } catch (...) { return -1; }
}
// WaveData::Write (feat/wave-reader.cc): 16-bit PCM RIFF file of one channel
int ref_write_wave(const char *path, const float *samples, int n, float samp_freq) {
  try {
    Matrix<BaseFloat> data(1, n);
    for (int i = 0; i < n; i++) data(0, i) = samples[i];
    WaveData wave(samp_freq, data);
    std::ofstream os(path, std::ios::binary);
    wave.Write(os);
    return os.good() ? 0 : -1;
  } catch (...) { return -1; }
}
// HmmTopology: Read (text) then Write hmm/hmm-topology.cc:39-196
int ref_write_topology(const char *path, const char *topo_text, int binary) {
  try {
    std::istringstream is(topo_text);
    HmmTopology topo;
    topo.Read(is, false);
    std::ofstream os(path, std::ios::binary);
    topo.Write(os, binary != 0);
    return os.good() ? 0 : -1;
  } catch (...) { return -1; }
}
// The feature chain in front of the iVector extractor exactly as OnlineIvectorFeature builds it
// (online2/online-ivector-feature.cc:333-360; that file itself needs OpenFst headers through
// decoder/lattice-faster-online-decoder.h, these classes do not): base -> OnlineSpliceFrames
// -> OnlineTransform(lda) and base -> OnlineCmvn(global stats only) -> OnlineSpliceFrames ->
// OnlineTransform(lda), feat/online-feature.cc.
int ref_online_cmvn_splice_lda(const float *feats, int T, int D, const double *global_stats, int cmn_window,
                               int speaker_frames, int global_frames, int norm_mean, int norm_var, int left, int right,
                               const float *lda, int lda_rows, int lda_cols, float *out_lda, float *out_lda_norm,
                               float *out_cmvn) {
  try {
    Matrix<BaseFloat> m = In(feats, T, D, D);
    OnlineMatrixFeature base(m);
    OnlineCmvnOptions co;
    co.cmn_window = cmn_window; co.speaker_frames = speaker_frames; co.global_frames = global_frames;
    co.normalize_mean = norm_mean != 0; co.normalize_variance = norm_var != 0;
    Matrix<double> gs(2, D + 1);
    for (int r = 0; r < 2; r++) for (int c = 0; c <= D; c++) gs(r, c) = global_stats[r * (D + 1) + c];
    OnlineSpliceOptions so;
    so.left_context = left; so.right_context = right;
    Matrix<BaseFloat> lda_mat = In(lda, lda_rows, lda_cols, lda_cols);
    OnlineSpliceFrames splice(so, &base);
    OnlineTransform lda_t(lda_mat, &splice);
    OnlineCmvnState st(gs);
    OnlineCmvn cmvn(co, st, &base);
    OnlineSpliceFrames splice_n(so, &cmvn);
    OnlineTransform lda_n(lda_mat, &splice_n);
    Vector<BaseFloat> v(lda_rows), c(D);
    for (int t = 0; t < T; t++) {
      lda_t.GetFrame(t, &v);
      for (int k = 0; k < lda_rows; k++) out_lda[static_cast<size_t>(t) * lda_rows + k] = v(k);
      lda_n.GetFrame(t, &v);
      for (int k = 0; k < lda_rows; k++) out_lda_norm[static_cast<size_t>(t) * lda_rows + k] = v(k);
      if (out_cmvn) {
        cmvn.GetFrame(t, &c);
        for (int k = 0; k < D; k++) out_cmvn[static_cast<size_t>(t) * D + k] = c(k);
      }
    }
    return 0;
  } catch (...) { return -1; }
}
// OnlineCmvn with a SPEAKER state (OnlineCmvnState::speaker_cmvn_stats, what
// OnlineIvectorExtractorAdaptationState carries from one utterance of a speaker to the next):
// the normalised frames, and the state GetState(last frame) returns.  speaker_stats: [2 x (D + 1)]
// or NULL (no rows).
int ref_online_cmvn_speaker(const float *feats, int T, int D, const double *global_stats, const double *speaker_stats,
                            int cmn_window, int speaker_frames, int global_frames, int norm_mean, int norm_var,
                            float *out_cmvn, double *out_speaker_stats) {
  try {
    Matrix<BaseFloat> m = In(feats, T, D, D);
    OnlineMatrixFeature base(m);
    OnlineCmvnOptions co;
    co.cmn_window = cmn_window; co.speaker_frames = speaker_frames; co.global_frames = global_frames;
    co.normalize_mean = norm_mean != 0; co.normalize_variance = norm_var != 0;
    Matrix<double> gs(2, D + 1);
    for (int r = 0; r < 2; r++) for (int c = 0; c <= D; c++) gs(r, c) = global_stats[r * (D + 1) + c];
    OnlineCmvnState st(gs);
    if (speaker_stats) {
      st.speaker_cmvn_stats.Resize(2, D + 1);
      for (int r = 0; r < 2; r++) for (int c = 0; c <= D; c++) st.speaker_cmvn_stats(r, c) = speaker_stats[r * (D + 1) + c];
    }
    OnlineCmvn cmvn(co, st, &base);
    Vector<BaseFloat> c(D);
    for (int t = 0; t < T; t++) {
      cmvn.GetFrame(t, &c);
      for (int k = 0; k < D; k++) out_cmvn[static_cast<size_t>(t) * D + k] = c(k);
    }
    OnlineCmvnState fin;
    cmvn.GetState(T - 1, &fin);
    for (int r = 0; r < 2; r++) for (int cc = 0; cc <= D; cc++) out_speaker_stats[r * (D + 1) + cc] = fin.speaker_cmvn_stats(r, cc);
    return 0;
  } catch (...) { return -1; }
}
// LinearCgd<double> matrix/optimization.cc:453-565 (what OnlineIvectorEstimationStats::GetIvector
// calls, ivector/ivector-extractor.cc:631-655); A in SpMatrix packed order.  Returns the iterations.
int ref_linear_cgd(int dim, const double *a_packed, const double *b, double *x, int max_iters) {
  try {
    SpMatrix<double> A(dim);
    memcpy(A.Data(), a_packed, sizeof(double) * dim * (dim + 1) / 2);
    Vector<double> bv(dim), xv(dim);
    for (int i = 0; i < dim; i++) { bv(i) = b[i]; xv(i) = x[i]; }
    LinearCgdOptions opts;
    opts.max_iters = max_iters;
    int it = LinearCgd(opts, A, bv, &xv);
    for (int i = 0; i < dim; i++) x[i] = xv(i);
    return it;
  } catch (...) { return -1; }
}
}  // extern "C"
