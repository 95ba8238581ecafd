// kaldi_oracle.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See kaldi_oracle.h.
//
// CPU restatement of the matrix / nnet2-forward / DiagGmm part of the hot path
// (SURVEY.md §8 rows a1-a9).  Pinned against the reference's own CPU code
// compiled from /root/reference (oracle/_ref) via tests/golden/make_golden.py
// and tests/test_oracle_golden.py.
//
// Compile with -ffp-contract=off: every expression below is written in the
// association order of the reference line it cites.

#include "kaldi_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <set>
#include <vector>

namespace {

const float kMinLogDiffFloat = logf(FLT_EPSILON);  // base/kaldi-math.h:121

inline const float *Row(const float *m, int r, int stride) {
  return m + static_cast<size_t>(r) * stride;
}
inline float *Row(float *m, int r, int stride) {
  return m + static_cast<size_t>(r) * stride;
}

}  // namespace

extern "C" {

// ---- a1: AddMatMat, matrix/kaldi-matrix.cc:160-175 --------------------------
void ko_add_mat_mat(float alpha, const float *A, int a_rows, int a_cols,
                    int a_stride, int transA, const float *B, int b_rows,
                    int b_cols, int b_stride, int transB, float beta, float *C,
                    int c_rows, int c_cols, int c_stride) {
  const int m = transA ? a_cols : a_rows;
  const int k = transA ? a_rows : a_cols;
  const int n = transB ? b_rows : b_cols;
  const int kb = transB ? b_cols : b_rows;
  if (m != c_rows || n != c_cols || k != kb) abort();  // KALDI_ASSERT :166-171
  // Pack op(B) as [n][k] so the inner loop is contiguous.
  std::vector<float> bt(static_cast<size_t>(n) * k);
  for (int j = 0; j < n; j++)
    for (int p = 0; p < k; p++)
      bt[static_cast<size_t>(j) * k + p] =
          transB ? B[static_cast<size_t>(j) * b_stride + p]
                 : B[static_cast<size_t>(p) * b_stride + j];
  std::vector<float> arow(k);
  for (int i = 0; i < m; i++) {
    for (int p = 0; p < k; p++)
      arow[p] = transA ? A[static_cast<size_t>(p) * a_stride + i]
                       : A[static_cast<size_t>(i) * a_stride + p];
    float *crow = Row(C, i, c_stride);
    for (int j = 0; j < n; j++) {
      const float *b = &bt[static_cast<size_t>(j) * k];
      float acc = 0.0f;
      for (int p = 0; p < k; p++) acc = fmaf(arow[p], b[p], acc);
      float prod = alpha * acc;
      crow[j] = (beta == 0.0f) ? prod : (beta * crow[j] + prod);
    }
  }
}

// ---- a2: softmax, matrix/kaldi-vector.cc:840-847 -----------------------------
void ko_softmax_per_row(const float *src, int rows, int cols, int src_stride,
                        float *dst, int dst_stride) {
  for (int r = 0; r < rows; r++) {
    const float *x = Row(src, r, src_stride);
    float *y = Row(dst, r, dst_stride);
    float max = -std::numeric_limits<float>::infinity();
    for (int c = 0; c < cols; c++) max = std::max(max, x[c]);
    float sum = 0.0f;
    for (int c = 0; c < cols; c++) sum += (y[c] = expf(x[c] - max));
    float scale = 1.0f / sum;  // this->Scale(1.0 / sum): Real(1.0/sum)
    for (int c = 0; c < cols; c++) y[c] *= scale;
  }
}

// kaldi-vector.cc:849-860
void ko_log_softmax_per_row(const float *src, int rows, int cols,
                            int src_stride, float *dst, int dst_stride) {
  for (int r = 0; r < rows; r++) {
    const float *x = Row(src, r, src_stride);
    float *y = Row(dst, r, dst_stride);
    float max = -std::numeric_limits<float>::infinity();
    for (int c = 0; c < cols; c++) max = std::max(max, x[c]);
    float sum = 0.0f;
    for (int c = 0; c < cols; c++) sum += expf((y[c] = x[c] - max));
    sum = logf(sum);
    float neg = -1.0f * sum;
    for (int c = 0; c < cols; c++) y[c] += neg;
  }
}

// ---- a3: CopyRows, matrix/kaldi-matrix.cc:2570-2584 ---------------------------
void ko_copy_rows(float *dst, int rows, int cols, int dst_stride,
                  const float *src, int src_stride, const int32_t *indices) {
  for (int r = 0; r < rows; r++) {
    float *d = Row(dst, r, dst_stride);
    int32_t idx = indices[r];
    if (idx < 0)
      memset(d, 0, sizeof(float) * cols);
    else
      memcpy(d, Row(src, idx, src_stride), sizeof(float) * cols);
  }
}

// ---- a4: cu::Splice, cudamatrix/cu-math.cc:147-163 ----------------------------
void ko_splice(const float *src, int rows, int cols, int src_stride,
               const int32_t *frame_offsets, int n_offsets, float *tgt,
               int tgt_stride) {
  for (int r = 0; r < rows; r++) {
    for (int off = 0; off < n_offsets; off++) {
      int r_off = r + frame_offsets[off];
      if (r_off < 0) r_off = 0;
      if (r_off >= rows) r_off = rows - 1;
      memcpy(Row(tgt, r, tgt_stride) + static_cast<size_t>(off) * cols,
             Row(src, r_off, src_stride), sizeof(float) * cols);
    }
  }
}

// ---- a5: GroupPnorm, kaldi-matrix.cc:2512-2520 + Norm kaldi-vector.cc:508-545 --
static float VecNorm(const float *d, int dim, float p) {
  float sum = 0.0f;
  if (p == 0.0f) {
    for (int i = 0; i < dim; i++)
      if (d[i] != 0.0f) sum += 1.0f;
    return sum;
  } else if (p == 1.0f) {
    for (int i = 0; i < dim; i++) sum += std::abs(d[i]);
    return sum;
  } else if (p == 2.0f) {
    for (int i = 0; i < dim; i++) sum += d[i] * d[i];
    return std::sqrt(sum);
  } else {
    float tmp;
    bool ok = true;
    for (int i = 0; i < dim; i++) {
      // unqualified pow() on floats binds to ::pow(double,double) in the
      // reference build (verified against oracle/_ref): double pow, float store.
      tmp = static_cast<float>(pow(static_cast<double>(std::abs(d[i])), static_cast<double>(p)));
      if (tmp == HUGE_VALF) ok = false;
      sum += tmp;
    }
    tmp = static_cast<float>(pow(static_cast<double>(sum), static_cast<double>(static_cast<float>(1.0 / p))));
    if (ok) return tmp;
    float maximum = d[0], minimum = d[0];
    for (int i = 1; i < dim; i++) {
      maximum = std::max(maximum, d[i]);
      minimum = std::min(minimum, d[i]);
    }
    float max_abs = std::max(maximum, -minimum);
    std::vector<float> t(d, d + dim);
    float s = 1.0f / max_abs;  // tmp.Scale(1.0 / max_abs)
    for (int i = 0; i < dim; i++) t[i] *= s;
    return VecNorm(t.data(), dim, p) * max_abs;
  }
}

void ko_group_pnorm(const float *src, int rows, int src_cols, int src_stride,
                    float power, float *dst, int dst_cols, int dst_stride) {
  if (src_cols % dst_cols != 0) abort();
  int group = src_cols / dst_cols;
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < dst_cols; j++)
      Row(dst, i, dst_stride)[j] =
          VecNorm(Row(src, i, src_stride) + j * group, group, power);
}

// ---- a6 ---------------------------------------------------------------------
// kaldi-vector.cc:1271-1281; cblas_sdot restated as a float fmaf-free sum.
void ko_add_diag_mat2(float alpha, const float *M, int rows, int cols,
                      int stride, float beta, float *v) {
  for (int i = 0; i < rows; i++) {
    const float *x = Row(M, i, stride);
    float dot = 0.0f;
    for (int c = 0; c < cols; c++) dot += x[c] * x[c];
    v[i] = beta * v[i] + alpha * dot;
  }
}

void ko_mul_rows_vec(float *M, int rows, int cols, int stride, const float *s) {
  for (int i = 0; i < rows; i++) {
    float *x = Row(M, i, stride);
    float si = s[i];
    for (int c = 0; c < cols; c++) x[c] *= si;
  }
}

void ko_mul_cols_vec(float *M, int rows, int cols, int stride, const float *s) {
  for (int i = 0; i < rows; i++) {
    float *x = Row(M, i, stride);
    for (int c = 0; c < cols; c++) x[c] *= s[c];
  }
}

// nnet2/nnet-component.cc:571-588
void ko_normalize(const float *src, int rows, int cols, int src_stride,
                  float *dst, int dst_stride) {
  const float kNormFloor = static_cast<float>(pow(2.0, -66));  // :571
  for (int r = 0; r < rows; r++)
    memcpy(Row(dst, r, dst_stride), Row(src, r, src_stride),
           sizeof(float) * cols);
  std::vector<float> norm(rows, 0.0f);
  // in_norm.AddDiagMat2(1.0 / in.NumCols(), in, kNoTrans, 0.0): alpha is
  // BaseFloat(1.0 / cols) (double division, then converted to float).
  float alpha = static_cast<float>(1.0 / cols);
  ko_add_diag_mat2(alpha, src, rows, cols, src_stride, 0.0f, norm.data());
  for (int r = 0; r < rows; r++) {
    if (norm[r] < kNormFloor) norm[r] = kNormFloor;  // ApplyFloor
    norm[r] = static_cast<float>(pow(static_cast<double>(norm[r]), -0.5));  // ApplyPow(-0.5): generic pow() branch (double pow)
  }
  ko_mul_rows_vec(dst, rows, cols, dst_stride, norm.data());
}

// ---- a7 ---------------------------------------------------------------------
void ko_copy_rows_from_vec(float *M, int rows, int cols, int stride,
                           const float *v) {
  for (int r = 0; r < rows; r++)
    memcpy(Row(M, r, stride), v, sizeof(float) * cols);
}

// cu-matrix.cc:916-939 CPU branch: if (beta != 1.0) Mat().Scale(beta);
// Mat().AddVecToRows(alpha, row.Vec()) -> M(r,c) += alpha * v(c).
void ko_add_vec_to_rows(float alpha, const float *v, float beta, float *M,
                        int rows, int cols, int stride) {
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++) {
      float cur = (beta != 1.0f) ? beta * x[c] : x[c];
      x[c] = cur + alpha * v[c];
    }
  }
}

void ko_apply_floor(float *M, int rows, int cols, int stride, float floor_val) {
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++)
      if (x[c] < floor_val) x[c] = floor_val;
  }
}

void ko_apply_log(float *M, int rows, int cols, int stride) {
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++) x[c] = logf(x[c]);
  }
}

void ko_apply_exp(float *M, int rows, int cols, int stride) {
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++) x[c] = expf(x[c]);
  }
}

// kaldi-vector.cc:448-469
void ko_apply_pow(float *M, int rows, int cols, int stride, float power) {
  if (power == 1.0f) return;
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++) {
      if (power == 2.0f)
        x[c] = x[c] * x[c];
      else if (power == 0.5f)
        x[c] = std::sqrt(x[c]);
      else
        x[c] = static_cast<float>(pow(static_cast<double>(x[c]), static_cast<double>(power)));
    }
  }
}

void ko_scale(float *M, int rows, int cols, int stride, float alpha) {
  for (int r = 0; r < rows; r++) {
    float *x = Row(M, r, stride);
    for (int c = 0; c < cols; c++) x[c] *= alpha;
  }
}

// cu-matrix.cc:2012-2027 CPU branch
void ko_sum_column_ranges(float *dst, int rows, int dst_cols, int dst_stride,
                          const float *src, int src_stride,
                          const int32_t *ranges) {
  for (int r = 0; r < rows; r++) {
    const float *x = Row(src, r, src_stride);
    float *y = Row(dst, r, dst_stride);
    for (int c = 0; c < dst_cols; c++) {
      int start = ranges[2 * c], end = ranges[2 * c + 1];
      float sum = 0.0f;
      for (int j = start; j < end; j++) sum += x[j];
      y[c] = sum;
    }
  }
}

void ko_matrix_lookup(const float *M, int rows, int cols, int stride,
                      const int32_t *row_col_pairs, int n, float *out) {
  for (int k = 0; k < n; k++) {
    int r = row_col_pairs[2 * k], c = row_col_pairs[2 * k + 1];
    if (r < 0 || r >= rows || c < 0 || c >= cols) abort();
    out[k] = Row(M, r, stride)[c];
  }
}

// ---- a8: nnet2 forward --------------------------------------------------------
namespace {

// ChunkInfo, nnet2/nnet-component.h:72-146 (single chunk).
struct Chunk {
  int first, last;
  std::vector<int> offsets;  // empty if contiguous
  int Size() const {
    return offsets.empty() ? last - first + 1 : static_cast<int>(offsets.size());
  }
  int GetOffset(int index) const {  // nnet-component.cc ChunkInfo::GetOffset
    return offsets.empty() ? first + index : offsets[index];
  }
  int GetIndex(int offset) const {  // ChunkInfo::GetIndex
    if (offsets.empty()) {
      if (offset < first || offset > last) abort();
      return offset - first;
    }
    std::vector<int>::const_iterator it =
        std::lower_bound(offsets.begin(), offsets.end(), offset);
    if (it == offsets.end() || *it != offset) abort();
    return static_cast<int>(it - offsets.begin());
  }
};

std::vector<int> Context(const KoComponent &c) {  // Component::Context()
  if (c.type == KO_SPLICE) return std::vector<int>(c.context, c.context + c.n_context);
  return std::vector<int>(1, 0);
}

}  // namespace

int ko_nnet_left_context(const KoComponent *comps, int n) {  // nnet-nnet.cc:45-53
  int ans = 0;
  for (int i = 0; i < n; i++) ans += Context(comps[i]).front();
  return -ans;
}
int ko_nnet_right_context(const KoComponent *comps, int n) {  // :55-63
  int ans = 0;
  for (int i = 0; i < n; i++) ans += Context(comps[i]).back();
  return ans;
}

int ko_nnet_forward(const KoComponent *comps, int n_comps, const float *feats,
                    int T, int feat_stride, int pad_input, float *out,
                    int out_stride) {
  if (n_comps <= 0 || T <= 0) return -1;
  const int dim = comps[0].input_dim;
  const int left = pad_input ? ko_nnet_left_context(comps, n_comps) : 0;
  const int right = pad_input ? ko_nnet_right_context(comps, n_comps) : 0;
  const int num_rows = left + T + right;  // nnet-compute.cc:76

  // Nnet::ComputeChunkInfo, nnet-nnet.cc:65-112
  const int L = ko_nnet_left_context(comps, n_comps),
            R = ko_nnet_right_context(comps, n_comps);
  const int out_rows = num_rows - L - R;
  if (out_rows <= 0) return -2;
  std::vector<Chunk> info(n_comps + 1);
  std::vector<int> cur;
  for (int i = 0; i < out_rows; i++) cur.push_back(i + L);
  info[n_comps].first = cur.front();
  info[n_comps].last = cur.back();
  for (int i = n_comps - 1; i >= 0; i--) {
    std::vector<int> ctx = Context(comps[i]);
    std::set<int> s;
    for (size_t j = 0; j < ctx.size(); j++)
      for (size_t k = 0; k < cur.size(); k++) s.insert(ctx[j] + cur[k]);
    cur.assign(s.begin(), s.end());
    info[i].first = cur.front();
    info[i].last = cur.back();
    if (static_cast<int>(cur.size()) != cur.back() - cur.front() + 1)
      info[i].offsets = cur;
  }

  // NnetComputer ctor, nnet-compute.cc:63-90: pad by edge-frame duplication.
  std::vector<float> in(static_cast<size_t>(num_rows) * dim);
  for (int r = 0; r < num_rows; r++) {
    int srow = r - left;
    if (srow < 0) srow = 0;
    if (srow > T - 1) srow = T - 1;
    memcpy(&in[static_cast<size_t>(r) * dim], Row(feats, srow, feat_stride),
           sizeof(float) * dim);
  }
  if (info[0].Size() != num_rows) return -3;  // ChunkInfo::CheckSize

  int in_rows = num_rows, in_dim = dim;
  std::vector<float> cur_out;
  for (int c = 0; c < n_comps; c++) {  // NnetComputer::Propagate :94-108
    const KoComponent &comp = comps[c];
    if (comp.input_dim != in_dim) return -4;
    const int o_rows = info[c + 1].Size(), o_dim = comp.output_dim;
    cur_out.assign(static_cast<size_t>(o_rows) * o_dim, 0.0f);
    switch (comp.type) {
      case KO_SPLICE: {  // nnet-component.cc:2628-2708
        const int const_dim = comp.const_dim, sd = in_dim - const_dim;
        if (o_dim != sd * comp.n_context + const_dim) return -5;
        std::vector<int32_t> idx(o_rows);
        for (int k = 0; k < comp.n_context; k++) {
          for (int oi = 0; oi < o_rows; oi++)
            idx[oi] = info[c].GetIndex(info[c + 1].GetOffset(oi) + comp.context[k]);
          ko_copy_rows(cur_out.data() + k * sd, o_rows, sd, o_dim, in.data(),
                       in_dim, idx.data());
        }
        if (const_dim != 0) {
          for (int oi = 0; oi < o_rows; oi++) idx[oi] = oi;  // :2682-2684
          ko_copy_rows(cur_out.data() + (o_dim - const_dim), o_rows, const_dim,
                       o_dim, in.data() + (in_dim - const_dim), in_dim,
                       idx.data());
        }
        break;
      }
      case KO_FIXED_AFFINE:  // :3333-3343
        if (o_rows != in_rows) return -6;
        ko_add_mat_mat(1.0f, in.data(), in_rows, in_dim, in_dim, 0, comp.linear,
                       o_dim, in_dim, in_dim, 1, 0.0f, cur_out.data(), o_rows,
                       o_dim, o_dim);
        ko_add_vec_to_rows(1.0f, comp.bias, 1.0f, cur_out.data(), o_rows, o_dim,
                           o_dim);
        break;
      case KO_AFFINE:  // :1212-1224
        if (o_rows != in_rows) return -6;
        ko_copy_rows_from_vec(cur_out.data(), o_rows, o_dim, o_dim, comp.bias);
        ko_add_mat_mat(1.0f, in.data(), in_rows, in_dim, in_dim, 0, comp.linear,
                       o_dim, in_dim, in_dim, 1, 1.0f, cur_out.data(), o_rows,
                       o_dim, o_dim);
        break;
      case KO_PNORM:  // :518-527
        ko_group_pnorm(in.data(), in_rows, in_dim, in_dim, comp.p,
                       cur_out.data(), o_dim, o_dim);
        break;
      case KO_NORMALIZE:  // :576-588
        ko_normalize(in.data(), in_rows, in_dim, in_dim, cur_out.data(), o_dim);
        break;
      case KO_SOFTMAX:  // :926-943
        ko_softmax_per_row(in.data(), in_rows, in_dim, in_dim, cur_out.data(),
                           o_dim);
        ko_apply_floor(cur_out.data(), o_rows, o_dim, o_dim, 1.0e-20f);
        break;
      case KO_SUM_GROUP: {  // :2491-2499; indexes_ from Init(sizes) :2440-2456
        std::vector<int32_t> ranges(2 * comp.n_sizes);
        int cur_index = 0;
        for (int i = 0; i < comp.n_sizes; i++) {
          ranges[2 * i] = cur_index;
          cur_index += comp.sizes[i];
          ranges[2 * i + 1] = cur_index;
        }
        if (cur_index != in_dim || comp.n_sizes != o_dim) return -7;
        ko_sum_column_ranges(cur_out.data(), o_rows, o_dim, o_dim, in.data(),
                             in_dim, ranges.data());
        break;
      }
      case KO_FIXED_SCALE:  // :3413-3419
        cur_out = in;
        ko_mul_cols_vec(cur_out.data(), o_rows, o_dim, o_dim, comp.bias);
        break;
      case KO_FIXED_BIAS:  // :3483-3489
        cur_out = in;
        ko_add_vec_to_rows(1.0f, comp.bias, 1.0f, cur_out.data(), o_rows, o_dim,
                           o_dim);
        break;
      default:
        return -8;
    }
    in.swap(cur_out);
    in_rows = o_rows;
    in_dim = o_dim;
  }
  for (int r = 0; r < in_rows; r++)
    memcpy(Row(out, r, out_stride), &in[static_cast<size_t>(r) * in_dim],
           sizeof(float) * in_dim);
  return in_rows;
}

// nnet2/decodable-am-nnet.h:39-73
int ko_decodable_am_nnet(const KoComponent *comps, int n_comps,
                         const float *priors, float prob_scale,
                         const float *feats, int T, int feat_stride,
                         float *log_probs, int out_stride) {
  int rows = ko_nnet_forward(comps, n_comps, feats, T, feat_stride, 1,
                             log_probs, out_stride);
  if (rows < 0) return rows;
  const int n = comps[n_comps - 1].output_dim;
  ko_apply_floor(log_probs, rows, n, out_stride, 1.0e-20f);  // :60
  ko_apply_log(log_probs, rows, n, out_stride);              // :61
  std::vector<float> log_priors(priors, priors + n);         // :62-65
  for (int i = 0; i < n; i++) log_priors[i] = logf(log_priors[i]);
  ko_add_vec_to_rows(-1.0f, log_priors.data(), 1.0f, log_probs, rows, n,
                     out_stride);                            // :67
  ko_scale(log_probs, rows, n, out_stride, prob_scale);      // :69
  return rows;
}

// ---- a9: DiagGmm --------------------------------------------------------------
// gmm/diag-gmm.cc:114-152
int ko_gmm_compute_gconsts(const float *weights, const float *means_invvars,
                           const float *inv_vars, int num_mix, int dim,
                           float *gconsts) {
  const double kLog2Pi = 1.8378770664093454835606594728112;  // M_LOG_2PI
  float offset = -0.5 * kLog2Pi * dim;
  int num_bad = 0;
  for (int mix = 0; mix < num_mix; mix++) {
    float gc = logf(weights[mix]) + offset;
    for (int d = 0; d < dim; d++) {
      float iv = inv_vars[static_cast<size_t>(mix) * dim + d];
      float mi = means_invvars[static_cast<size_t>(mix) * dim + d];
      // gc += 0.5 * Log(iv) - 0.5 * mi * mi / iv  (double arithmetic: the 0.5
      // literals promote the products; the sum is rounded to float by +=).
      gc += 0.5 * logf(iv) - 0.5 * mi * mi / iv;
    }
    if (std::isinf(gc)) {
      num_bad++;
      if (gc > 0) gc = -gc;
    }
    gconsts[mix] = gc;
  }
  return num_bad;
}

// gmm/diag-gmm.cc:546-562
void ko_diag_gmm_loglikes(const float *data, int T, int dim, int data_stride,
                          const float *gconsts, const float *means_invvars,
                          const float *inv_vars, int num_mix, float *loglikes,
                          int ll_stride) {
  std::vector<float> sq(static_cast<size_t>(T) * dim);
  for (int t = 0; t < T; t++)
    for (int d = 0; d < dim; d++) {
      float x = Row(data, t, data_stride)[d];
      sq[static_cast<size_t>(t) * dim + d] = x * x;  // ApplyPow(2.0)
    }
  ko_copy_rows_from_vec(loglikes, T, num_mix, ll_stride, gconsts);
  ko_add_mat_mat(1.0f, data, T, dim, data_stride, 0, means_invvars, num_mix,
                 dim, dim, 1, 1.0f, loglikes, T, num_mix, ll_stride);
  ko_add_mat_mat(-0.5f, sq.data(), T, dim, dim, 0, inv_vars, num_mix, dim, dim,
                 1, 1.0f, loglikes, T, num_mix, ll_stride);
}

// matrix/kaldi-vector.cc:745-763
float ko_log_sum_exp(const float *v, int dim, float prune) {
  float max_elem = -std::numeric_limits<float>::infinity();
  for (int i = 0; i < dim; i++) max_elem = std::max(max_elem, v[i]);
  float cutoff = max_elem + kMinLogDiffFloat;
  if (prune > 0.0f && max_elem - prune > cutoff) cutoff = max_elem - prune;
  double sum_relto_max_elem = 0.0;
  for (int i = 0; i < dim; i++) {
    float f = v[i];
    if (f >= cutoff) sum_relto_max_elem += expf(f - max_elem);
  }
  // "return max_elem + Log(sum_relto_max_elem)": double log, double add,
  // converted to Real (float) on return.
  return static_cast<float>(max_elem + log(sum_relto_max_elem));
}

void ko_am_gmm_loglikes(const float *data, int T, int dim, int data_stride,
                        const float *gconsts, const float *means_invvars,
                        const float *inv_vars, const int32_t *pdf_offsets,
                        int num_pdfs, float log_sum_exp_prune, float *out,
                        int out_stride) {
  const int num_mix = pdf_offsets[num_pdfs];
  std::vector<float> ll(static_cast<size_t>(T) * num_mix);
  ko_diag_gmm_loglikes(data, T, dim, data_stride, gconsts, means_invvars,
                       inv_vars, num_mix, ll.data(), num_mix);
  for (int t = 0; t < T; t++)
    for (int j = 0; j < num_pdfs; j++)
      Row(out, t, out_stride)[j] =
          ko_log_sum_exp(&ll[static_cast<size_t>(t) * num_mix + pdf_offsets[j]],
                         pdf_offsets[j + 1] - pdf_offsets[j], log_sum_exp_prune);
}

}  // extern "C"
