// lattice_oracle.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the lattice forward-backward:
//   LatticeStateTimes        lat/lattice-functions.cc:36-67
//   LatticeForwardBackward   lat/lattice-functions.cc:272-354
//   ComputeLatticeAlphasAndBetas           :412-463
//   LatticeForwardBackwardMpeVariants      :740-919
//   RescoreLattice                         :1307-1358
//   CuMatrix::CompObjfAndDeriv (CPU branch) cudamatrix/cu-matrix.cc:1236-1248
//   LogAdd (double)          base/kaldi-math.h:178-195
//   ConvertToCost            fstext/lattice-weight.h:794-806 (value1 + value2)
//
// PARITY UNPINNED by the reference's tests (none exercises LatticeForwardBackward;
// src/lat cannot be compiled here without OpenFst).  Pinned by
// tests/test_lattice_oracle.py: brute-force enumeration of all lattice paths and
// the reference's own self-check forward total == backward total (:346).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <vector>

#include "decoder_oracle.h"

namespace {
const double kLogZeroDouble = -std::numeric_limits<double>::infinity();
const double kMinLogDiffDouble = log(DBL_EPSILON);  // kaldi-math.h:120

// base/kaldi-math.h:178-195
inline double LogAdd(double x, double y) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  // diff is negative.  x is now the larger one.
  if (diff >= kMinLogDiffDouble) {
    double res;
    res = x + log1p(exp(diff));
    return res;
  } else {
    return x;  // return the larger one.
  }
}
}  // namespace

extern "C" double ko_lattice_forward_backward(int num_states, const int64_t *arc_offsets,
                                              const int32_t *arc_ilabel, const int32_t *arc_nextstate,
                                              const float *arc_graph, const float *arc_acoustic,
                                              const float *state_final, float *arc_post,
                                              double *acoustic_like_sum, int32_t *state_times,
                                              double *tot_forward) {
  const float kInfF = std::numeric_limits<float>::infinity();
  // LatticeStateTimes :36-67
  std::vector<int32_t> times(num_states, -1);
  times[0] = 0;
  for (int s = 0; s < num_states; s++) {
    int32_t cur_time = times[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      int32_t ns = arc_nextstate[a];
      if (arc_ilabel[a] != 0) {
        if (times[ns] == -1) times[ns] = cur_time + 1;
        else if (times[ns] != cur_time + 1) abort();  // KALDI_ASSERT :55
      } else {
        if (times[ns] == -1) times[ns] = cur_time;
        else if (times[ns] != cur_time) abort();
      }
    }
  }
  if (state_times)
    for (int s = 0; s < num_states; s++) state_times[s] = times[s];

  if (acoustic_like_sum) *acoustic_like_sum = 0.0;
  std::vector<double> alpha(num_states, kLogZeroDouble);
  std::vector<double> &beta(alpha);  // same memory, :289-291
  double tot_forward_prob = kLogZeroDouble;
  alpha[0] = 0.0;
  for (int s = 0; s < num_states; s++) {  // :300-316
    double this_alpha = alpha[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]);  // -ConvertToCost (float sum)
      alpha[arc_nextstate[a]] = LogAdd(alpha[arc_nextstate[a]], this_alpha + arc_like);
    }
    if (state_final[s] != kInfF) {  // f != Weight::Zero()
      double final_like = this_alpha - static_cast<double>(state_final[s] + 0.0f);  // f.Value1()+f.Value2()
      tot_forward_prob = LogAdd(tot_forward_prob, final_like);
    }
  }
  for (int s = num_states - 1; s >= 0; s--) {  // :317-344
    double this_beta = -static_cast<double>(state_final[s] + 0.0f);  // -(inf) = -inf for non-final
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]),
             arc_beta = beta[arc_nextstate[a]] + arc_like;
      this_beta = LogAdd(this_beta, arc_beta);
      double posterior = exp(alpha[s] + arc_beta - tot_forward_prob);
      if (arc_post) arc_post[a] = static_cast<float>(posterior);
      if (acoustic_like_sum != NULL) *acoustic_like_sum -= posterior * arc_acoustic[a];
    }
    // final-prob term of acoustic_like_sum (:337-341): f.Value2() == 0 for the
    // (final_cost, 0) weights decoder lattices carry, so it contributes nothing.
    beta[s] = this_beta;
  }
  if (tot_forward) *tot_forward = tot_forward_prob;
  return beta[0];
}

namespace {
// base/kaldi-math.h ApproxEqual (double)
inline bool ApproxEqualD(double a, double b, double tol) {
  if (a == b) return true;
  double diff = std::fabs(a - b);
  if (diff == std::numeric_limits<double>::infinity() || diff != diff) return false;
  return diff <= tol * (std::fabs(a) + std::fabs(b));
}
}  // namespace

// ComputeLatticeAlphasAndBetas :412-463 (Lattice instance); returns 0.5 * (fwd + bwd).
extern "C" double ko_lattice_alphas_betas(int num_states, const int64_t *arc_offsets, const int32_t *arc_nextstate,
                                          const float *arc_graph, const float *arc_acoustic,
                                          const float *state_final, int viterbi, double *alpha, double *beta) {
  const float kInfF = std::numeric_limits<float>::infinity();
  for (int s = 0; s < num_states; s++) alpha[s] = beta[s] = kLogZeroDouble;
  double tot_forward_prob = kLogZeroDouble;
  alpha[0] = 0.0;
  for (int s = 0; s < num_states; s++) {
    double this_alpha = alpha[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]);
      double &dst = alpha[arc_nextstate[a]];
      dst = viterbi ? std::max(dst, this_alpha + arc_like) : LogAdd(dst, this_alpha + arc_like);  // LogAddOrMax :395-410
    }
    if (state_final[s] != kInfF) {
      double final_like = this_alpha - static_cast<double>(state_final[s] + 0.0f);
      tot_forward_prob = viterbi ? std::max(tot_forward_prob, final_like) : LogAdd(tot_forward_prob, final_like);
    }
  }
  for (int s = num_states - 1; s >= 0; s--) {
    double this_beta = -static_cast<double>(state_final[s] + 0.0f);
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]),
             arc_beta = beta[arc_nextstate[a]] + arc_like;
      this_beta = viterbi ? std::max(this_beta, arc_beta) : LogAdd(this_beta, arc_beta);
    }
    beta[s] = this_beta;
  }
  return 0.5 * (beta[0] + tot_forward_prob);
}

// LatticeForwardBackwardMpeVariants :740-919.  tid2phone / tid2pdf: TransitionIdToPhone /
// TransitionIdToPdf as arrays indexed by transition-id; silence_phones sorted.  arc_post
// receives posterior_smbr per arc (0 for epsilon arcs; the caller merges per frame as
// MergePairVectorSumming :916-917).  Returns 0 and tot_forward_score, or -1 / -2 when the
// first / second forward-backward check fails (KALDI_ERR :808-811, :909-912).
extern "C" int ko_lattice_forward_backward_mpe(int num_states, const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                               const int32_t *arc_nextstate, const float *arc_graph,
                                               const float *arc_acoustic, const float *state_final,
                                               const int32_t *tid2phone, const int32_t *tid2pdf,
                                               const int32_t *silence_phones, int n_sil, const int32_t *num_ali,
                                               int max_time, int is_mpfe, int one_silence_class, float *arc_post,
                                               double *tot_forward_score_out) {
  const float kInfF = std::numeric_limits<float>::infinity();
  std::vector<int32_t> state_times(num_states, -1);
  state_times[0] = 0;
  int t_max = 0;
  for (int s = 0; s < num_states; s++) {
    int32_t cur_time = state_times[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      int32_t want = cur_time + (arc_ilabel[a] != 0 ? 1 : 0), ns = arc_nextstate[a];
      if (state_times[ns] == -1) state_times[ns] = want;
      else if (state_times[ns] != want) abort();
      if (want > t_max) t_max = want;
    }
  }
  if (t_max != max_time) abort();  // KALDI_ASSERT :764
  std::vector<double> alpha(num_states, kLogZeroDouble), alpha_smbr(num_states, 0), beta(num_states, kLogZeroDouble),
      beta_smbr(num_states, 0);
  double tot_forward_prob = kLogZeroDouble, tot_forward_score = 0;
  alpha[0] = 0.0;
  for (int s = 0; s < num_states; s++) {  // first pass forward :776-791
    double this_alpha = alpha[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]);
      alpha[arc_nextstate[a]] = LogAdd(alpha[arc_nextstate[a]], this_alpha + arc_like);
    }
    if (state_final[s] != kInfF) {
      double final_like = this_alpha - (static_cast<double>(state_final[s]) + 0.0);  // f.Value1() + f.Value2() (float sum)
      tot_forward_prob = LogAdd(tot_forward_prob, final_like);
    }
  }
  for (int s = num_states - 1; s >= 0; s--) {  // first pass backward :793-803
    double this_beta = -static_cast<double>(state_final[s] + 0.0f);
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]), arc_beta = beta[arc_nextstate[a]] + arc_like;
      this_beta = LogAdd(this_beta, arc_beta);
    }
    beta[s] = this_beta;
  }
  if (!ApproxEqualD(tot_forward_prob, beta[0], 1e-6)) return -1;
  auto frame_acc_of = [&](int s, int64_t a) -> double {  // :820-842 / :862-885
    if (arc_ilabel[a] == 0) return 0.0;
    int32_t cur_time = state_times[s];
    int32_t phone = tid2phone[arc_ilabel[a]], ref_phone = tid2phone[num_ali[cur_time]];
    bool phone_is_sil = std::binary_search(silence_phones, silence_phones + n_sil, phone),
         ref_phone_is_sil = std::binary_search(silence_phones, silence_phones + n_sil, ref_phone),
         both_sil = phone_is_sil && ref_phone_is_sil;
    if (!is_mpfe) {
      int32_t pdf = tid2pdf[arc_ilabel[a]], ref_pdf = tid2pdf[num_ali[cur_time]];
      if (!one_silence_class) return (pdf == ref_pdf && !phone_is_sil) ? 1.0 : 0.0;
      return (pdf == ref_pdf || both_sil) ? 1.0 : 0.0;
    }
    if (!one_silence_class) return (phone == ref_phone && !phone_is_sil) ? 1.0 : 0.0;
    return (phone == ref_phone || both_sil) ? 1.0 : 0.0;
  };
  alpha_smbr[0] = 0.0;
  for (int s = 0; s < num_states; s++) {  // second pass forward :813-855
    double this_alpha = alpha[s];
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]);
      double frame_acc = frame_acc_of(s, a);
      double arc_scale = exp(alpha[s] + arc_like - alpha[arc_nextstate[a]]);
      alpha_smbr[arc_nextstate[a]] += arc_scale * (alpha_smbr[s] + frame_acc);
    }
    if (state_final[s] != kInfF) {
      double final_like = this_alpha - (static_cast<double>(state_final[s]) + 0.0);
      double arc_scale = exp(final_like - tot_forward_prob);
      tot_forward_score += arc_scale * alpha_smbr[s];
    }
  }
  for (int s = num_states - 1; s >= 0; s--) {  // second pass backward :857-903
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      double arc_like = -static_cast<double>(arc_graph[a] + arc_acoustic[a]), arc_beta = beta[arc_nextstate[a]] + arc_like;
      double frame_acc = frame_acc_of(s, a);
      double arc_scale = exp(beta[arc_nextstate[a]] + arc_like - beta[s]);
      if (arc_scale != arc_scale) arc_scale = 0;  // KALDI_ISNAN :890
      beta_smbr[s] += arc_scale * (beta_smbr[arc_nextstate[a]] + frame_acc);
      float p = 0.0f;
      if (arc_ilabel[a] != 0) {
        double posterior = exp(alpha[s] + arc_beta - tot_forward_prob);
        double acc_diff = alpha_smbr[s] + frame_acc + beta_smbr[arc_nextstate[a]] - tot_forward_score;
        p = static_cast<float>(posterior * acc_diff);
      }
      if (arc_post) arc_post[a] = p;
    }
  }
  if (!ApproxEqualD(tot_forward_score, beta_smbr[0], 1e-4)) return -2;
  if (tot_forward_score_out) *tot_forward_score_out = tot_forward_score;
  return 0;
}

// RescoreLattice :1307-1358 with a matrix decodable (LogLikelihood(t, tid) =
// loglikes[t][tid2pdf ? tid2pdf[tid] : tid - 1]): the acoustic cost of every arc with a
// transition-id gets -log_like added.  Returns 0, or -1 if the features are too short.
extern "C" int ko_rescore_lattice(int num_states, const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                  const int32_t *arc_nextstate, float *arc_acoustic, const float *loglikes,
                                  int num_frames, int ll_stride, const int32_t *tid2pdf) {
  std::vector<int32_t> times(num_states, -1);
  times[0] = 0;
  int utt_len = 0;
  for (int s = 0; s < num_states; s++)
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      int32_t want = times[s] + (arc_ilabel[a] != 0 ? 1 : 0);
      if (times[arc_nextstate[a]] == -1) times[arc_nextstate[a]] = want;
      if (want > utt_len) utt_len = want;
    }
  if (utt_len > num_frames) return -1;  // "Features are too short for lattice" :1337-1341
  for (int s = 0; s < num_states; s++) {
    int t = times[s];
    if (t < 0 || t >= utt_len) continue;
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++)
      if (arc_ilabel[a] != 0) {
        int pdf = tid2pdf ? tid2pdf[arc_ilabel[a]] : arc_ilabel[a] - 1;
        float log_like = loglikes[static_cast<size_t>(t) * ll_stride + pdf];
        arc_acoustic[a] = -log_like + arc_acoustic[a];  // :1350
      }
  }
  return 0;
}

// CuMatrix::CompObjfAndDeriv, CPU branch cu-matrix.cc:1236-1248.
extern "C" void ko_comp_objf_and_deriv(int n, const int32_t *rows, const int32_t *cols, const float *weights,
                                       const float *output, int out_stride, float *deriv, int deriv_stride,
                                       float *tot_objf, float *tot_weight) {
  *tot_objf = 0.0f;
  *tot_weight = 0.0f;
  for (int i = 0; i < n; i++) {
    int m = rows[i], label = cols[i];
    float weight = weights[i];
    float this_prob = output[static_cast<size_t>(m) * out_stride + label];
    *tot_objf += weight * logf(this_prob);
    *tot_weight += weight;
    deriv[static_cast<size_t>(m) * deriv_stride + label] += weight / this_prob;
  }
}
