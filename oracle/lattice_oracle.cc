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

// ---------------------------------------------------------------------------
// LatticeForwardBackwardMmi lat/lattice-functions.cc:1361-1396 with the Posterior
// algebra it uses, restated from hmm/posterior.cc (ScalePosterior :204-214,
// PosteriorEntriesAreDisjoint :228-238, MergePosteriors :244-274, AlignmentToPosterior
// :276-285, ConvertPosteriorToPdfs :308-332) and util/stl-utils.h:303-322
// (MergePairVectorSumming).  hmm/posterior.cc includes hmm/transition-model.h, which
// needs OpenFst's fst-decl.h: not buildable here, hence restated.  PARITY UNPINNED by the
// reference (no test of it there); pinned by tests/test_lattice_oracle.py through path
// enumeration: den posterior of (t, id) = sum over paths through an arc with that id at
// time t of P(path), MMI posterior = numerator indicator - that.
#include <unordered_map>
#include <unordered_set>
#include <utility>
namespace {
typedef std::vector<std::vector<std::pair<int32_t, float> > > Posterior;

void MergePairVectorSumming(std::vector<std::pair<int32_t, float> > *vec) {  // stl-utils.h:303-322
  std::sort(vec->begin(), vec->end(),
            [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.first < b.first; });
  std::vector<std::pair<int32_t, float> >::iterator out = vec->begin(), in = vec->begin(), end = vec->end();
  while (in < end) {
    *out = *in;
    ++in;
    while (in < end && in->first == out->first) {
      out->second += in->second;
      ++in;
    }
    if (out->second != 0.0f) out++;
  }
  vec->erase(out, end);
}

void ScalePosterior(float scale, Posterior *post) {  // posterior.cc:204-214
  if (scale == 1.0) return;
  for (size_t i = 0; i < post->size(); i++) {
    if (scale == 0.0) (*post)[i].clear();
    else
      for (size_t j = 0; j < (*post)[i].size(); j++) (*post)[i][j].second *= scale;
  }
}

void ConvertPosteriorToPdfs(const int32_t *tid2pdf, const Posterior &post_in, Posterior *post_out) {  // :308-332
  post_out->clear();
  post_out->resize(post_in.size());
  for (size_t i = 0; i < post_out->size(); i++) {
    std::unordered_map<int32_t, float> pdf_to_post;
    for (size_t j = 0; j < post_in[i].size(); j++) {
      int32_t tid = post_in[i][j].first, pdf_id = tid2pdf[tid];
      float post = post_in[i][j].second;
      if (pdf_to_post.count(pdf_id) == 0) pdf_to_post[pdf_id] = post;
      else pdf_to_post[pdf_id] += post;
    }
    (*post_out)[i].reserve(pdf_to_post.size());
    for (std::unordered_map<int32_t, float>::const_iterator iter = pdf_to_post.begin(); iter != pdf_to_post.end(); ++iter)
      if (iter->second != 0.0) (*post_out)[i].push_back(std::make_pair(iter->first, iter->second));
  }
}

bool PosteriorEntriesAreDisjoint(const std::vector<std::pair<int32_t, float> > &e1,
                                 const std::vector<std::pair<int32_t, float> > &e2) {  // :228-238
  std::unordered_set<int32_t> set1;
  for (size_t i = 0; i < e1.size(); i++) set1.insert(e1[i].first);
  for (size_t i = 0; i < e2.size(); i++)
    if (set1.count(e2[i].first) != 0) return false;
  return true;
}

int32_t MergePosteriors(const Posterior &post1, const Posterior &post2, bool merge, bool drop_frames, Posterior *post) {  // :244-274
  if (post1.size() != post2.size()) abort();  // KALDI_ASSERT :249
  post->resize(post1.size());
  int32_t num_disjoint = 0;
  for (size_t i = 0; i < post->size(); i++) {
    (*post)[i].reserve(post1[i].size() + post2[i].size());
    (*post)[i].insert((*post)[i].end(), post1[i].begin(), post1[i].end());
    (*post)[i].insert((*post)[i].end(), post2[i].begin(), post2[i].end());
    if (merge) MergePairVectorSumming(&((*post)[i]));
    else std::sort((*post)[i].begin(), (*post)[i].end());
    if (PosteriorEntriesAreDisjoint(post1[i], post2[i])) {
      num_disjoint++;
      if (drop_frames) (*post)[i].clear();
    }
  }
  return num_disjoint;
}
}  // namespace

// Returns the denominator's total log-likelihood (:1372-1374).  Output posterior as
// frame_offsets[num_frames + 1] + (ids, weights); -1.0e30 and *n_entries = needed if cap is too small.
extern "C" double ko_lattice_forward_backward_mmi(int num_states, const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                                  const int32_t *arc_nextstate, const float *arc_graph,
                                                  const float *arc_acoustic, const float *state_final,
                                                  const int32_t *tid2pdf, const int32_t *num_ali, int n_ali,
                                                  int drop_frames, int convert_to_pdf_ids, int cancel,
                                                  int32_t *frame_offsets, int32_t *ids, float *weights, int cap,
                                                  int32_t *n_entries, int32_t *num_disjoint) {
  const int64_t n_arcs = arc_offsets[num_states];
  std::vector<float> arc_post(n_arcs);
  std::vector<int32_t> times(num_states);
  // LatticeForwardBackward(lat, &den_post, NULL) :1372-1374
  const double ans = ko_lattice_forward_backward(num_states, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic,
                                                 state_final, arc_post.data(), NULL, times.data(), NULL);
  int32_t max_time = 0;
  for (int s = 0; s < num_states; s++) max_time = std::max(max_time, times[s]);
  Posterior den_post(max_time);
  // the reference pushes the entries in its backward sweep (states descending, arcs ascending) :317-333
  for (int s = num_states - 1; s >= 0; s--)
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++)
      if (arc_ilabel[a] != 0) den_post[times[s]].push_back(std::make_pair(arc_ilabel[a], arc_post[a]));
  for (int32_t t = 0; t < max_time; t++) MergePairVectorSumming(&den_post[t]);  // :349-351
  Posterior num_post(n_ali);  // AlignmentToPosterior posterior.cc:276-285
  for (int i = 0; i < n_ali; i++) num_post[i].assign(1, std::make_pair(num_ali[i], 1.0f));
  ScalePosterior(-1.0f, &den_post);  // :1381
  if (convert_to_pdf_ids) {  // :1383-1390
    Posterior num_tmp;
    ConvertPosteriorToPdfs(tid2pdf, num_post, &num_tmp);
    num_tmp.swap(num_post);
    Posterior den_tmp;
    ConvertPosteriorToPdfs(tid2pdf, den_post, &den_tmp);
    den_tmp.swap(den_post);
  }
  Posterior post;
  const int32_t nd = MergePosteriors(num_post, den_post, cancel != 0, drop_frames != 0, &post);  // :1392-1393
  if (num_disjoint) *num_disjoint = nd;
  int32_t total = 0;
  for (size_t t = 0; t < post.size(); t++) total += static_cast<int32_t>(post[t].size());
  *n_entries = total;
  if (total > cap) return -1.0e30;
  int32_t k = 0;
  for (size_t t = 0; t < post.size(); t++) {
    frame_offsets[t] = k;
    for (size_t j = 0; j < post[t].size(); j++) {
      ids[k] = post[t][j].first;
      weights[k] = post[t][j].second;
      k++;
    }
  }
  frame_offsets[post.size()] = k;
  return ans;
}

// ---------------------------------------------------------------------------
// NnetDiscriminativeUpdater::LatticeComputations nnet2/nnet-compute-discriminative.cc:178-321
// for ONE example, given the network's output (the posteriors matrix forward_data_.back()):
// requested indexes (:203-226), Lookup, floor + pseudo log-likelihoods (:231-247), numerator
// likelihood (:253-258), lattice acoustic costs (:261-277), GetDiscriminativePosteriors
// (:324-343), ScalePosterior (:283), sv_labels + CompObjfAndDeriv (:285-316).  boost == 0.
// stats[5] = {tot_t, tot_t_weighted, tot_num_count, tot_num_objf, tot_den_objf} are ADDED to.
// Returns 0, or -k when a forward-backward self-check fails.  PARITY UNPINNED by the
// reference (needs OpenFst); every piece it calls is pinned by path enumeration above.
extern "C" int ko_discriminative_lattice_computations(
    const float *posteriors, int num_frames, int num_pdfs, int post_stride, const float *priors, int num_states,
    const int64_t *arc_offsets, const int32_t *arc_ilabel, const int32_t *arc_nextstate, const float *arc_graph,
    const float *state_final, const int32_t *tid2pdf, const int32_t *tid2phone, const int32_t *silence_phones, int n_sil,
    const int32_t *num_ali, int criterion /*0 mmi, 1 smbr, 2 mpfe*/, float acoustic_scale, int drop_frames,
    int one_silence_class, float weight, double *stats, float *deriv, int deriv_stride) {
  const int64_t n_arcs = arc_offsets[num_states];
  stats[0] += num_frames;
  stats[1] += num_frames * weight;
  std::vector<std::pair<int32_t, int32_t> > requested;
  if (criterion == 0)
    for (int t = 0; t < num_frames; t++) requested.push_back(std::make_pair(t, tid2pdf[num_ali[t]]));
  // LatticeStateTimes :36-67
  std::vector<int32_t> times(num_states, -1);
  times[0] = 0;
  for (int s = 0; s < num_states; s++)
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++) {
      int32_t want = times[s] + (arc_ilabel[a] != 0 ? 1 : 0), ns = arc_nextstate[a];
      if (times[ns] == -1) times[ns] = want;
      else if (times[ns] != want) abort();
    }
  for (int s = 0; s < num_states; s++)
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++)
      if (arc_ilabel[a] != 0) requested.push_back(std::make_pair(times[s], tid2pdf[arc_ilabel[a]]));
  std::vector<float> answers(requested.size());
  const float floor_val = 1.0e-20f;
  for (size_t i = 0; i < requested.size(); i++) {
    float post = posteriors[static_cast<size_t>(requested[i].first) * post_stride + requested[i].second];  // Lookup
    if (post < floor_val) post = floor_val;
    answers[i] = logf(post / priors[requested[i].second]) * acoustic_scale;  // :241
  }
  size_t index = 0;
  if (criterion == 0) {
    double tot_num_like = 0.0;
    for (; index < static_cast<size_t>(num_frames); index++) tot_num_like += answers[index];
    stats[3] += weight * tot_num_like;
  }
  std::vector<float> acoustic(n_arcs, 0.0f), fin(state_final, state_final + num_states);
  for (int s = 0; s < num_states; s++)
    for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++)
      if (arc_ilabel[a] != 0) acoustic[a] = -answers[index++];
  if (index != answers.size()) abort();
  // GetDiscriminativePosteriors :324-343 -> Posterior by pdf
  Posterior post;
  if (criterion == 0) {
    std::vector<int32_t> fo(num_frames + 1), ids(n_arcs + num_frames + 16);
    std::vector<float> w(n_arcs + num_frames + 16);
    int32_t n_ent = 0;
    const double den = ko_lattice_forward_backward_mmi(num_states, arc_offsets, arc_ilabel, arc_nextstate, arc_graph,
                                                       acoustic.data(), fin.data(), tid2pdf, num_ali, num_frames, drop_frames,
                                                       1, 1, fo.data(), ids.data(), w.data(), static_cast<int>(ids.size()),
                                                       &n_ent, NULL);
    stats[4] += weight * den;
    post.resize(num_frames);
    for (int t = 0; t < num_frames; t++)
      for (int k = fo[t]; k < fo[t + 1]; k++) post[t].push_back(std::make_pair(ids[k], w[k]));
  } else {
    std::vector<float> arc_post(n_arcs);
    double score = 0.0;
    const int rc = ko_lattice_forward_backward_mpe(num_states, arc_offsets, arc_ilabel, arc_nextstate, arc_graph,
                                                   acoustic.data(), fin.data(), tid2phone, tid2pdf, silence_phones, n_sil,
                                                   num_ali, num_frames, criterion == 2, one_silence_class, arc_post.data(),
                                                   &score);
    if (rc != 0) return rc;
    stats[4] += weight * score;
    // the Posterior LatticeForwardBackwardMpeVariants returns: (tid, posterior) per frame, merged (:914-916)
    Posterior tid_post(num_frames);
    for (int s = 0; s < num_states; s++)
      for (int64_t a = arc_offsets[s]; a < arc_offsets[s + 1]; a++)
        if (arc_ilabel[a] != 0) tid_post[times[s]].push_back(std::make_pair(arc_ilabel[a], arc_post[a]));
    for (int t = 0; t < num_frames; t++) MergePairVectorSumming(&tid_post[t]);
    ConvertPosteriorToPdfs(tid2pdf, tid_post, &post);
  }
  ScalePosterior(weight, &post);  // :283
  double tot_num_post = 0.0;
  std::vector<int32_t> rows, cols;
  std::vector<float> wts;
  for (size_t t = 0; t < post.size(); t++)
    for (size_t i = 0; i < post[t].size(); i++) {
      const float wgt = post[t][i].second;
      if (wgt > 0.0) tot_num_post += wgt;
      rows.push_back(static_cast<int32_t>(t));
      cols.push_back(post[t][i].first);
      wts.push_back(wgt);
    }
  stats[2] += tot_num_post;
  float tot_objf, tot_weight;
  ko_comp_objf_and_deriv(static_cast<int>(rows.size()), rows.data(), cols.data(), wts.data(), posteriors, post_stride, deriv,
                         deriv_stride, &tot_objf, &tot_weight);
  return 0;
}
