// lattice_oracle.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the lattice forward-backward:
//   LatticeStateTimes        lat/lattice-functions.cc:36-67
//   LatticeForwardBackward   lat/lattice-functions.cc:272-354
//   LogAdd (double)          base/kaldi-math.h:178-195
//   ConvertToCost            fstext/lattice-weight.h:794-806 (value1 + value2)
//
// PARITY UNPINNED by the reference's tests (none exercises LatticeForwardBackward;
// src/lat cannot be compiled here without OpenFst).  Pinned by
// tests/test_lattice_oracle.py: brute-force enumeration of all lattice paths and
// the reference's own self-check forward total == backward total (:346).
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
