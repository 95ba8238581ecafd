// kh_lattice.hip — lattice forward-backward on gfx950.
//
// Replaces LatticeStateTimes (lat/lattice-functions.cc:36-67) and
// LatticeForwardBackward (:272-354) for a batch of top-sorted lattices.
//
// The reference sweeps the states sequentially (alpha forward, beta backward,
// LogAdd in double).  Here each lattice gets one workgroup; states are grouped
// into dependency levels (longest distance from the start state, computed on
// the host in O(arcs) while the CSR is validated), a level is processed by all
// threads at once, and each state accumulates its incoming (alpha) / outgoing
// (beta) arcs in ascending arc order — the same order of LogAdd operands as the
// reference's sequential sweep, so alpha/beta differ from it only through the
// device's exp/log1p.  Arc posteriors are written per arc; the caller merges
// them per (frame, transition-id) as MergePairVectorSumming does (:351-352).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <vector>

#include "kh_common.h"

using namespace kh;

namespace {

constexpr int kThreads = 256;

struct LatDesc {
  int32_t state_b, n_states;   // range in the concatenated state arrays
  int64_t arc_b;               // first arc
  int32_t n_levels, level_b;   // range in level_off (n_levels + 1 entries)
  int32_t final_b, n_final;    // range in final_list
};

// base/kaldi-math.h:178-195 (double)
__device__ __forceinline__ double LogAddD(double x, double y, double min_log_diff) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= min_log_diff) return x + log1p(exp(diff));
  return x;
}

__global__ void __launch_bounds__(kThreads)
ForwardBackwardKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off,
                      const int32_t *__restrict__ arc_next, const float *__restrict__ arc_g,
                      const float *__restrict__ arc_a, const float *__restrict__ state_final,
                      const int32_t *__restrict__ level_off, const int32_t *__restrict__ level_states,
                      const int64_t *__restrict__ in_off, const int64_t *__restrict__ in_arc,
                      const int32_t *__restrict__ in_src, const int32_t *__restrict__ final_list,
                      double *__restrict__ alpha, double *__restrict__ beta,
                      float *__restrict__ arc_post, double *__restrict__ tot_like,
                      double *__restrict__ ac_sum, double min_log_diff) {
  __shared__ double s_tot;
  __shared__ double s_red[kThreads / 64];
  const LatDesc L = lats[blockIdx.x];
  const double kLogZero = -INFINITY;
  double *al = alpha + L.state_b, *be = beta + L.state_b;
  const float *fin = state_final + L.state_b;
  const int32_t *lstates = level_states + L.state_b;
  const int32_t *loff = level_off + L.level_b;
  // ---- forward :300-316
  for (int lv = 0; lv < L.n_levels; lv++) {
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double a = (s == 0) ? 0.0 : kLogZero;
      const int64_t ib = in_off[L.state_b + s], ie = in_off[L.state_b + s + 1];
      for (int64_t j = ib; j < ie; j++) {
        const int64_t arc = in_arc[j];
        const double arc_like = -static_cast<double>(arc_g[arc] + arc_a[arc]);  // -ConvertToCost
        a = LogAddD(a, al[in_src[j]] + arc_like, min_log_diff);
      }
      al[s] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double tot = kLogZero;
    for (int k = 0; k < L.n_final; k++) {  // ascending state order, as the sweep meets them
      const int s = final_list[L.final_b + k];
      const double final_like = al[s] - static_cast<double>(fin[s] + 0.0f);
      tot = LogAddD(tot, final_like, min_log_diff);
    }
    s_tot = tot;
  }
  __syncthreads();
  const double tot_forward = s_tot;
  // ---- backward :317-344
  double my_ac = 0.0;
  for (int lv = L.n_levels - 1; lv >= 0; lv--) {
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double this_beta = -static_cast<double>(fin[s] + 0.0f);
      const int64_t ab = arc_off[L.state_b + s], ae = arc_off[L.state_b + s + 1];
      const double as = al[s];
      for (int64_t arc = ab; arc < ae; arc++) {
        const double arc_like = -static_cast<double>(arc_g[arc] + arc_a[arc]);
        const double arc_beta = be[arc_next[arc]] + arc_like;
        this_beta = LogAddD(this_beta, arc_beta, min_log_diff);
        const double posterior = exp(as + arc_beta - tot_forward);
        arc_post[arc] = static_cast<float>(posterior);
        my_ac -= posterior * static_cast<double>(arc_a[arc]);
      }
      be[s] = this_beta;
    }
    __syncthreads();
  }
  my_ac = kh_wave_sum_d(my_ac);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = my_ac;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads / 64; i++) t += s_red[i];
    ac_sum[blockIdx.x] = t;
    tot_like[blockIdx.x] = be[0];  // tot_backward_prob :345
  }
}

template <class T>
struct DevArr {
  T *p = nullptr;
  ~DevArr() { if (p) PoolFree(p); }
  int Upload(const std::vector<T> &h, hipStream_t st) {
    p = static_cast<T *>(PoolMalloc(sizeof(T) * (h.size() ? h.size() : 1)));
    if (!p) return KH_ENOMEM;
    if (!h.empty()) KH_HIP(hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
    return KH_OK;
  }
  int Alloc(size_t n) {
    p = static_cast<T *>(PoolMalloc(sizeof(T) * (n ? n : 1)));
    return p ? KH_OK : KH_ENOMEM;
  }
};

}  // namespace

extern "C" int kh_lattice_forward_backward(int n_lats, const int32_t *lat_state_offsets,
                                           const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                           const int32_t *arc_nextstate, const float *arc_graph,
                                           const float *arc_acoustic, const float *state_final,
                                           float *arc_post, double *tot_like,
                                           double *acoustic_like_sum, int32_t *state_times) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate &&
               arc_graph && arc_acoustic && state_final);
  const int total_states = lat_state_offsets[n_lats];
  const int64_t total_arcs = arc_offsets[total_states];
  const float inf = std::numeric_limits<float>::infinity();
  std::vector<LatDesc> descs(n_lats);
  std::vector<int32_t> level_off, level_states(total_states), final_list, in_src(total_arcs);
  std::vector<int64_t> in_off(static_cast<size_t>(total_states) + 1, 0), in_arc(total_arcs);
  std::vector<int32_t> times(total_states, -1);
  for (int l = 0; l < n_lats; l++) {
    const int sb = lat_state_offsets[l], ns = lat_state_offsets[l + 1] - sb;
    KH_CHECK_ARG(ns > 0);
    LatDesc &d = descs[l];
    d.state_b = sb;
    d.n_states = ns;
    d.arc_b = arc_offsets[sb];
    std::vector<int32_t> level(ns, 0);
    int max_level = 0;
    // LatticeStateTimes :36-67 + level assignment + topological-order check
    // ("Input lattice must be topologically sorted", :38-39,:285-286).
    times[sb] = 0;
    for (int s = 0; s < ns; s++) {
      const int cur_time = times[sb + s];
      for (int64_t a = arc_offsets[sb + s]; a < arc_offsets[sb + s + 1]; a++) {
        const int nxt = arc_nextstate[a];
        if (nxt <= s || nxt >= ns) {
          SetError("lattice %d: arc %lld (state %d -> %d): input lattice must be topologically sorted",
                   l, static_cast<long long>(a), s, nxt);
          return KH_EINVAL;
        }
        if (cur_time >= 0) {
          const int want = cur_time + (arc_ilabel[a] != 0 ? 1 : 0);
          if (times[sb + nxt] == -1) times[sb + nxt] = want;
          else if (times[sb + nxt] != want) {
            SetError("lattice %d: inconsistent state times at state %d (KALDI_ASSERT lattice-functions.cc:55,61)", l, nxt);
            return KH_EINVAL;
          }
        }
        level[nxt] = std::max(level[nxt], level[s] + 1);
        in_off[static_cast<size_t>(sb) + nxt + 1]++;
      }
      max_level = std::max(max_level, level[s]);
    }
    d.n_levels = max_level + 1;
    d.level_b = static_cast<int32_t>(level_off.size());
    std::vector<int32_t> cnt(d.n_levels + 1, 0);
    for (int s = 0; s < ns; s++) cnt[level[s] + 1]++;
    for (int i = 0; i < d.n_levels; i++) cnt[i + 1] += cnt[i];
    for (int i = 0; i <= d.n_levels; i++) level_off.push_back(cnt[i]);
    std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
    for (int s = 0; s < ns; s++) level_states[sb + fill[level[s]]++] = s;
    d.final_b = static_cast<int32_t>(final_list.size());
    for (int s = 0; s < ns; s++)
      if (state_final[sb + s] != inf) final_list.push_back(s);
    d.n_final = static_cast<int32_t>(final_list.size()) - d.final_b;
  }
  for (size_t i = 0; i < static_cast<size_t>(total_states); i++) in_off[i + 1] += in_off[i];
  {
    std::vector<int64_t> fill(in_off.begin(), in_off.end() - 1);
    for (int l = 0; l < n_lats; l++) {
      const int sb = lat_state_offsets[l], ns = lat_state_offsets[l + 1] - sb;
      for (int s = 0; s < ns; s++)
        for (int64_t a = arc_offsets[sb + s]; a < arc_offsets[sb + s + 1]; a++) {
          const int64_t pos = fill[static_cast<size_t>(sb) + arc_nextstate[a]]++;
          in_arc[pos] = a;  // ascending arc index per destination
          in_src[pos] = s;
        }
    }
  }
  if (state_times) memcpy(state_times, times.data(), sizeof(int32_t) * total_states);

  hipStream_t st = Stream();
  DevArr<LatDesc> d_descs;
  DevArr<int64_t> d_arc_off, d_in_off, d_in_arc;
  DevArr<int32_t> d_next, d_level_off, d_level_states, d_in_src, d_final_list;
  DevArr<float> d_g, d_a, d_fin, d_post;
  DevArr<double> d_alpha, d_beta, d_tot, d_ac;
  std::vector<int64_t> h_arc_off(arc_offsets, arc_offsets + total_states + 1);
  std::vector<int32_t> h_next(arc_nextstate, arc_nextstate + total_arcs);
  std::vector<float> h_g(arc_graph, arc_graph + total_arcs), h_a(arc_acoustic, arc_acoustic + total_arcs),
      h_fin(state_final, state_final + total_states);
#define UP(dev, host) do { rc = dev.Upload(host, st); if (rc) return rc; } while (0)
  UP(d_descs, descs); UP(d_arc_off, h_arc_off); UP(d_next, h_next); UP(d_g, h_g); UP(d_a, h_a);
  UP(d_fin, h_fin); UP(d_level_off, level_off); UP(d_level_states, level_states); UP(d_in_off, in_off);
  UP(d_in_arc, in_arc); UP(d_in_src, in_src); UP(d_final_list, final_list);
#undef UP
  if (d_post.Alloc(total_arcs) || d_alpha.Alloc(total_states) || d_beta.Alloc(total_states) ||
      d_tot.Alloc(n_lats) || d_ac.Alloc(n_lats))
    return KH_ENOMEM;
  const double min_log_diff = log(DBL_EPSILON);  // kMinLogDiffDouble kaldi-math.h:120
  hipLaunchKernelGGL(ForwardBackwardKernel, dim3(n_lats), dim3(kThreads), 0, st, d_descs.p,
                     d_arc_off.p, d_next.p, d_g.p, d_a.p, d_fin.p, d_level_off.p, d_level_states.p,
                     d_in_off.p, d_in_arc.p, d_in_src.p, d_final_list.p, d_alpha.p, d_beta.p,
                     d_post.p, d_tot.p, d_ac.p, min_log_diff);
  KH_LAUNCH_CHECK();
  if (arc_post)
    KH_HIP(hipMemcpyAsync(arc_post, d_post.p, sizeof(float) * total_arcs, hipMemcpyDeviceToHost, st));
  if (tot_like)
    KH_HIP(hipMemcpyAsync(tot_like, d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  if (acoustic_like_sum)
    KH_HIP(hipMemcpyAsync(acoustic_like_sum, d_ac.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  return KH_OK;
}
