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
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <functional>
#include <limits>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <type_traits>

#include "kh_common.h"
#include "kh_logadd.h"

using namespace kh;

namespace {

constexpr int kThreads = 256;

struct LatDesc {
  int32_t state_b, n_states;   // range in the concatenated state arrays
  int64_t arc_b;               // first arc
  int32_t n_levels, level_b;   // range in level_off (n_levels + 1 entries)
  int32_t final_b, n_final;    // range in final_list
  int32_t max_time, n_reached; // largest state time; states the level sweep reached
};

// LogAddD: base/kaldi-math.h:178-195 (double), kh_logadd.h

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

// ComputeLatticeAlphasAndBetas :412-463 (LogAddOrMax :395-410): alpha, beta and the
// forward total of every lattice.
__global__ void __launch_bounds__(kThreads)
AlphaBetaKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off,
                const int32_t *__restrict__ arc_next, const float *__restrict__ arc_g,
                const float *__restrict__ arc_a, const float *__restrict__ state_final,
                const int32_t *__restrict__ level_off, const int32_t *__restrict__ level_states,
                const int64_t *__restrict__ in_off, const int64_t *__restrict__ in_arc,
                const int32_t *__restrict__ in_src, const int32_t *__restrict__ final_list,
                double *__restrict__ alpha, double *__restrict__ beta, double *__restrict__ tot_forward,
                int viterbi, double min_log_diff) {
  const LatDesc L = lats[blockIdx.x];
  const double kLogZero = -INFINITY;
  double *al = alpha + L.state_b, *be = beta + L.state_b;
  const float *fin = state_final + L.state_b;
  const int32_t *lstates = level_states + L.state_b;
  const int32_t *loff = level_off + L.level_b;
  for (int lv = 0; lv < L.n_levels; lv++) {
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double a = (s == 0) ? 0.0 : kLogZero;
      for (int64_t j = in_off[L.state_b + s]; j < in_off[L.state_b + s + 1]; j++) {
        const int64_t arc = in_arc[j];
        const double v = al[in_src[j]] - static_cast<double>(arc_g[arc] + arc_a[arc]);
        a = viterbi ? fmax(a, v) : LogAddD(a, v, min_log_diff);
      }
      al[s] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double tot = kLogZero;
    for (int k = 0; k < L.n_final; k++) {
      const int s = final_list[L.final_b + k];
      const double final_like = al[s] - static_cast<double>(fin[s] + 0.0f);
      tot = viterbi ? fmax(tot, final_like) : LogAddD(tot, final_like, min_log_diff);
    }
    tot_forward[blockIdx.x] = tot;
  }
  for (int lv = L.n_levels - 1; lv >= 0; lv--) {
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double this_beta = -static_cast<double>(fin[s] + 0.0f);
      for (int64_t arc = arc_off[L.state_b + s]; arc < arc_off[L.state_b + s + 1]; arc++) {
        const double arc_beta = be[arc_next[arc]] - static_cast<double>(arc_g[arc] + arc_a[arc]);
        this_beta = viterbi ? fmax(this_beta, arc_beta) : LogAddD(this_beta, arc_beta, min_log_diff);
      }
      be[s] = this_beta;
    }
    __syncthreads();
  }
}

struct MpeArgs {
  const int32_t *tid2phone, *tid2pdf, *sil, *num_ali, *ali_off, *times, *ilabel;
  int n_sil, is_mpfe, one_silence_class;
};

// frame accuracy of an arc (:820-842)
__device__ __forceinline__ double FrameAcc(const MpeArgs &m, int lat, int state_global, int64_t arc) {
  const int il = m.ilabel[arc];
  if (il == 0) return 0.0;
  const int ref_tid = m.num_ali[m.ali_off[lat] + m.times[state_global]];
  const int phone = m.tid2phone[il], ref_phone = m.tid2phone[ref_tid];
  bool phone_is_sil = false, ref_is_sil = false;
  for (int i = 0; i < m.n_sil; i++) {  // short sorted list
    phone_is_sil |= m.sil[i] == phone;
    ref_is_sil |= m.sil[i] == ref_phone;
  }
  const bool both_sil = phone_is_sil && ref_is_sil;
  if (!m.is_mpfe) {
    const int pdf = m.tid2pdf[il], ref_pdf = m.tid2pdf[ref_tid];
    if (!m.one_silence_class) return (pdf == ref_pdf && !phone_is_sil) ? 1.0 : 0.0;
    return (pdf == ref_pdf || both_sil) ? 1.0 : 0.0;
  }
  if (!m.one_silence_class) return (phone == ref_phone && !phone_is_sil) ? 1.0 : 0.0;
  return (phone == ref_phone || both_sil) ? 1.0 : 0.0;
}

// LatticeForwardBackwardMpeVariants :740-919 after AlphaBetaKernel: the second forward
// pass (alpha_smbr, expected accuracy) and the second backward pass (beta_smbr,
// posterior_smbr per arc).  Sums run in the reference's order: a state adds its
// incoming arcs by ascending (source, arc), its outgoing arcs by ascending arc.
__global__ void __launch_bounds__(kThreads)
MpeKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off,
          const int32_t *__restrict__ arc_next, const float *__restrict__ arc_g,
          const float *__restrict__ arc_a, const float *__restrict__ state_final,
          const int32_t *__restrict__ level_off, const int32_t *__restrict__ level_states,
          const int64_t *__restrict__ in_off, const int64_t *__restrict__ in_arc,
          const int32_t *__restrict__ in_src, const int32_t *__restrict__ final_list,
          const double *__restrict__ alpha, const double *__restrict__ beta,
          const double *__restrict__ tot_forward, double *__restrict__ alpha_smbr,
          double *__restrict__ beta_smbr, MpeArgs m, float *__restrict__ arc_post,
          double *__restrict__ tot_score, double *__restrict__ tot_backward_score) {
  __shared__ double s_score;
  const LatDesc L = lats[blockIdx.x];
  const double *al = alpha + L.state_b, *be = beta + L.state_b;
  double *as = alpha_smbr + L.state_b, *bs = beta_smbr + L.state_b;
  const float *fin = state_final + L.state_b;
  const int32_t *lstates = level_states + L.state_b;
  const int32_t *loff = level_off + L.level_b;
  const double tot_forward_prob = tot_forward[blockIdx.x];
  for (int lv = 0; lv < L.n_levels; lv++) {  // :813-846
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double acc = 0.0;
      for (int64_t j = in_off[L.state_b + s]; j < in_off[L.state_b + s + 1]; j++) {
        const int64_t arc = in_arc[j];
        const int src = in_src[j];
        const double arc_like = -static_cast<double>(arc_g[arc] + arc_a[arc]);
        const double frame_acc = FrameAcc(m, blockIdx.x, L.state_b + src, arc);
        const double arc_scale = exp(al[src] + arc_like - al[s]);
        acc += arc_scale * (as[src] + frame_acc);
      }
      as[s] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // :847-854
    double score = 0.0;
    for (int k = 0; k < L.n_final; k++) {
      const int s = final_list[L.final_b + k];
      const double final_like = al[s] - static_cast<double>(fin[s] + 0.0f);
      score += exp(final_like - tot_forward_prob) * as[s];
    }
    s_score = score;
    tot_score[blockIdx.x] = score;
  }
  __syncthreads();
  const double tot_forward_score = s_score;
  for (int lv = L.n_levels - 1; lv >= 0; lv--) {  // :857-903
    for (int k = loff[lv] + threadIdx.x; k < loff[lv + 1]; k += kThreads) {
      const int s = lstates[k];
      double b = 0.0;
      for (int64_t arc = arc_off[L.state_b + s]; arc < arc_off[L.state_b + s + 1]; arc++) {
        const int nxt = arc_next[arc];
        const double arc_like = -static_cast<double>(arc_g[arc] + arc_a[arc]), arc_beta = be[nxt] + arc_like;
        const double frame_acc = FrameAcc(m, blockIdx.x, L.state_b + s, arc);
        double arc_scale = exp(be[nxt] + arc_like - be[s]);
        if (arc_scale != arc_scale) arc_scale = 0;  // KALDI_ISNAN :890
        b += arc_scale * (bs[nxt] + frame_acc);
        float post = 0.0f;
        if (m.ilabel[arc] != 0) {
          const double posterior = exp(al[s] + arc_beta - tot_forward_prob);
          const double acc_diff = as[s] + frame_acc + bs[nxt] - tot_forward_score;
          post = static_cast<float>(posterior * acc_diff);
        }
        arc_post[arc] = post;
      }
      bs[s] = b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) tot_backward_score[blockIdx.x] = bs[0];
}

// RescoreLattice :1307-1358: arcs with a transition-id get -log_like added to their
// acoustic cost.  One lane per (state, its arcs).
__global__ void RescoreKernel(int total_states, const int64_t *__restrict__ arc_off, const int32_t *__restrict__ ilabel,
                              const int32_t *__restrict__ times, const int32_t *__restrict__ state_lat,
                              const int32_t *__restrict__ ll_row_off, float *__restrict__ arc_a,
                              const float *__restrict__ loglikes, int ll_stride, const int32_t *__restrict__ tid2pdf) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total_states; s += gridDim.x * blockDim.x) {
    const int t = times[s];
    if (t < 0) continue;
    const float *row = loglikes + static_cast<size_t>(ll_row_off[state_lat[s]] + t) * ll_stride;
    for (int64_t a = arc_off[s]; a < arc_off[s + 1]; a++) {
      const int il = ilabel[a];
      if (il != 0) arc_a[a] = -row[tid2pdf ? tid2pdf[il] : il - 1] + arc_a[a];  // :1349-1350
    }
  }
}

// CuMatrix::CompObjfAndDeriv cu-matrix.cc:1198-1248 (_cuda_comp_obj_deriv cu-kernels.cu:997-1035):
// objf = sum w log(output(r, c)), weight = sum w, deriv(r, c) += w / output(r, c).
// Duplicated (r, c) pairs are summed (atomicAdd; the reference's kernel races on them).
__global__ void __launch_bounds__(kThreads)
CompObjfKernel(int n, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
               const float *__restrict__ weights, const float *__restrict__ output, int out_stride,
               float *__restrict__ deriv, int deriv_stride, double *__restrict__ sums) {
  __shared__ double s_red[2][kThreads / 64];
  double objf = 0.0, wsum = 0.0;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const float w = weights[i], p = output[static_cast<size_t>(rows[i]) * out_stride + cols[i]];
    objf += static_cast<double>(w * logf(p));
    wsum += static_cast<double>(w);
    atomicAdd(&deriv[static_cast<size_t>(rows[i]) * deriv_stride + cols[i]], w / p);
  }
  objf = kh_wave_sum_d(objf);
  wsum = kh_wave_sum_d(wsum);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = objf; s_red[1][threadIdx.x >> 6] = wsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kThreads / 64; i++) { a += s_red[0][i]; b += s_red[1][i]; }
    atomicAdd(&sums[0], a);
    atomicAdd(&sums[1], b);
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

// ---------------------------------------------------------------- device radix sort (64-bit keys, 32-bit payload)
// A stable least-significant-digit radix sort over chosen bit fields of the key, 8 bits per pass, written for this file's
// one use - the (row, pdf) order of a batch's lattice arcs, 1.4-11 M pairs, of whose 64 key bits only the pdf's ~13 and the
// row's ~20 are ever set.  (Rounds 3-4 called hipcub::DeviceRadixSort here; the repository's rule is that device code on
// the path is its own.)  Per pass three launches:
//   SortHistKernel     a block counts the digits of its tile of kSortTile consecutive pairs -> hist[digit][block];
//   SortScanKernel     block d scans row d of the table in place (exclusive) and leaves the row's total; one more block
//                      scans the 256 totals;
//   SortScatterKernel  a block counts its tile again per WAVE (a wave owns a contiguous quarter of the tile), turns
//                      (digit base + blocks before + waves before) into the wave's cursors, and walks its quarter 64 pairs
//                      at a time: the lanes that hold the same digit find each other with ballots (eight, one per digit
//                      bit), a lane's rank is the number of its peers below it, the first of them advances the cursor.
//                      Blocks, waves and tiles are visited in order, so equal keys keep their order: the sort is stable.
// Traffic per pass: 40 bytes per pair (keys three times - the third read hits L2 -, payload once, both written once).
constexpr int kSortThreads = 256, kSortWaves = kSortThreads / 64, kSortItems = 16, kSortTile = kSortThreads * kSortItems, kSortBins = 256;

__global__ void __launch_bounds__(kSortThreads)
SortHistKernel(const unsigned long long *__restrict__ keys, int64_t n, int shift, uint32_t mask, uint32_t *__restrict__ hist, int n_blocks) {
  __shared__ uint32_t h[kSortBins];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll 4
  for (int j = 0; j < kSortItems; j++) {
    const int64_t i = b0 + j * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[static_cast<uint32_t>(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  hist[static_cast<size_t>(threadIdx.x) * n_blocks + blockIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ uint32_t SortBlockExScan(uint32_t v, uint32_t *total, uint32_t *ws) {   // 256 threads; ws[kSortWaves]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();   // (ws of the previous call has been read)
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int k = 0; k < kSortWaves; k++) {
    before += k < w ? ws[k] : 0u;
    all += ws[k];
  }
  *total = all;
  return before + inc - v;
}

// gridDim.x = kSortBins + 1: block d < 256 scans row d in place; the last block waits for nobody - it is launched as a second
// call (rows first, totals after), see SortPairs64.
__global__ void __launch_bounds__(kSortThreads)
SortScanKernel(uint32_t *__restrict__ hist, int n_blocks, uint32_t *__restrict__ totals, int totals_only) {
  __shared__ uint32_t ws[kSortWaves];
  if (totals_only) {   // one block: exclusive scan of the 256 row totals
    uint32_t tot;
    const uint32_t ex = SortBlockExScan(totals[threadIdx.x], &tot, ws);
    totals[threadIdx.x] = ex;
    return;
  }
  uint32_t *row = hist + static_cast<size_t>(blockIdx.x) * n_blocks;
  uint32_t run = 0;
  for (int c0 = 0; c0 < n_blocks; c0 += kSortThreads) {   // (uniform)
    const int c = c0 + threadIdx.x;
    const uint32_t v = c < n_blocks ? row[c] : 0u;
    uint32_t tot;
    const uint32_t ex = SortBlockExScan(v, &tot, ws);
    if (c < n_blocks) row[c] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = run;
}

__global__ void __launch_bounds__(kSortThreads)
SortScatterKernel(const unsigned long long *__restrict__ kin, const int32_t *__restrict__ vin, unsigned long long *__restrict__ kout,
                  int32_t *__restrict__ vout, int64_t n, int shift, uint32_t mask, const uint32_t *__restrict__ hist,
                  const uint32_t *__restrict__ base, int n_blocks) {
  __shared__ uint32_t cur[kSortWaves][kSortBins];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < kSortWaves * kSortBins; k += kSortThreads) (&cur[0][0])[k] = 0u;
  __syncthreads();
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * kSortTile + static_cast<int64_t>(w) * (kSortTile / kSortWaves);   // the wave's quarter
  constexpr int kTiles = kSortTile / kSortWaves / 64;
  for (int t = 0; t < kTiles; t++) {
    const int64_t i = w0 + t * 64 + lane;
    if (i < n) atomicAdd(&cur[w][static_cast<uint32_t>(kin[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  {  // counts per wave -> the wave's first output slot per digit
    const int d = threadIdx.x;
    uint32_t at = base[d] + hist[static_cast<size_t>(d) * n_blocks + blockIdx.x];
#pragma unroll
    for (int k = 0; k < kSortWaves; k++) {
      const uint32_t c = cur[k][d];
      cur[k][d] = at;
      at += c;
    }
  }
  __syncthreads();
  for (int t = 0; t < kTiles; t++) {   // (uniform over the wave)
    const int64_t i = w0 + t * 64 + lane;
    const bool act = i < n;
    const unsigned long long key = act ? kin[i] : 0ull;
    const int32_t val = act ? vin[i] : 0;
    const uint32_t d = static_cast<uint32_t>(key >> shift) & mask;
    unsigned long long peers = __ballot(act);
    if (peers == 0ull) break;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const bool set = ((d >> bit) & 1u) != 0u;
      const unsigned long long m = __ballot(set);
      peers &= set ? m : ~m;
    }
    if (act) {
      const int rank = __popcll(peers & ((1ull << lane) - 1ull));
      const uint32_t at = cur[w][d];
      if (rank == 0) cur[w][d] = at + static_cast<uint32_t>(__popcll(peers));
      kout[at + rank] = key;
      vout[at + rank] = val;
    }
  }
}

// Sorts the n pairs (k0, v0) by the key bits [lo_bits) of the low word and [32, 32 + hi_bits) of the high word; the result is
// in (k1, v1) (returned through *in_second = 1) or back in (k0, v0).  work: kSortBins * n_blocks + kSortBins words.
int SortPairs64(unsigned long long *k0, int32_t *v0, unsigned long long *k1, int32_t *v1, int64_t n, int lo_bits, int hi_bits,
                uint32_t *work, hipStream_t st, int *in_second) {
  if (n <= 0) { *in_second = 0; return KH_OK; }   // nothing to sort: a 0-block launch is an error on HIP
  const int n_blocks = static_cast<int>((n + kSortTile - 1) / kSortTile);
  uint32_t *hist = work, *totals = work + static_cast<size_t>(kSortBins) * n_blocks;
  int cur = 0;
  for (int half = 0; half < 2; half++) {
    const int bits = half ? hi_bits : lo_bits;
    for (int sb = 0; sb < bits; sb += 8) {
      const int shift = half * 32 + sb;
      const uint32_t mask = (1u << std::min(8, bits - sb)) - 1u;
      const unsigned long long *kin = cur ? k1 : k0;
      const int32_t *vin = cur ? v1 : v0;
      hipLaunchKernelGGL(SortHistKernel, dim3(n_blocks), dim3(kSortThreads), 0, st, kin, n, shift, mask, hist, n_blocks);
      hipLaunchKernelGGL(SortScanKernel, dim3(kSortBins), dim3(kSortThreads), 0, st, hist, n_blocks, totals, 0);
      hipLaunchKernelGGL(SortScanKernel, dim3(1), dim3(kSortThreads), 0, st, hist, n_blocks, totals, 1);
      hipLaunchKernelGGL(SortScatterKernel, dim3(n_blocks), dim3(kSortThreads), 0, st, kin, vin, cur ? k0 : k1, cur ? v0 : v1, n, shift, mask,
                         hist, totals, n_blocks);
      KH_LAUNCH_CHECK();
      cur ^= 1;
    }
  }
  *in_second = cur;
  return KH_OK;
}

}  // namespace

// Preparation shared by the lattice sweeps, ON THE DEVICE (one workgroup per lattice):
// validation (top-sorted, consistent state times: LatticeStateTimes :36-67 and the
// "must be topologically sorted" checks :38-39,:285-286), dependency levels (Kahn's
// algorithm: a state joins the level after its last predecessor's = longest distance from
// a source), the incoming-arc CSR (ascending arc index per destination: the operand order
// of the reference's sequential sweep), the ascending list of final states.  The host only
// uploads the caller's arrays (round 1 built all of this on host threads and uploaded
// twice the bytes: 82 ms per 256-lattice batch for 2.2 ms of kernel).
namespace {

// Barrier that also waits for this wave's outstanding global stores: words initialised with
// plain stores are updated by L2 atomics of other waves right after it (hipcc's
// __syncthreads() is a workgroup-scope fence and does not wait for stores to reach L2).
__device__ __forceinline__ void PrepSync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ int PrepScan(int v, int *total, int *s_w) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  PrepSync();  // s_w of the previous call has been read
  if (lane == 63) s_w[w] = inc;
  PrepSync();
  int before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < kThreads / 64; i++) {
    const int t = s_w[i];
    before += i < w ? t : 0;
    all += t;
  }
  *total = all;
  return before + inc - v;
}

// The incoming lists (filled in arrival order) sorted by arc index, a wave per block of 64 consecutive states: the block's
// lists are contiguous - staged in LDS with coalesced loads, every ENTRY finds its place by counting the smaller arc indices
// of its list, coalesced-ish stores (PrepKernelDF does the same inside its state-time pass).  A block with more entries than
// the staging area holds: one lane per state, insertion sort in memory.
constexpr int kSortStage = 384;
__device__ __forceinline__ void SortIncomingLists(const int64_t *__restrict__ in_off, int64_t *__restrict__ in_arc, int32_t *__restrict__ in_src,
                                                  int sb, int ns, int64_t arc_b, int (*st_src)[kSortStage], int (*st_arc)[kSortStage],
                                                  int (*st_seg)[kSortStage]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int blk = wave; blk * 64 < ns; blk += kThreads / 64) {
    const int s = blk * 64 + lane;
    const bool active = s < ns;
    int64_t j = 0, ie = 0;
    if (active) { j = in_off[sb + s]; ie = in_off[sb + s + 1]; }
    const int64_t jb = in_off[sb + blk * 64], je = in_off[sb + min(ns, blk * 64 + 64)];
    if (je - jb <= kSortStage) {
      const int n_st = static_cast<int>(je - jb);
      for (int e2 = lane; e2 < n_st; e2 += 64) {
        st_src[wave][e2] = in_src[jb + e2];
        st_arc[wave][e2] = static_cast<int>(in_arc[jb + e2] - arc_b);
      }
      if (active) {
        const int b0 = static_cast<int>(j - jb), len = static_cast<int>(ie - j);
        for (int k = 0; k < len; k++) st_seg[wave][b0 + k] = b0 | (len << 16);
      }
      constexpr int kPer = kSortStage / 64;
      int pos[kPer], ksrc[kPer], karc[kPer];
#pragma unroll
      for (int r = 0; r < kPer; r++) {
        const int e2 = lane + 64 * r;
        pos[r] = -1;
        ksrc[r] = 0;
        karc[r] = 0;
        if (e2 < n_st) {
          const int seg = st_seg[wave][e2], b0 = seg & 0xffff, len = seg >> 16;
          karc[r] = st_arc[wave][e2];
          ksrc[r] = st_src[wave][e2];
          int rank = 0;
          for (int k = 0; k < len; k++) rank += st_arc[wave][b0 + k] < karc[r] ? 1 : 0;
          pos[r] = b0 + rank;
        }
      }
#pragma unroll
      for (int r = 0; r < kPer; r++) {
        if (pos[r] < 0) continue;
        in_arc[jb + pos[r]] = arc_b + karc[r];
        in_src[jb + pos[r]] = ksrc[r];
      }
    } else if (active) {
      for (int64_t i = j + 1; i < ie; i++) {
        const int64_t ka = in_arc[i];
        const int32_t ks = in_src[i];
        int64_t q = i - 1;
        while (q >= j && in_arc[q] > ka) {
          in_arc[q + 1] = in_arc[q];
          in_src[q + 1] = in_src[q];
          q--;
        }
        in_arc[q + 1] = ka;
        in_src[q + 1] = ks;
      }
    }
  }
}

constexpr int kPrepLdsStates = 5120;   // 60 KB of LDS for the three work arrays
__global__ void __launch_bounds__(kThreads)
PrepKernel(const int32_t *__restrict__ lat_off, const int64_t *__restrict__ arc_off, const int32_t *__restrict__ il,
           const int32_t *__restrict__ next, const float *__restrict__ fin, LatDesc *__restrict__ descs,
           int32_t *times, int32_t *__restrict__ level_off, int32_t *__restrict__ level_states,
           int64_t *__restrict__ in_off, int64_t *__restrict__ in_arc, int32_t *__restrict__ in_src,
           int32_t *__restrict__ final_list, int32_t *indeg, int32_t *fill,
           int32_t *__restrict__ err) {
  __shared__ int s_w[kThreads / 64];
  __shared__ int s_err[4];
  __shared__ int s_qend, s_maxt;
  // The three per-state work arrays (state times, in-degrees, fill counters) take ~1.3 M atomic updates per 256-lattice
  // batch; in device memory every one of them is a read-modify-write at the memory side (4.4 ms for 256 lattices, more
  // than both sweeps).  A lattice of up to kPrepLdsStates states keeps them in LDS; the times are copied out at the end.
  __shared__ int s_work[3 * kPrepLdsStates];
  const int l = blockIdx.x, t = threadIdx.x;
  const int sb = lat_off[l], ns = lat_off[l + 1] - sb;
  const int64_t arc_b = arc_off[sb];
  const bool in_lds = ns <= kPrepLdsStates;
  // (indexed with the lattice-local state from here on)
  int32_t *const tw = in_lds ? s_work : times + sb;
  int32_t *const dw = in_lds ? s_work + kPrepLdsStates : indeg + sb;
  int32_t *const fw = in_lds ? s_work + 2 * kPrepLdsStates : fill + sb;
  if (t == 0) { s_err[0] = 0; s_err[1] = 0; s_err[2] = 0; s_err[3] = 0; s_maxt = 0; }
  for (int s = t; s < ns; s += kThreads) {
    tw[s] = s == 0 ? 0 : -1;
    dw[s] = 0;
    fw[s] = 0;
  }
  PrepSync();
  // ---- in-degrees + "input lattice must be topologically sorted"
  for (int s = t; s < ns; s += kThreads)
    for (int64_t a = arc_off[sb + s]; a < arc_off[sb + s + 1]; a++) {
      const int nxt = next[a];
      if (nxt <= s || nxt >= ns) {
        if (atomicCAS(&s_err[0], 0, 1) == 0) { s_err[1] = s; s_err[2] = nxt; s_err[3] = static_cast<int>(a - arc_b); }
      } else {
        atomicAdd(&dw[nxt], 1);
      }
    }
  PrepSync();
  if (s_err[0] != 0) {
    if (t < 4) err[4 * l + t] = s_err[t];
    return;
  }
  // ---- incoming-arc offsets (exclusive scan of the in-degrees), finals and sources (ascending)
  int carry = 0, nf = 0, nq = 0;
  for (int base = 0; base < ns; base += kThreads) {
    const int s = base + t;
    const int deg = s < ns ? __hip_atomic_load(&dw[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int is_final = (s < ns && fin[sb + s] != INFINITY) ? 1 : 0;
    const int is_root = (s < ns && deg == 0) ? 1 : 0;
    int tot;
    const int off = PrepScan(deg, &tot, s_w);
    if (s < ns) in_off[sb + s] = arc_b + carry + off;
    carry += tot;
    const int foff = PrepScan(is_final, &tot, s_w);
    if (is_final) final_list[sb + nf + foff] = s;
    nf += tot;
    const int qoff = PrepScan(is_root, &tot, s_w);
    if (is_root) level_states[sb + nq + qoff] = s;
    nq += tot;
  }
  if (t == 0) in_off[sb + ns] = arc_b + carry;
  PrepSync();
  // ---- incoming lists (filled in arrival order, then sorted by arc index)
  for (int s = t; s < ns; s += kThreads)
    for (int64_t a = arc_off[sb + s]; a < arc_off[sb + s + 1]; a++) {
      const int nxt = next[a];
      const int64_t pos = in_off[sb + nxt] + atomicAdd(&fw[nxt], 1);
      in_arc[pos] = a;
      in_src[pos] = s;
    }
  PrepSync();
  {
    __shared__ int so_src[kThreads / 64][kSortStage], so_arc[kThreads / 64][kSortStage], so_seg[kThreads / 64][kSortStage];
    SortIncomingLists(in_off, in_arc, in_src, sb, ns, arc_b, so_src, so_arc, so_seg);
  }
  // ---- levels (Kahn) + LatticeStateTimes
  const int lvb = sb + l;  // level_off has n_states + 1 entries per lattice
  int lb = 0, le = nq, lv = 0;
  if (t == 0) { s_qend = nq; level_off[lvb] = 0; }
  int my_maxt = 0;
  for (;;) {
    PrepSync();
    for (int k = lb + t; k < le; k += kThreads) {
      const int s = level_states[sb + k];
      const int ts = __hip_atomic_load(&tw[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      my_maxt = ts > my_maxt ? ts : my_maxt;
      for (int64_t a = arc_off[sb + s]; a < arc_off[sb + s + 1]; a++) {
        const int nxt = next[a];
        if (ts >= 0) {
          const int want = ts + (il[a] != 0 ? 1 : 0);
          const int old = atomicCAS(&tw[nxt], -1, want);
          if (old != -1 && old != want && atomicCAS(&s_err[0], 0, 2) == 0) { s_err[1] = nxt; s_err[2] = old; s_err[3] = want; }
        }
        if (atomicSub(&dw[nxt], 1) == 1) level_states[sb + atomicAdd(&s_qend, 1)] = nxt;
      }
    }
    PrepSync();
    const int new_end = s_qend;
    lv++;
    if (t == 0) level_off[lvb + lv] = le;
    if (new_end == le) break;
    lb = le;
    le = new_end;
  }
  atomicMax(&s_maxt, my_maxt);
  PrepSync();
  if (in_lds)
    for (int s = t; s < ns; s += kThreads) times[sb + s] = tw[s];
  if (t < 4) err[4 * l + t] = s_err[t];
  if (t == 0) {
    LatDesc d;
    d.state_b = sb;
    d.n_states = ns;
    d.arc_b = arc_b;
    d.n_levels = lv;
    d.level_b = lvb;
    d.final_b = sb;
    d.n_final = nf;
    d.max_time = s_maxt;
    d.n_reached = le;
    descs[l] = d;
  }
}


constexpr int kWin = 4096;   // states whose value the dataflow sweeps keep in their LDS window
static_assert((kWin / 64) % (kThreads / 64) == 0, "a window slot is rewritten by the wave that wrote it");
// KH_LATTICE_PROFILE=1: shader-clock stamps of thread 0 of every workgroup (kProf instantiations; [16] words per lattice)
constexpr int kLatProf = 16;
#define LAT_STAMP(k) do { if (kProf && threadIdx.x == 0) { const long long now_ = clock64(); prof[blockIdx.x * kLatProf + (k)] += now_ - t_prev_; t_prev_ = now_; } } while (0)
// lds_states ints of dynamic LDS: the per-state counter of a lattice of up to that many states (in-degrees, then - the
// in-degrees are consumed by the scan - the fill counters of the incoming lists); win_slots 8-byte words behind it: the
// state-time window.
template <bool kProf>
__global__ void __launch_bounds__(kThreads)
PrepKernelDF(const int32_t *__restrict__ lat_off, const int64_t *__restrict__ arc_off, const int32_t *__restrict__ il,
           const int32_t *__restrict__ next, const float *__restrict__ fin, LatDesc *__restrict__ descs,
           int32_t *times, int32_t *__restrict__ level_off, int32_t *__restrict__ level_states,
           int64_t *__restrict__ in_off, int64_t *__restrict__ in_arc, int32_t *__restrict__ in_src,
           int32_t *__restrict__ final_list, int32_t *indeg, int32_t *fill,
           int32_t *__restrict__ err, long long *__restrict__ prof, int lds_states, int win_slots) {
  __shared__ int s_w[kThreads / 64];
  __shared__ int s_err[4];
  __shared__ int s_qend, s_maxt;
  // The per-state work arrays (in-degrees, fill counters) take ~1.3 M atomic updates per 256-lattice batch; in device
  // memory every one of them is a read-modify-write at the memory side (4.4 ms for 256 lattices, more than both sweeps).
  // A lattice of up to lds_states states keeps them in LDS.
  extern __shared__ unsigned long long dyn_prep[];   // [win_slots] window words, then [lds_states] ints
  unsigned long long *const w_tt = dyn_prep;
  int *const s_work = reinterpret_cast<int *>(dyn_prep + win_slots);
  long long t_prev_ = kProf ? clock64() : 0;
  const int l = blockIdx.x, t = threadIdx.x;
  const int sb = lat_off[l], ns = lat_off[l + 1] - sb;
  const int64_t arc_b = arc_off[sb];
  const bool in_lds = ns <= lds_states;
  // (indexed with the lattice-local state from here on)
  int32_t *const tw = times + sb;   // (only initialised here: the dataflow pass below writes the times)
  int32_t *const dw = in_lds ? s_work : indeg + sb;
  int32_t *const fw = in_lds ? s_work : fill + sb;   // (in LDS: the same words, zeroed again once the scan has read the in-degrees)
  if (t == 0) { s_err[0] = 0; s_err[1] = 0; s_err[2] = 0; s_err[3] = 0; s_maxt = 0; }
  for (int s = t; s < ns; s += kThreads) {
    tw[s] = s == 0 ? 0 : -1;
    dw[s] = 0;
    if (!in_lds) fw[s] = 0;
  }
  PrepSync();
  LAT_STAMP(0);
  // ---- in-degrees + "input lattice must be topologically sorted"
  for (int s = t; s < ns; s += kThreads)
    for (int64_t a = arc_off[sb + s]; a < arc_off[sb + s + 1]; a++) {
      const int nxt = next[a];
      if (nxt <= s || nxt >= ns) {
        if (atomicCAS(&s_err[0], 0, 1) == 0) { s_err[1] = s; s_err[2] = nxt; s_err[3] = static_cast<int>(a - arc_b); }
      } else {
        atomicAdd(&dw[nxt], 1);
      }
    }
  PrepSync();
  if (s_err[0] != 0) {
    if (t < 4) err[4 * l + t] = s_err[t];
    return;
  }
  LAT_STAMP(1);
  // ---- incoming-arc offsets (exclusive scan of the in-degrees), finals and sources (ascending)
  int carry = 0, nf = 0, nq = 0;
  for (int base = 0; base < ns; base += kThreads) {
    const int s = base + t;
    const int deg = s < ns ? __hip_atomic_load(&dw[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int is_final = (s < ns && fin[sb + s] != INFINITY) ? 1 : 0;
    const int is_root = (s < ns && deg == 0) ? 1 : 0;
    int tot;
    const int off = PrepScan(deg, &tot, s_w);
    if (s < ns) in_off[sb + s] = arc_b + carry + off;
    carry += tot;
    const int foff = PrepScan(is_final, &tot, s_w);
    if (is_final) final_list[sb + nf + foff] = s;
    nf += tot;
    const int qoff = PrepScan(is_root, &tot, s_w);
    if (is_root) level_states[sb + nq + qoff] = s;
    nq += tot;
  }
  if (t == 0) in_off[sb + ns] = arc_b + carry;
  PrepSync();
  if (in_lds) {
    for (int s = t; s < ns; s += kThreads) fw[s] = 0;
    PrepSync();
  }
  LAT_STAMP(2);
  // ---- incoming lists (filled in arrival order, then sorted by arc index)
  for (int s = t; s < ns; s += kThreads)
    for (int64_t a = arc_off[sb + s]; a < arc_off[sb + s + 1]; a++) {
      const int nxt = next[a];
      const int64_t pos = in_off[sb + nxt] + atomicAdd(&fw[nxt], 1);
      in_arc[pos] = a;
      in_src[pos] = s;
    }
  PrepSync();
  LAT_STAMP(3);
  // ---- LatticeStateTimes (:36-67) without dependency levels: the states in INDEX order (the lattice is top-sorted), a
  // wave per block of 64 consecutive states, a state waits until the predecessor it reads next has published its time.
  // Sliding LDS window, ONE 8-byte word per slot = state << 32 | time (all ones: nothing yet): an operand is one LDS read, a
  // result one LDS store; a slot that holds a later state means the wanted one has left the window - its time is read
  // from memory, where it was stored before it was published (a slot is rewritten by the state win_slots further on, which
  // belongs to the same wave, and the wave stores and waits at the end of every block).  One operand per lane and pass.
  // The incoming lists were filled in arrival order; the sweeps fold a state's arcs in ascending arc order (the reference's
  // sequential sweep), so every list is sorted by arc index - in the block's LDS staging area (the block's lists are
  // contiguous: coalesced loads, a lane sorts its own state's few entries at LDS latency, coalesced stores; one thread per
  // state sorting in device memory took 3.7 M of the 6.4 M cycles of a 16 k-state lattice's preparation).
  constexpr int kTStage = 384;
  __shared__ int st_src[kThreads / 64][kTStage];
  __shared__ int st_arc[kThreads / 64][kTStage];   // arc index - the lattice's first arc
  __shared__ int st_seg[kThreads / 64][kTStage];   // the entry's list: first entry | length << 16
  __shared__ signed char st_inc[kThreads / 64][kTStage];
  for (int i = t; i < win_slots; i += kThreads) w_tt[i] = ~0ull;
  PrepSync();
  LAT_STAMP(4);
  int my_maxt = 0;
  {
    const int lane = t & 63, wave = t >> 6;
    const int wmask = win_slots - 1;
    for (int blk = wave; blk * 64 < ns; blk += kThreads / 64) {
      const int s = blk * 64 + lane;
      const bool active = s < ns;
      int64_t j = 0, ie = 0;
      if (active) { j = in_off[sb + s]; ie = in_off[sb + s + 1]; }
      int tval = (active && s == 0) ? 0 : -1;
      bool done = !active;
      // (the block's entries staged in LDS before anything is waited for: see ForwardBackwardDFKernel)
      const int64_t jb = in_off[sb + blk * 64], je = in_off[sb + min(ns, blk * 64 + 64)];
      const int n_st = static_cast<int>(min<int64_t>(je - jb, kTStage));
      if (je - jb <= kTStage) {
        for (int e2 = lane; e2 < n_st; e2 += 64) {
          st_src[wave][e2] = in_src[jb + e2];
          st_arc[wave][e2] = static_cast<int>(in_arc[jb + e2] - arc_b);
        }
        // every ENTRY finds its place in its state's list by counting the smaller arc indices of that list (a lane per
        // entry: the longest list of the block costs its length in LDS reads, not its square in dependent round trips)
        if (active) {
          const int b0 = static_cast<int>(j - jb), len = static_cast<int>(ie - j);
          for (int k = 0; k < len; k++) st_seg[wave][b0 + k] = b0 | (len << 16);
        }
        constexpr int kPer = kTStage / 64;
        int pos[kPer], ksrc[kPer], karc[kPer];
#pragma unroll
        for (int r = 0; r < kPer; r++) {
          const int e2 = lane + 64 * r;
          pos[r] = -1;
          ksrc[r] = 0;
          karc[r] = 0;
          if (e2 < n_st) {
            const int seg = st_seg[wave][e2], b0 = seg & 0xffff, len = seg >> 16;
            karc[r] = st_arc[wave][e2];
            ksrc[r] = st_src[wave][e2];
            int rank = 0;
            for (int k = 0; k < len; k++) rank += st_arc[wave][b0 + k] < karc[r] ? 1 : 0;
            pos[r] = b0 + rank;
          }
        }
#pragma unroll
        for (int r = 0; r < kPer; r++) {
          if (pos[r] < 0) continue;
          const int64_t arc = arc_b + karc[r];
          st_src[wave][pos[r]] = ksrc[r];
          in_arc[jb + pos[r]] = arc;
          in_src[jb + pos[r]] = ksrc[r];
          st_inc[wave][pos[r]] = il[arc] != 0 ? 1 : 0;
        }
      } else {   // more entries than the staging area holds: every lane sorts its state's list in memory
        if (active) {
          for (int64_t i = j + 1; i < ie; i++) {
            const int64_t ka = in_arc[i];
            const int32_t ks = in_src[i];
            int64_t q = i - 1;
            while (q >= j && in_arc[q] > ka) {
              in_arc[q + 1] = in_arc[q];
              in_src[q + 1] = in_src[q];
              q--;
            }
            in_arc[q + 1] = ka;
            in_src[q + 1] = ks;
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int e2 = lane; e2 < n_st; e2 += 64) {
          st_src[wave][e2] = in_src[jb + e2];
          st_inc[wave][e2] = il[in_arc[jb + e2]] != 0 ? 1 : 0;
        }
      }
      int src = 0, inc = 0;
      auto fetch = [&](int64_t jj) {
        const int64_t e2 = jj - jb;
        if (e2 < kTStage) { src = st_src[wave][e2]; inc = st_inc[wave][e2]; }
        else { src = in_src[jj]; inc = il[in_arc[jj]] != 0 ? 1 : 0; }
      };
      if (!done && j < ie) fetch(j);
      while (__ballot(!done) != 0ull) {
        if (!done) {
          if (j < ie) {
            const unsigned long long wv = __hip_atomic_load(&w_tt[src & wmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int tag = static_cast<int>(wv >> 32);   // (-1: empty)
            bool have = false;
            int tp = -1;
            if (tag == src) { tp = static_cast<int>(static_cast<uint32_t>(wv)); have = true; }
            else if (tag > src) { tp = __hip_atomic_load(&times[sb + src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); have = true; }
            if (have) {
              if (tp >= 0) {
                const int want = tp + inc;
                if (tval == -1) tval = want;
                else if (tval != want && atomicCAS(&s_err[0], 0, 2) == 0) { s_err[1] = s; s_err[2] = tval; s_err[3] = want; }
              }
              j++;
              if (j < ie) fetch(j);
            }
          }
          if (j >= ie) {
            __hip_atomic_store(&w_tt[s & wmask], (static_cast<unsigned long long>(static_cast<uint32_t>(s)) << 32) | static_cast<uint32_t>(tval),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (to memory at the end of the block: see WinPublish)
            my_maxt = tval > my_maxt ? tval : my_maxt;
            done = true;
          }
        }
      }
      if (active) __hip_atomic_store(&times[sb + s], tval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  const int lv = 0, le = ns, lvb = sb + l;   // (no levels: n_levels = 0 tells the level-based kernels apart)
  atomicMax(&s_maxt, my_maxt);
  PrepSync();
  LAT_STAMP(5);
  if (t < 4) err[4 * l + t] = s_err[t];
  if (t == 0) {
    LatDesc d;
    d.state_b = sb;
    d.n_states = ns;
    d.arc_b = arc_b;
    d.n_levels = lv;
    d.level_b = lvb;
    d.final_b = sb;
    d.n_final = nf;
    d.max_time = s_maxt;
    d.n_reached = le;
    descs[l] = d;
  }
}

}  // namespace

namespace {

// LatticeForwardBackward :272-354 WITHOUT dependency levels (PrepKernelDF batches).  The level-based kernel above spends
// its time on latency: ~420 levels per lattice, each a workgroup barrier plus a chain of four dependent loads from device
// memory (2.6 ms per 256-lattice batch for 2.8 M arc visits).  Here the states go in INDEX order - a top-sorted lattice's
// own valid schedule - a wave per block of 64 consecutive states; a state's first two incoming (forward) / outgoing
// (backward) arcs are fetched before it waits for anything, the alpha / beta values it needs come from a sliding LDS
// window of the last kWin states (tags say which state a slot holds; a value that has left the window is read from
// memory, where it was stored before it was published), and a state proceeds as soon as the operands it needs next are
// there: no barrier, no level structure, the dependent chain runs at LDS latency.  Operand order per state = ascending
// arc index, as in the reference's sequential sweep: results equal the level-based kernel's bit for bit.
// The window.  Two forms, chosen per workgroup: a lattice whose states all fit (n_states <= win_slots) uses the area as ONE
// double per state, "not there yet" = a NaN bit pattern no sweep produces (kWinEmpty; a NaN result with exactly these bits is
// published as the canonical quiet NaN): an operand is ONE 8-byte LDS read and a result one 8-byte LDS store, nothing is ever
// evicted and nothing is read from memory.  A larger lattice uses half the area as values and the other half as tags
// (which state a slot holds; -2 while it is rewritten): three dependent LDS operations per operand, and a value that has
// left the window is read from memory, where it was stored before it was published.
constexpr unsigned long long kWinEmpty = 0x7ff8dead00000001ull;
struct Win {
  double *val;
  int *tag;
  int mask;      // tagged form: slots - 1
  bool direct;
};
template <bool kDirect>
__device__ __forceinline__ bool WinRead(const Win &w, const double *mem, int idx, bool forward, double *out) {
  if (kDirect) {
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&w.val[idx]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (b == kWinEmpty) return false;
    *out = __longlong_as_double(static_cast<long long>(b));
    return true;
  }
  const int slot = idx & w.mask;
  const int tag1 = __hip_atomic_load(&w.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (tag1 == idx) {
    const double v = __hip_atomic_load(&w.val[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__hip_atomic_load(&w.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == idx) { *out = v; return true; }
    *out = __hip_atomic_load(&mem[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // evicted while we looked
    return true;
  }
  // a slot that holds a LATER state of the sweep's order means idx has been through the window already
  const bool evicted = tag1 >= 0 && (forward ? tag1 > idx : tag1 < idx);
  if (!evicted) return false;   // not computed yet (or the slot is being rewritten: look again)
  *out = __hip_atomic_load(&mem[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
// Publishes a value in the window (LDS only).  The value goes to memory when the wave has finished its block of states
// (the callers), NOT here: on gfx9 `vmcnt` counts stores, and the compiler's s_waitcnt vmcnt(0) in front of the polling
// loop's next load would put the store's round trip to device memory (~2 us) on every step of the dependent chain - that
// was 3 ms per 256-lattice batch, more than the arithmetic.  (Tagged form) in memory before anybody can find the slot
// evicted: a slot is rewritten by the state `slots` further on, which belongs to the SAME wave (slots / 64 is a multiple of
// the waves), and the wave stores and waits at the end of every block.
template <bool kDirect>
__device__ __forceinline__ void WinPublish(const Win &w, int idx, double v) {
  if (kDirect) {
    unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    if (b == kWinEmpty) b = 0x7ff8000000000000ull;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&w.val[idx]), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  const int slot = idx & w.mask;
  __hip_atomic_store(&w.tag[slot], -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_store(&w.val[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_store(&w.tag[slot], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void WinReset(const Win &w, int ns) {
  if (w.direct) {
    for (int i = threadIdx.x; i < ns; i += kThreads) reinterpret_cast<unsigned long long *>(w.val)[i] = kWinEmpty;
  } else {
    for (int i = threadIdx.x; i <= w.mask; i += kThreads) w.tag[i] = -1;
  }
}

// win_slots (a power of two >= 512) doubles of dynamic LDS: the window.  kStage: arcs of a 64-state block staged per wave.
// A lane takes ONE operand per pass of its wave's loop (a lane with six operands ready used to hold the wave's other lanes -
// among them the successors of states that had just been finished - for six LogAdds), and the staged words of its next
// operand are requested before the LogAdd of the current one.
template <bool kProf, int kStage>
__global__ void __launch_bounds__(kThreads)
ForwardBackwardDFKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off,
                        const int32_t *__restrict__ arc_next, const float *__restrict__ arc_g,
                        const float *__restrict__ arc_a, const float *__restrict__ state_final,
                        const int64_t *__restrict__ in_off, const int64_t *__restrict__ in_arc,
                        const int32_t *__restrict__ in_src, const int32_t *__restrict__ final_list,
                        double *alpha, double *beta, float *__restrict__ arc_post, double *__restrict__ tot_like,
                        double *__restrict__ ac_sum, double min_log_diff, long long *__restrict__ prof, int win_slots) {
  // kProf: wave 0 of every workgroup keeps, in registers, the shader cycles of [6] staging, [7] passes of its loop in which no
  // lane had its operand (waiting for other waves), [8] passes in which some lane folded an operand (LDS read + LogAdd
  // [+ posterior] + publish), [9] stores + wait at the end of a block; [10] / [11] the number of such passes, [12] blocks,
  // [13] the whole kernel, [14] the forward sweep - written once at the end
  long long t_prev_ = kProf ? clock64() : 0;
  const long long t_begin_ = t_prev_;
  long long pf_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FB_ADD(k, v) do { if (kProf) pf_[k] += (v); } while (0)
#define FB_STAMP(k) do { if (kProf) { const long long now_ = clock64(); pf_[k] += now_ - t_prev_; t_prev_ = now_; } } while (0)
  // A block's arcs are staged in LDS before the block waits for anything (they do not depend on alpha / beta): inside the
  // polling loop ONE lane's load from device memory would stall the whole wave for a round trip on every step of the
  // dependent chain (SIMT) - that, not the arithmetic, was the 3 ms of the first version of this kernel.
  constexpr int kWaves = kThreads / 64;
  extern __shared__ double dyn_win[];
  __shared__ double st_like[kWaves][kStage];
  __shared__ int st_idx[kWaves][kStage];     // forward: source state; backward: next state
  __shared__ float st_ac[kWaves][kStage];    // backward: acoustic cost, then the arc's posterior
  __shared__ double s_tot;
  __shared__ double s_red[kWaves];
  const LatDesc L = lats[blockIdx.x];
  const double kLogZero = -INFINITY;
  const int ns = L.n_states, sb = L.state_b, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *al = alpha + sb, *be = beta + sb;
  const float *fin = state_final + sb;
  Win w;
  w.direct = ns <= win_slots;
  w.val = dyn_win;
  w.tag = reinterpret_cast<int *>(dyn_win + (win_slots >> 1));
  w.mask = (win_slots >> 1) - 1;
  WinReset(w, ns);
  __syncthreads();
  // ---- forward :300-316
  for (int blk = wave; blk * 64 < ns; blk += kWaves) {
    const int s = blk * 64 + lane;
    const bool active = s < ns;
    int64_t ib = 0, ie = 0;
    if (active) { ib = in_off[sb + s]; ie = in_off[sb + s + 1]; }
    // the block's incoming entries are contiguous: [eb, ee)
    const int64_t eb = in_off[sb + blk * 64], ee = in_off[sb + min(ns, blk * 64 + 64)];
    const int n_st = static_cast<int>(min<int64_t>(ee - eb, kStage));
    for (int e = lane; e < n_st; e += 64) {
      const int64_t arc = in_arc[eb + e];
      st_idx[wave][e] = in_src[eb + e];
      st_like[wave][e] = -static_cast<double>(arc_g[arc] + arc_a[arc]);  // -ConvertToCost
    }
    FB_STAMP(6);
    FB_ADD(12, 1);
    double a = (active && s == 0) ? 0.0 : kLogZero;
    // (the loop in four forms - window form x "every arc of the block is staged" - so that the form that runs carries no
    // test of the other ones: a pass of a lone wave is paid in instructions issued)
    auto sweep = [&](auto direct_c, auto staged_c) {
      constexpr bool kDirect = decltype(direct_c)::value, kStaged = decltype(staged_c)::value;
      int64_t j = ib;
      bool done = !active;
      int src = 0;
      double like = 0.0;
      auto fetch = [&](int64_t jj) {
        const int64_t e = jj - eb;
        if (kStaged || e < kStage) { src = st_idx[wave][e]; like = st_like[wave][e]; }
        else { const int64_t arc = in_arc[jj]; src = in_src[jj]; like = -static_cast<double>(arc_g[arc] + arc_a[arc]); }
      };
      if (!done && j < ie) fetch(j);
      while (__ballot(!done) != 0ull) {
        bool folded = false;
        if (!done) {
          if (j < ie) {
            double av;
            if (WinRead<kDirect>(w, al, src, true, &av)) {
              const double term = av + like;
              j++;
              if (j < ie) fetch(j);
              a = LogAddD(a, term, min_log_diff);
              folded = true;
            }
          }
          if (j >= ie) {
            WinPublish<kDirect>(w, s, a);
            done = true;
          }
        }
        if (kProf) { if (__ballot(folded) != 0ull) { FB_STAMP(8); FB_ADD(10, 1); } else { FB_STAMP(7); FB_ADD(11, 1); } }
      }
    };
    const bool all_staged = ee - eb <= kStage;
    if (w.direct) { if (all_staged) sweep(std::true_type{}, std::true_type{}); else sweep(std::true_type{}, std::false_type{}); }
    else { if (all_staged) sweep(std::false_type{}, std::true_type{}); else sweep(std::false_type{}, std::false_type{}); }
    if (active) __hip_atomic_store(&al[s], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (see WinPublish)
    FB_STAMP(9);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = kLogZero;
    for (int k = 0; k < L.n_final; k++) {  // ascending state order, as the sweep meets them
      const int s = final_list[L.final_b + k];
      const double final_like = __hip_atomic_load(&al[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - static_cast<double>(fin[s] + 0.0f);
      tot = LogAddD(tot, final_like, min_log_diff);
    }
    s_tot = tot;
  }
  WinReset(w, ns);
  __syncthreads();
  if (kProf) { t_prev_ = clock64(); pf_[14] = t_prev_ - t_begin_; }   // the forward sweep
  const double tot_forward = s_tot;
  // ---- backward :317-344 (blocks of states from the end; inside a block the highest state is the first to be ready)
  double my_ac = 0.0;
  const int n_blk = (ns + 63) / 64;
  for (int bi = wave; bi < n_blk; bi += kWaves) {
    const int s0 = (n_blk - 1 - bi) * 64, s = s0 + lane;
    const bool active = s < ns;
    int64_t ab = 0, ae = 0;
    double as = 0.0, this_beta = 0.0;
    if (active) {
      ab = arc_off[sb + s];
      ae = arc_off[sb + s + 1];
      as = __hip_atomic_load(&al[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      this_beta = -static_cast<double>(fin[s] + 0.0f);
    }
    const int64_t eb = arc_off[sb + s0], ee = arc_off[sb + min(ns, s0 + 64)];   // the block's outgoing arcs: contiguous
    const int n_st = static_cast<int>(min<int64_t>(ee - eb, kStage));
    for (int e = lane; e < n_st; e += 64) {
      const float ac = arc_a[eb + e];
      st_idx[wave][e] = arc_next[eb + e];
      st_ac[wave][e] = ac;
      st_like[wave][e] = -static_cast<double>(arc_g[eb + e] + ac);
    }
    FB_STAMP(6);
    FB_ADD(12, 1);
    auto sweep = [&](auto direct_c, auto staged_c) {
      constexpr bool kDirect = decltype(direct_c)::value, kStaged = decltype(staged_c)::value;
      int64_t arc = ab;
      bool done = !active;
      int nx = 0;
      double like = 0.0;
      float ac = 0.f;
      auto fetch = [&](int64_t aa) {
        const int64_t e = aa - eb;
        if (kStaged || e < kStage) { nx = st_idx[wave][e]; like = st_like[wave][e]; if (!kStaged) ac = st_ac[wave][e]; }
        else { nx = arc_next[aa]; ac = arc_a[aa]; like = -static_cast<double>(arc_g[aa] + ac); }
      };
      if (!done && arc < ae) fetch(arc);
      while (__ballot(!done) != 0ull) {
        bool folded = false;
        if (!done) {
          if (arc < ae) {
            double bv;
            if (WinRead<kDirect>(w, be, nx, false, &bv)) {
              const double arc_beta = bv + like;
              const double acd = static_cast<double>(ac);
              const int64_t e = arc - eb;
              arc++;
              if (arc < ae) fetch(arc);
              this_beta = LogAddD(this_beta, arc_beta, min_log_diff);
              if (kStaged || e < kStage) {
                st_like[wave][e] = as + arc_beta;   // (the posterior's exp is not part of the chain: after the loop, every lane busy)
              } else {
                const double posterior = exp(as + arc_beta - tot_forward);
                arc_post[eb + e] = static_cast<float>(posterior);
                my_ac -= posterior * acd;
              }
              folded = true;
            }
          }
          if (arc >= ae) {
            WinPublish<kDirect>(w, s, this_beta);
            done = true;
          }
        }
        if (kProf) { if (__ballot(folded) != 0ull) { FB_STAMP(8); FB_ADD(10, 1); } else { FB_STAMP(7); FB_ADD(11, 1); } }
      }
    };
    const bool all_staged = ee - eb <= kStage;
    if (w.direct) { if (all_staged) sweep(std::true_type{}, std::true_type{}); else sweep(std::true_type{}, std::false_type{}); }
    else { if (all_staged) sweep(std::false_type{}, std::true_type{}); else sweep(std::false_type{}, std::false_type{}); }
    if (active) __hip_atomic_store(&be[s], this_beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int e = lane; e < n_st; e += 64) {   // arc posteriors :333-336 of the block's staged arcs
      const double posterior = exp(st_like[wave][e] - tot_forward);
      arc_post[eb + e] = static_cast<float>(posterior);
      my_ac -= posterior * static_cast<double>(st_ac[wave][e]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (see WinPublish)
    FB_STAMP(9);
  }
  my_ac = kh_wave_sum_d(my_ac);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = my_ac;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kWaves; i++) t += s_red[i];
    ac_sum[blockIdx.x] = t;
    tot_like[blockIdx.x] = __hip_atomic_load(&be[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tot_backward_prob :345
  }
  if (kProf && threadIdx.x == 0) {
    pf_[13] = clock64() - t_begin_;
    for (int k = 0; k < kLatProf; k++) prof[blockIdx.x * kLatProf + k] = pf_[k];
  }
#undef FB_ADD
#undef FB_STAMP
}
}  // namespace

// Where the time of the last lattice call of this thread went (kh_lattice_last_timings): upload of the caller's arrays,
// device preparation (PrepKernel + the descriptors' way back), sweeps, download.  HIP events on the library's stream.
struct LatTimer {
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float ms[4] = {0.f, 0.f, 0.f, 0.f};
  bool have[5] = {false, false, false, false, false};
  void Mark(int i, hipStream_t st) {
    if (!ev[i] && hipEventCreate(&ev[i]) != hipSuccess) { (void)hipGetLastError(); return; }
    have[i] = hipEventRecord(ev[i], st) == hipSuccess;
  }
  void Begin() { for (int i = 0; i < 5; i++) have[i] = false; for (int i = 0; i < 4; i++) ms[i] = 0.f; }
  void Finish() {   // (the stream has been synchronised)
    for (int i = 0; i < 4; i++)
      if (have[i] && have[i + 1] && hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) { (void)hipGetLastError(); ms[i] = 0.f; }
  }
};
static thread_local LatTimer g_lat_timer;

// KH_LATTICE_PROFILE=1: the kProf instantiations of the dataflow kernels; prints where thread 0 / wave 0 of the workgroups
// spent their shader cycles (means over the lattices of the call) to stderr.
static bool LatProfile() {
  static const bool on = getenv("KH_LATTICE_PROFILE") != nullptr && atoi(getenv("KH_LATTICE_PROFILE")) != 0;
  return on;
}
static void LatProfilePrint(const char *what, const DevArr<long long> &d_prof, int n_lats, hipStream_t st) {
  std::vector<long long> h(static_cast<size_t>(n_lats) * kLatProf);
  if (hipMemcpyAsync(h.data(), d_prof.p, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return; }
  double m[kLatProf] = {0};
  for (int l = 0; l < n_lats; l++)
    for (int k = 0; k < kLatProf; k++) m[k] += static_cast<double>(h[static_cast<size_t>(l) * kLatProf + k]) / n_lats;
  {  // the slowest workgroup sets the kernel's duration
    const bool prep = strcmp(what, "prep") == 0;
    long long worst = -1;
    int at = 0;
    for (int l = 0; l < n_lats; l++) {
      long long tot = 0;
      if (prep) for (int k = 0; k < 6; k++) tot += h[static_cast<size_t>(l) * kLatProf + k];
      else tot = h[static_cast<size_t>(l) * kLatProf + 13];
      if (tot > worst) { worst = tot; at = l; }
    }
    fprintf(stderr, "[kh_lattice profile] %s: slowest workgroup = lattice %d, %lld cycles [", what, at, worst);
    for (int k = 0; k < kLatProf; k++) fprintf(stderr, "%s%lld", k ? " " : "", h[static_cast<size_t>(at) * kLatProf + k]);
    fprintf(stderr, "]\n");
  }
  if (strcmp(what, "prep") == 0)
    fprintf(stderr, "[kh_lattice profile] PrepKernelDF, %d lattices, shader cycles of thread 0 (mean): init %.0f, in-degrees %.0f, "
            "scans %.0f, incoming lists %.0f, window reset %.0f, sort of the lists + state times (dataflow) %.0f\n", n_lats, m[0], m[1], m[2], m[3], m[4], m[5]);
  else
    fprintf(stderr, "[kh_lattice profile] ForwardBackwardDFKernel, %d lattices, shader cycles of wave 0 (mean): kernel %.0f (forward "
            "%.0f); staging %.0f, %.0f passes with no operand ready %.0f (%.0f each), %.0f passes that folded an operand %.0f (%.0f each: "
            "LDS read + LogAdd [+ posterior] + publish), end of block (stores + wait) %.0f; %.0f blocks\n", n_lats, m[13], m[14], m[6],
            m[11], m[7], m[11] > 0 ? m[7] / m[11] : 0.0, m[10], m[8], m[10] > 0 ? m[8] / m[10] : 0.0, m[9], m[12]);
}

struct LatBatch {
  int n_lats = 0, total_states = 0;
  int64_t total_arcs = 0;
  std::vector<LatDesc> descs;      // host copy (max_time, n_levels)
  DevArr<LatDesc> d_descs;
  DevArr<int64_t> d_arc_off, d_in_off, d_in_arc;
  DevArr<int32_t> d_lat_off, d_next, d_ilabel, d_level_off, d_level_states, d_in_src, d_final_list, d_times, d_indeg, d_fill, d_err;
  DevArr<float> d_g, d_a, d_fin;

  bool has_levels = true;   // dependency levels built (the MPE / alpha-beta / discriminative kernels sweep by level)
  int Build(int n, const int32_t *lat_state_offsets, const int64_t *arc_offsets, const int32_t *arc_ilabel,
            const int32_t *arc_nextstate, const float *arc_graph, const float *arc_acoustic,
            const float *state_final, hipStream_t st, bool want_levels = true) {
    n_lats = n;
    has_levels = want_levels;
    for (int l = 0; l < n_lats; l++) KH_CHECK_ARG(lat_state_offsets[l + 1] - lat_state_offsets[l] > 0);
    total_states = lat_state_offsets[n_lats];
    total_arcs = arc_offsets[total_states];
    const size_t S = total_states, A = static_cast<size_t>(total_arcs);
    if (d_lat_off.Alloc(n + 1) || d_arc_off.Alloc(S + 1) || d_next.Alloc(A) || d_ilabel.Alloc(A) || d_g.Alloc(A) ||
        d_a.Alloc(A) || d_fin.Alloc(S) || d_descs.Alloc(n) || d_times.Alloc(S) || d_level_off.Alloc(S + n) ||
        d_level_states.Alloc(S) || d_in_off.Alloc(S + 1) || d_in_arc.Alloc(A) || d_in_src.Alloc(A) ||
        d_final_list.Alloc(S) || d_indeg.Alloc(S) || d_fill.Alloc(S) || d_err.Alloc(4 * static_cast<size_t>(n)))
      return KH_ENOMEM;
    // the caller's arrays go up as they are (pageable copies are staged by the runtime and
    // complete before the call returns: the synchronisation below)
#define UP(dev, host, count, type) KH_HIP(hipMemcpyAsync(dev.p, host, sizeof(type) * (count), hipMemcpyHostToDevice, st))
    g_lat_timer.Begin();
    g_lat_timer.Mark(0, st);
    UP(d_lat_off, lat_state_offsets, n + 1, int32_t);
    UP(d_arc_off, arc_offsets, S + 1, int64_t);
    UP(d_next, arc_nextstate, A, int32_t);
    UP(d_ilabel, arc_ilabel, A, int32_t);
    UP(d_fin, state_final, S, float);
    if (arc_graph) UP(d_g, arc_graph, A, float);
    if (arc_acoustic) UP(d_a, arc_acoustic, A, float);
#undef UP
    g_lat_timer.Mark(1, st);
    if (want_levels)
      hipLaunchKernelGGL(PrepKernel, dim3(n_lats), dim3(kThreads), 0, st, d_lat_off.p, d_arc_off.p, d_ilabel.p, d_next.p,
                         d_fin.p, d_descs.p, d_times.p, d_level_off.p, d_level_states.p, d_in_off.p, d_in_arc.p,
                         d_in_src.p, d_final_list.p, d_indeg.p, d_fill.p, d_err.p);
    else {
      // dynamic LDS: the state-time window + the per-state counters of the batch's largest lattice, as far as the workgroups
      // that share a CU leave room (larger lattices count in device memory)
      int max_ns = 1;
      for (int l = 0; l < n_lats; l++) max_ns = std::max(max_ns, lat_state_offsets[l + 1] - lat_state_offsets[l]);
      const int cus = std::max(1, NumCUs());
      const int per_cu = std::min(4, (n_lats + cus - 1) / cus);
      const int budget = (160 * 1024) / per_cu - 21 * 1024;   // (staging + the static words)
      const int win = per_cu <= 2 ? 4096 : 1024;
      const int lds_states = std::max(0, std::min(max_ns, (budget - win * 8) / 4));
      const size_t dyn = static_cast<size_t>(win) * 8 + static_cast<size_t>(lds_states) * 4;
      DevArr<long long> d_prof;
      const bool prof = LatProfile();
      if (prof) {
        if (d_prof.Alloc(static_cast<size_t>(n_lats) * kLatProf)) return KH_ENOMEM;
        KH_HIP(hipMemsetAsync(d_prof.p, 0, sizeof(long long) * n_lats * kLatProf, st));
      }
#define KH_PREP_LAUNCH(PROF)                                                                                                     \
      do {                                                                                                                       \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) {                                                                                                         \
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&PrepKernelDF<PROF>), hipFuncAttributeMaxDynamicSharedMemorySize, 151 * 1024); \
          (void)hipGetLastError();                                                                                               \
          attr_set = true;                                                                                                       \
        }                                                                                                                        \
        hipLaunchKernelGGL(PrepKernelDF<PROF>, dim3(n_lats), dim3(kThreads), dyn, st, d_lat_off.p, d_arc_off.p, d_ilabel.p,      \
                           d_next.p, d_fin.p, d_descs.p, d_times.p, d_level_off.p, d_level_states.p, d_in_off.p, d_in_arc.p,      \
                           d_in_src.p, d_final_list.p, d_indeg.p, d_fill.p, d_err.p, d_prof.p, lds_states, win);                  \
      } while (0)
      if (prof) KH_PREP_LAUNCH(true); else KH_PREP_LAUNCH(false);
#undef KH_PREP_LAUNCH
      if (prof) {
        KH_LAUNCH_CHECK();
        fprintf(stderr, "[kh_lattice profile] prep: counters of up to %d states in LDS (largest lattice %d), window %d, %d workgroups per CU planned\n",
                lds_states, max_ns, win, per_cu);
        LatProfilePrint("prep", d_prof, n_lats, st);
      }
    }
    KH_LAUNCH_CHECK();
    descs.resize(n_lats);
    std::vector<int32_t> h_err(4 * static_cast<size_t>(n_lats));
    KH_HIP(hipMemcpyAsync(descs.data(), d_descs.p, sizeof(LatDesc) * n_lats, hipMemcpyDeviceToHost, st));
    KH_HIP(hipMemcpyAsync(h_err.data(), d_err.p, sizeof(int32_t) * h_err.size(), hipMemcpyDeviceToHost, st));
    g_lat_timer.Mark(2, st);
    KH_HIP(hipStreamSynchronize(st));
    for (int l = 0; l < n_lats; l++) {
      const int32_t *e = &h_err[4 * static_cast<size_t>(l)];
      if (e[0] == 1) {
        SetError("lattice %d: arc %d (state %d -> %d): input lattice must be topologically sorted", l, e[3], e[1], e[2]);
        return KH_EINVAL;
      }
      if (e[0] == 2) {
        SetError("lattice %d: inconsistent state times at state %d (%d vs %d; KALDI_ASSERT lattice-functions.cc:55,61)", l, e[1], e[2], e[3]);
        return KH_EINVAL;
      }
    }
    return KH_OK;
  }
  // LatticeStateTimes of every state (device -> caller)
  int FetchTimes(int32_t *state_times, hipStream_t st) const {
    KH_HIP(hipMemcpyAsync(state_times, d_times.p, sizeof(int32_t) * total_states, hipMemcpyDeviceToHost, st));
    return KH_OK;
  }
};

// The dataflow sweeps on a batch prepared without levels (PrepKernelDF): window sizing + launch.
static int LaunchForwardBackwardDF(const LatBatch &B, double *alpha, double *beta, float *post, double *tot, double *ac,
                                   double min_log_diff, hipStream_t st) {
  const int n_lats = B.n_lats;
  struct { double *p; } d_alpha{alpha}, d_beta{beta}, d_tot{tot}, d_ac{ac};
  struct { float *p; } d_post{post};
  // the window: as many states as the workgroups that share a CU leave room for (a lattice that fits it whole takes one LDS
  // read per operand), 512 ... 16384 slots and no more than the batch's largest lattice needs; KH_LATTICE_WIN overrides
  int max_ns = 1;
  for (const LatDesc &d : B.descs) max_ns = std::max(max_ns, d.n_states);
  const int cus = std::max(1, NumCUs());
  const int per_cu = std::min(4, (n_lats + cus - 1) / cus);   // (more than four resident workgroups per CU: the window gets too small)
  const bool small_stage = per_cu > 2;
  const int stage_bytes = (small_stage ? 192 : 384) * 16 * (kThreads / 64) + 256;
  int win = 512;
  while (win < 16384 && win < max_ns && (2 * win) * 8 + stage_bytes <= (160 * 1024) / per_cu) win *= 2;
  if (const char *e = getenv("KH_LATTICE_WIN")) { const int v = atoi(e); if (v >= 512 && v <= 16384 && (v & (v - 1)) == 0) win = v; }
  const size_t dyn = static_cast<size_t>(win) * sizeof(double);
  DevArr<long long> d_prof;
  const bool prof = LatProfile();
  if (prof) {
    if (d_prof.Alloc(static_cast<size_t>(n_lats) * kLatProf)) return KH_ENOMEM;
    KH_HIP(hipMemsetAsync(d_prof.p, 0, sizeof(long long) * n_lats * kLatProf, st));
  }
#define KH_FB_LAUNCH(PROF, STAGE)                                                                                              \
  do {                                                                                                                       \
    static bool attr_set = false;                                                                                            \
    if (!attr_set) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ForwardBackwardDFKernel<PROF, STAGE>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);                                      \
      (void)hipGetLastError();                                                                                               \
      attr_set = true;                                                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL((ForwardBackwardDFKernel<PROF, STAGE>), dim3(n_lats), dim3(kThreads), dyn, st, B.d_descs.p,           \
                       B.d_arc_off.p, B.d_next.p, B.d_g.p, B.d_a.p, B.d_fin.p, B.d_in_off.p, B.d_in_arc.p, B.d_in_src.p,     \
                       B.d_final_list.p, d_alpha.p, d_beta.p, d_post.p, d_tot.p, d_ac.p, min_log_diff, d_prof.p, win);       \
  } while (0)
  if (prof) { if (small_stage) KH_FB_LAUNCH(true, 192); else KH_FB_LAUNCH(true, 384); }
  else { if (small_stage) KH_FB_LAUNCH(false, 192); else KH_FB_LAUNCH(false, 384); }
#undef KH_FB_LAUNCH
  if (prof) {
    KH_LAUNCH_CHECK();
    fprintf(stderr, "[kh_lattice profile] window %d slots (largest lattice %d states), %d workgroups per CU planned\n", win, max_ns, per_cu);
    LatProfilePrint("sweeps", d_prof, n_lats, st);
  }
  KH_LAUNCH_CHECK();
  return KH_OK;
}

// LatticeForwardBackward :272-354 on a batch that is resident on the device: sweeps + download.
// post_dev != NULL: the posteriors stay on the device, in the caller's buffer (arc_post is then ignored)
static int RunForwardBackward(const LatBatch &B, float *arc_post, double *tot_like, double *acoustic_like_sum, hipStream_t st,
                              float *post_dev = nullptr) {
  const int n_lats = B.n_lats, total_states = B.total_states;
  const int64_t total_arcs = B.total_arcs;
  DevArr<float> d_post_own;
  DevArr<double> d_alpha, d_beta, d_tot, d_ac;
  if ((post_dev == nullptr && d_post_own.Alloc(total_arcs)) || d_alpha.Alloc(total_states) || d_beta.Alloc(total_states) ||
      d_tot.Alloc(n_lats) || d_ac.Alloc(n_lats))
    return KH_ENOMEM;
  struct { float *p; } d_post{post_dev ? post_dev : d_post_own.p};
  const double min_log_diff = log(DBL_EPSILON);  // kMinLogDiffDouble kaldi-math.h:120
  g_lat_timer.Mark(2, st);
  if (B.has_levels && !getenv("KH_LATTICE_DATAFLOW"))
    hipLaunchKernelGGL(ForwardBackwardKernel, dim3(n_lats), dim3(kThreads), 0, st, B.d_descs.p,
                       B.d_arc_off.p, B.d_next.p, B.d_g.p, B.d_a.p, B.d_fin.p, B.d_level_off.p, B.d_level_states.p,
                       B.d_in_off.p, B.d_in_arc.p, B.d_in_src.p, B.d_final_list.p, d_alpha.p, d_beta.p,
                       d_post.p, d_tot.p, d_ac.p, min_log_diff);
  else if (int rc_df = LaunchForwardBackwardDF(B, d_alpha.p, d_beta.p, d_post.p, d_tot.p, d_ac.p, min_log_diff, st))
    return rc_df;
  KH_LAUNCH_CHECK();
  g_lat_timer.Mark(3, st);
  if (arc_post && !post_dev)
    KH_HIP(hipMemcpyAsync(arc_post, d_post.p, sizeof(float) * total_arcs, hipMemcpyDeviceToHost, st));
  if (tot_like)
    KH_HIP(hipMemcpyAsync(tot_like, d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  if (acoustic_like_sum)
    KH_HIP(hipMemcpyAsync(acoustic_like_sum, d_ac.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  g_lat_timer.Mark(4, st);
  KH_HIP(hipStreamSynchronize(st));
  g_lat_timer.Finish();
  return KH_OK;
}

extern "C" int kh_lattice_last_timings(float *ms) {
  KH_CHECK_ARG(ms);
  for (int i = 0; i < 4; i++) ms[i] = g_lat_timer.ms[i];
  return KH_OK;
}

// ---- a batch of lattices kept on the device (one upload + one preparation for every computation on them)
struct KhLatticeBatch {
  LatBatch B;
  std::vector<int32_t> lat_off;   // host copy of the state offsets
};

extern "C" KhLatticeBatch *kh_lattice_batch_create(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                                                   const int32_t *arc_ilabel, const int32_t *arc_nextstate, const float *arc_graph,
                                                   const float *arc_acoustic, const float *state_final) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (!(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && arc_graph && arc_acoustic && state_final)) {
    SetError("kh_lattice_batch_create: bad arguments");
    return nullptr;
  }
  KhLatticeBatch *h = new KhLatticeBatch();
  if (h->B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, Stream(),
                 getenv("KH_LATTICE_LEVELS") != nullptr) != KH_OK) {
    delete h;
    return nullptr;
  }
  h->lat_off.assign(lat_state_offsets, lat_state_offsets + n_lats + 1);
  g_lat_timer.Finish();
  return h;
}

extern "C" void kh_lattice_batch_destroy(KhLatticeBatch *h) { delete h; }

extern "C" int kh_lattice_batch_sizes(const KhLatticeBatch *h, int32_t *n_lats, int32_t *total_states, int64_t *total_arcs) {
  KH_CHECK_ARG(h);
  if (n_lats) *n_lats = h->B.n_lats;
  if (total_states) *total_states = h->B.total_states;
  if (total_arcs) *total_arcs = h->B.total_arcs;
  return KH_OK;
}

extern "C" int kh_lattice_batch_forward_backward(KhLatticeBatch *h, float *arc_post, double *tot_like, double *acoustic_like_sum,
                                                 int32_t *state_times) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h);
  hipStream_t st = Stream();
  g_lat_timer.Begin();
  if (state_times && (rc = h->B.FetchTimes(state_times, st))) return rc;
  return RunForwardBackward(h->B, arc_post, tot_like, acoustic_like_sum, st);
}

// The same with the arc posteriors left on the device (arc_post_dev: DEVICE, total_arcs floats, in the batch's arc order):
// what a training loop does with them next (the posterior algebra, the derivative) runs there too.
extern "C" int kh_lattice_batch_forward_backward_dev(KhLatticeBatch *h, float *arc_post_dev, double *tot_like, double *acoustic_like_sum) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h && arc_post_dev);
  g_lat_timer.Begin();
  return RunForwardBackward(h->B, nullptr, tot_like, acoustic_like_sum, Stream(), arc_post_dev);
}

// RescoreLattice :1307-1358 on the resident batch: one workgroup per lattice, the acoustic costs are updated on the device
// (states at time utt_len and beyond have no transition-id arcs to rescore: :1343-1345 skips them)
__global__ void __launch_bounds__(kThreads)
RescoreBatchKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off, const int32_t *__restrict__ ilabel,
                   const int32_t *__restrict__ times, const int32_t *__restrict__ ll_row_off, float *__restrict__ arc_a,
                   const float *__restrict__ loglikes, int ll_stride, const int32_t *__restrict__ tid2pdf) {
  const LatDesc L = lats[blockIdx.x];
  for (int s = threadIdx.x; s < L.n_states; s += kThreads) {
    const int t = times[L.state_b + s];
    if (t < 0 || t >= L.max_time) continue;
    const float *row = loglikes + static_cast<size_t>(ll_row_off[blockIdx.x] + t) * ll_stride;
    for (int64_t a = arc_off[L.state_b + s]; a < arc_off[L.state_b + s + 1]; a++) {
      const int il = ilabel[a];
      if (il != 0) arc_a[a] = -row[tid2pdf ? tid2pdf[il] : il - 1] + arc_a[a];  // :1349-1350
    }
  }
}

extern "C" int kh_lattice_batch_rescore(KhLatticeBatch *h, const float *loglikes, int ll_stride, const int32_t *ll_row_offsets,
                                        const int32_t *tid2pdf, float *arc_acoustic_out) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h && loglikes && ll_stride > 0 && ll_row_offsets);
  const LatBatch &B = h->B;
  for (int l = 0; l < B.n_lats; l++)
    if (B.descs[l].max_time > ll_row_offsets[l + 1] - ll_row_offsets[l]) {
      SetError("lattice %d: Features are too short for lattice: utt-len is %d (lattice-functions.cc:1337-1341)", l, B.descs[l].max_time);
      return KH_EINVAL;
    }
  hipStream_t st = Stream();
  DevArr<int32_t> d_rows;
  std::vector<int32_t> h_rows(ll_row_offsets, ll_row_offsets + B.n_lats + 1);
  if ((rc = d_rows.Upload(h_rows, st))) return rc;
  hipLaunchKernelGGL(RescoreBatchKernel, dim3(B.n_lats), dim3(kThreads), 0, st, B.d_descs.p, B.d_arc_off.p, B.d_ilabel.p, B.d_times.p,
                     d_rows.p, B.d_a.p, loglikes, ll_stride, tid2pdf);
  KH_LAUNCH_CHECK();
  if (arc_acoustic_out)
    KH_HIP(hipMemcpyAsync(arc_acoustic_out, B.d_a.p, sizeof(float) * B.total_arcs, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  return KH_OK;
}

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
  hipStream_t st = Stream();
  LatBatch B;
  rc = B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, st,
               getenv("KH_LATTICE_LEVELS") != nullptr);
  if (rc) return rc;
  if (state_times && (rc = B.FetchTimes(state_times, st))) return rc;
  return RunForwardBackward(B, arc_post, tot_like, acoustic_like_sum, st);
}

// LatticeStateTimes (lat/lattice-functions.cc:36-67) of a batch of top-sorted lattices: the
// time of every state (-1 for a state no path from the start reaches) and, per lattice, the
// number of frames.  Device preparation only (the caller of
// NnetDiscriminativeUpdater::LatticeComputations needs the times before any sweep,
// nnet-compute-discriminative.cc:218-220).
extern "C" int kh_lattice_state_times(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                                      const int32_t *arc_ilabel, const int32_t *arc_nextstate, const float *state_final,
                                      int32_t *state_times, int32_t *max_times) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && state_final && state_times);
  hipStream_t st = Stream();
  LatBatch B;
  rc = B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, nullptr, nullptr, state_final, st,
               getenv("KH_LATTICE_LEVELS") != nullptr);
  if (rc) return rc;
  if ((rc = B.FetchTimes(state_times, st))) return rc;
  KH_HIP(hipStreamSynchronize(st));
  if (max_times)
    for (int l = 0; l < n_lats; l++) max_times[l] = B.descs[l].max_time;
  return KH_OK;
}

// base/kaldi-math.h ApproxEqual
static bool ApproxEqualD(double a, double b, double tol) {
  if (a == b) return true;
  const double diff = std::fabs(a - b);
  if (diff == std::numeric_limits<double>::infinity() || diff != diff) return false;
  return diff <= tol * (std::fabs(a) + std::fabs(b));
}

static int RunAlphaBeta(const LatBatch &B, DevArr<double> &d_alpha, DevArr<double> &d_beta, DevArr<double> &d_tot,
                        int viterbi, hipStream_t st) {
  if (d_alpha.Alloc(B.total_states) || d_beta.Alloc(B.total_states) || d_tot.Alloc(B.n_lats)) return KH_ENOMEM;
  hipLaunchKernelGGL(AlphaBetaKernel, dim3(B.n_lats), dim3(kThreads), 0, st, B.d_descs.p, B.d_arc_off.p, B.d_next.p,
                     B.d_g.p, B.d_a.p, B.d_fin.p, B.d_level_off.p, B.d_level_states.p, B.d_in_off.p, B.d_in_arc.p,
                     B.d_in_src.p, B.d_final_list.p, d_alpha.p, d_beta.p, d_tot.p, viterbi, log(DBL_EPSILON));
  KH_LAUNCH_CHECK();
  return KH_OK;
}

extern "C" int kh_lattice_alphas_betas(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                                       const int32_t *arc_ilabel, const int32_t *arc_nextstate,
                                       const float *arc_graph, const float *arc_acoustic, const float *state_final,
                                       int viterbi, double *alpha, double *beta, double *tot) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && arc_graph &&
               arc_acoustic && state_final);
  hipStream_t st = Stream();
  LatBatch B;
  rc = B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, st);
  if (rc) return rc;
  DevArr<double> d_alpha, d_beta, d_tot;
  rc = RunAlphaBeta(B, d_alpha, d_beta, d_tot, viterbi, st);
  if (rc) return rc;
  std::vector<double> h_beta(B.total_states), h_tot(n_lats);
  if (alpha) KH_HIP(hipMemcpyAsync(alpha, d_alpha.p, sizeof(double) * B.total_states, hipMemcpyDeviceToHost, st));
  KH_HIP(hipMemcpyAsync(h_beta.data(), d_beta.p, sizeof(double) * B.total_states, hipMemcpyDeviceToHost, st));
  KH_HIP(hipMemcpyAsync(h_tot.data(), d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  if (beta) memcpy(beta, h_beta.data(), sizeof(double) * B.total_states);
  if (tot)
    for (int l = 0; l < n_lats; l++) tot[l] = 0.5 * (h_beta[lat_state_offsets[l]] + h_tot[l]);  // :462
  return KH_OK;
}

extern "C" int kh_lattice_forward_backward_mpe(int n_lats, const int32_t *lat_state_offsets,
                                               const int64_t *arc_offsets, const int32_t *arc_ilabel,
                                               const int32_t *arc_nextstate, const float *arc_graph,
                                               const float *arc_acoustic, const float *state_final,
                                               const int32_t *tid2phone, const int32_t *tid2pdf, int num_tids,
                                               const int32_t *silence_phones, int n_sil, const int32_t *num_ali,
                                               const int32_t *num_ali_offsets, int is_mpfe, int one_silence_class,
                                               float *arc_post, double *tot_forward_score, int32_t *state_times) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && arc_graph &&
               arc_acoustic && state_final && tid2phone && tid2pdf && num_tids > 0 && n_sil >= 0 &&
               (n_sil == 0 || silence_phones) && num_ali && num_ali_offsets);
  hipStream_t st = Stream();
  LatBatch B;
  rc = B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, st);
  if (rc) return rc;
  if (state_times && (rc = B.FetchTimes(state_times, st))) return rc;
  for (int64_t a = 0; a < B.total_arcs; a++) KH_CHECK_ARG(arc_ilabel[a] >= 0 && arc_ilabel[a] <= num_tids);
  for (int l = 0; l < n_lats; l++) {  // max_time == num_ali.size() :764
    const int max_time = B.descs[l].max_time;
    if (max_time != num_ali_offsets[l + 1] - num_ali_offsets[l]) {
      SetError("lattice %d: max_time %d != num_ali.size() %d (KALDI_ASSERT lattice-functions.cc:764)", l, max_time,
               num_ali_offsets[l + 1] - num_ali_offsets[l]);
      return KH_EINVAL;
    }
  }
  const int n_ali = num_ali_offsets[n_lats];
  for (int i = 0; i < n_ali; i++) KH_CHECK_ARG(num_ali[i] > 0 && num_ali[i] <= num_tids);
  DevArr<double> d_alpha, d_beta, d_tot, d_as, d_bs, d_score, d_bscore;
  DevArr<float> d_post;
  DevArr<int32_t> d_t2ph, d_t2pdf, d_sil, d_ali, d_ali_off;
  rc = RunAlphaBeta(B, d_alpha, d_beta, d_tot, 0, st);
  if (rc) return rc;
  std::vector<int32_t> h_t2ph(tid2phone, tid2phone + num_tids + 1), h_t2pdf(tid2pdf, tid2pdf + num_tids + 1),
      h_sil(silence_phones, silence_phones + n_sil), h_ali(num_ali, num_ali + n_ali),
      h_ali_off(num_ali_offsets, num_ali_offsets + n_lats + 1);
  if ((rc = d_t2ph.Upload(h_t2ph, st)) || (rc = d_t2pdf.Upload(h_t2pdf, st)) || (rc = d_sil.Upload(h_sil, st)) ||
      (rc = d_ali.Upload(h_ali, st)) || (rc = d_ali_off.Upload(h_ali_off, st)))
    return rc;
  if (d_as.Alloc(B.total_states) || d_bs.Alloc(B.total_states) || d_score.Alloc(n_lats) || d_bscore.Alloc(n_lats) ||
      d_post.Alloc(B.total_arcs))
    return KH_ENOMEM;
  MpeArgs m;
  m.tid2phone = d_t2ph.p; m.tid2pdf = d_t2pdf.p; m.sil = d_sil.p; m.num_ali = d_ali.p; m.ali_off = d_ali_off.p;
  m.times = B.d_times.p; m.ilabel = B.d_ilabel.p; m.n_sil = n_sil; m.is_mpfe = is_mpfe; m.one_silence_class = one_silence_class;
  hipLaunchKernelGGL(MpeKernel, dim3(n_lats), dim3(kThreads), 0, st, B.d_descs.p, B.d_arc_off.p, B.d_next.p, B.d_g.p,
                     B.d_a.p, B.d_fin.p, B.d_level_off.p, B.d_level_states.p, B.d_in_off.p, B.d_in_arc.p, B.d_in_src.p,
                     B.d_final_list.p, d_alpha.p, d_beta.p, d_tot.p, d_as.p, d_bs.p, m, d_post.p, d_score.p, d_bscore.p);
  KH_LAUNCH_CHECK();
  std::vector<double> h_tot(n_lats), h_beta0(n_lats), h_score(n_lats), h_bscore(n_lats);
  KH_HIP(hipMemcpyAsync(h_tot.data(), d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  KH_HIP(hipMemcpyAsync(h_score.data(), d_score.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  KH_HIP(hipMemcpyAsync(h_bscore.data(), d_bscore.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, st));
  for (int l = 0; l < n_lats; l++)
    KH_HIP(hipMemcpyAsync(&h_beta0[l], d_beta.p + lat_state_offsets[l], sizeof(double), hipMemcpyDeviceToHost, st));
  if (arc_post) KH_HIP(hipMemcpyAsync(arc_post, d_post.p, sizeof(float) * B.total_arcs, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  for (int l = 0; l < n_lats; l++) {
    if (!ApproxEqualD(h_tot[l], h_beta0[l], 1e-6)) {  // :808-811
      SetError("lattice %d: Total forward probability over lattice = %g, while total backward probability = %g", l,
               h_tot[l], h_beta0[l]);
      return KH_ESTATE;
    }
    if (!ApproxEqualD(h_score[l], h_bscore[l], 1e-4)) {  // :909-912
      SetError("lattice %d: Total forward score over lattice = %g, while total backward score = %g", l, h_score[l],
               h_bscore[l]);
      return KH_ESTATE;
    }
    if (tot_forward_score) tot_forward_score[l] = h_score[l];
  }
  return KH_OK;
}

extern "C" int kh_rescore_lattice(int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets,
                                  const int32_t *arc_ilabel, const int32_t *arc_nextstate, float *arc_acoustic,
                                  const float *loglikes, int ll_stride, const int32_t *ll_row_offsets,
                                  const int32_t *tid2pdf) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && arc_acoustic &&
               loglikes && ll_stride > 0 && ll_row_offsets);
  const int total_states = lat_state_offsets[n_lats];
  const int64_t total_arcs = arc_offsets[total_states];
  // LatticeStateTimes; states with t == utt_len have no transition-id arcs to rescore
  std::vector<int32_t> times(total_states, -1), state_lat(total_states);
  for (int l = 0; l < n_lats; l++) {
    const int sb = lat_state_offsets[l], ns = lat_state_offsets[l + 1] - sb;
    int utt_len = 0;
    times[sb] = 0;
    for (int s = 0; s < ns; s++) {
      state_lat[sb + s] = l;
      for (int64_t a = arc_offsets[sb + s]; a < arc_offsets[sb + s + 1]; a++) {
        const int nxt = arc_nextstate[a];
        KH_CHECK_ARG(nxt > s && nxt < ns);  // top-sorted (the reference sorts first, :1313-1318)
        if (times[sb + s] >= 0) {
          const int want = times[sb + s] + (arc_ilabel[a] != 0 ? 1 : 0);
          if (times[sb + nxt] == -1) times[sb + nxt] = want;
          utt_len = std::max(utt_len, want);
        }
      }
    }
    if (utt_len > ll_row_offsets[l + 1] - ll_row_offsets[l]) {
      SetError("lattice %d: Features are too short for lattice: utt-len is %d (lattice-functions.cc:1337-1341)", l, utt_len);
      return KH_EINVAL;
    }
    for (int s = 0; s < ns; s++)
      if (times[sb + s] >= utt_len) times[sb + s] = -1;
  }
  hipStream_t st = Stream();
  DevArr<int64_t> d_off;
  DevArr<int32_t> d_il, d_times, d_lat, d_rows;
  DevArr<float> d_a;
  std::vector<int64_t> h_off(arc_offsets, arc_offsets + total_states + 1);
  std::vector<int32_t> h_il(arc_ilabel, arc_ilabel + total_arcs), h_rows(ll_row_offsets, ll_row_offsets + n_lats + 1);
  std::vector<float> h_a(arc_acoustic, arc_acoustic + total_arcs);
  if ((rc = d_off.Upload(h_off, st)) || (rc = d_il.Upload(h_il, st)) || (rc = d_times.Upload(times, st)) ||
      (rc = d_lat.Upload(state_lat, st)) || (rc = d_rows.Upload(h_rows, st)) || (rc = d_a.Upload(h_a, st)))
    return rc;
  hipLaunchKernelGGL(RescoreKernel, dim3(std::min(1024, (total_states + 255) / 256)), dim3(256), 0, st, total_states,
                     d_off.p, d_il.p, d_times.p, d_lat.p, d_rows.p, d_a.p, loglikes, ll_stride, tid2pdf);
  KH_LAUNCH_CHECK();
  KH_HIP(hipMemcpyAsync(arc_acoustic, d_a.p, sizeof(float) * total_arcs, hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  return KH_OK;
}

extern "C" int kh_comp_objf_and_deriv(int n, const int32_t *rows, const int32_t *cols, const float *weights,
                                      const float *output, KhMatrixDim d_output, float *deriv, KhMatrixDim d_deriv,
                                      float *tot_objf, float *tot_weight) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n >= 0 && output && deriv && tot_objf && tot_weight && d_output.rows == d_deriv.rows &&
               d_output.cols == d_deriv.cols && (n == 0 || (rows && cols && weights)));
  *tot_objf = 0.0f;
  *tot_weight = 0.0f;
  if (n == 0) return KH_OK;  // "Empty supervision labels" :1214-1218
  for (int i = 0; i < n; i++)  // :1202-1207
    KH_CHECK_ARG(rows[i] >= 0 && rows[i] < d_deriv.rows && cols[i] >= 0 && cols[i] < d_deriv.cols);
  hipStream_t st = Stream();
  DevArr<int32_t> d_r, d_c;
  DevArr<float> d_w;
  DevArr<double> d_sums;
  std::vector<int32_t> h_r(rows, rows + n), h_c(cols, cols + n);
  std::vector<float> h_w(weights, weights + n);
  if ((rc = d_r.Upload(h_r, st)) || (rc = d_c.Upload(h_c, st)) || (rc = d_w.Upload(h_w, st))) return rc;
  if (d_sums.Alloc(2)) return KH_ENOMEM;
  KH_HIP(hipMemsetAsync(d_sums.p, 0, sizeof(double) * 2, st));
  hipLaunchKernelGGL(CompObjfKernel, dim3(std::min(256, (n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, n,
                     d_r.p, d_c.p, d_w.p, output, d_output.stride, deriv, d_deriv.stride, d_sums.p);
  KH_LAUNCH_CHECK();
  double h[2];
  KH_HIP(hipMemcpyAsync(h, d_sums.p, sizeof(h), hipMemcpyDeviceToHost, st));
  KH_HIP(hipStreamSynchronize(st));
  *tot_objf = static_cast<float>(h[0]);
  *tot_weight = static_cast<float>(h[1]);
  return KH_OK;
}

// MergePairVectorSumming (util/stl-utils.h:303-322) for the posteriors of all frames at once
// (the Posterior algebra of hmm/posterior.cc runs on the host in the reference too): the entries
// (row = frame, key = transition-id or pdf-id, weight) are sorted by (row, key) — counting sort by
// row, insertion sort inside a row (a frame holds a handful of entries) — equal keys summed in that
// order, exact zeros dropped.
extern "C" int kh_merge_pair_vector_summing(int64_t n, const int32_t *rows, const int32_t *keys, const float *weights,
                                            int32_t n_rows, int32_t *out_rows, int32_t *out_keys, float *out_weights,
                                            int64_t *n_out) {
  KH_CHECK_ARG(n >= 0 && n_rows >= 0 && n_out && (n == 0 || (rows && keys && weights && out_rows && out_keys && out_weights)));
  std::vector<int64_t> start(static_cast<size_t>(n_rows) + 1, 0);
  for (int64_t i = 0; i < n; i++) {
    KH_CHECK_ARG(rows[i] >= 0 && rows[i] < n_rows);
    start[rows[i] + 1]++;
  }
  for (int32_t r = 0; r < n_rows; r++) start[r + 1] += start[r];
  std::vector<int64_t> fill(start.begin(), start.end() - 1);
  std::vector<int32_t> k(n);
  std::vector<float> w(n);
  for (int64_t i = 0; i < n; i++) {   // stable
    const int64_t d = fill[rows[i]]++;
    k[d] = keys[i];
    w[d] = weights[i];
  }
  int64_t o = 0;
  for (int32_t r = 0; r < n_rows; r++) {
    const int64_t b = start[r], e = start[r + 1];
    for (int64_t i = b + 1; i < e; i++) {   // stable insertion sort by key
      const int32_t ki = k[i];
      const float wi = w[i];
      int64_t j = i;
      for (; j > b && k[j - 1] > ki; j--) { k[j] = k[j - 1]; w[j] = w[j - 1]; }
      k[j] = ki;
      w[j] = wi;
    }
    for (int64_t i = b; i < e;) {
      const int32_t key = k[i];
      float sum = w[i];
      for (i++; i < e && k[i] == key; i++) sum += w[i];
      if (sum != 0.0f) { out_rows[o] = r; out_keys[o] = key; out_weights[o] = sum; o++; }
    }
  }
  *n_out = o;
  return KH_OK;
}

// ===================================================================================================
// NnetDiscriminativeUpdater::LatticeComputations (nnet2/nnet-compute-discriminative.cc:178-321) for a
// batch of examples with everything between the network output and the derivative ON THE DEVICE:
//   Lookup of the posteriors the arcs / the numerator alignment need (:196-226) and the pseudo
//   log-likelihoods log(max(post, 1e-20) / prior) x acoustic_scale (:231-247), written into the
//   lattice (:259-277)                                              -> PseudoLikeKernel, NumLikeKernel
//   GetDiscriminativePosteriors (:324-343): LatticeForwardBackwardMmi / ...MpeVariants
//                                                                    -> the sweeps above
//   the Posterior algebra (hmm/posterior.cc): MergePairVectorSumming per (frame, transition-id)
//   (lattice-functions.cc:351-352), ScalePosterior(-1), ConvertPosteriorToPdfs, AlignmentToPosterior,
//   MergePosteriors with cancel + drop_frames (:244-274), ScalePosterior(eg.weight) (:283)
//                                    -> one stable radix sort of (frame row, pdf) keys + SegmentKernel
//   CompObjfAndDeriv (:301-316)                                      -> EmitKernel
// The sums run in the reference's order: within a (row, pdf) the transition-ids ascend (the order
// MergePairVectorSumming leaves and ConvertPosteriorToPdfs then accumulates in), within a transition-id
// the arcs ascend (the order LatticeForwardBackward pushes them); the sort is stable on the arc index so
// a segment holds its arcs in that order.  Round 2 ran this algebra in numpy between the device steps:
// 117 ms per 256-lattice batch for 22 ms of device work.
namespace {

// one workgroup per lattice: key (row << 32 | pdf) and pseudo log-likelihood of every arc with a transition-id
__global__ void __launch_bounds__(kThreads)
PseudoLikeKernel(const LatDesc *__restrict__ lats, const int64_t *__restrict__ arc_off, const int32_t *__restrict__ ilabel,
                 const int32_t *__restrict__ times, const int32_t *__restrict__ row_off, const int32_t *__restrict__ tid2pdf,
                 const float *__restrict__ post, int post_stride, const float *__restrict__ priors, float acoustic_scale,
                 int total_rows, float *__restrict__ arc_a, unsigned long long *__restrict__ keys, int32_t *__restrict__ vals,
                 int what /* 1: the keys (they depend on the lattice alone); 2: the pseudo log-likelihoods; 3: both */) {
  const LatDesc L = lats[blockIdx.x];
  const int row0 = row_off[blockIdx.x];
  const unsigned long long none = static_cast<unsigned long long>(total_rows) << 32;   // sorts behind every row
  for (int s = threadIdx.x; s < L.n_states; s += kThreads) {
    const int t = times[L.state_b + s];
    for (int64_t a = arc_off[L.state_b + s]; a < arc_off[L.state_b + s + 1]; a++) {
      const int il = ilabel[a];
      if (what & 1) vals[a] = static_cast<int32_t>(a);
      if (il == 0 || t < 0) { if (what & 1) keys[a] = none; continue; }
      const int pdf = tid2pdf[il], row = row0 + t;
      if (what & 1) keys[a] = (static_cast<unsigned long long>(row) << 32) | static_cast<unsigned>(pdf);
      if (!(what & 2)) continue;
      float p = post[static_cast<size_t>(row) * post_stride + pdf];
      if (p < 1.0e-20f) p = 1.0e-20f;                                   // ApplyFloor(1e-20) :233
      const float pseudo = logf(p / priors[pdf]) * acoustic_scale;       // :241-247
      arc_a[a] = -pseudo;                                                // SetValue2(-log_like) :270
    }
  }
}

// MMI numerator: sum over the alignment of its pseudo log-likelihoods (:249-257), one workgroup per example
__global__ void __launch_bounds__(kThreads)
NumLikeKernel(const int32_t *__restrict__ row_off, const int32_t *__restrict__ ali, const int32_t *__restrict__ tid2pdf,
              const float *__restrict__ post, int post_stride, const float *__restrict__ priors, float acoustic_scale,
              double *__restrict__ num_like) {
  __shared__ double s_part[kThreads];
  double acc = 0.0;
  for (int r = row_off[blockIdx.x] + threadIdx.x; r < row_off[blockIdx.x + 1]; r += kThreads) {
    const int pdf = tid2pdf[ali[r]];
    float p = post[static_cast<size_t>(r) * post_stride + pdf];
    if (p < 1.0e-20f) p = 1.0e-20f;
    acc += static_cast<double>(logf(p / priors[pdf]) * acoustic_scale);
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {       // fixed tree: the same sum on every run
    if (threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) num_like[blockIdx.x] = s_part[0];
}

// sorted entries -> per (row, pdf) segment head: the sum of its arcs' posteriors, transition-ids ascending, arcs
// ascending inside one; zero sums of a transition-id are dropped (MergePairVectorSumming), the sign is the MMI
// denominator's ScalePosterior(-1).  row_has_num[row] = the numerator's pdf occurs in the row's denominator.
__global__ void SegmentKernel(int64_t n, const unsigned long long *__restrict__ keys, const int32_t *__restrict__ vals,
                              const int32_t *__restrict__ ilabel, const float *__restrict__ arc_post, int total_rows, float sign,
                              const int32_t *__restrict__ ali_pdf, float *__restrict__ seg_sum, int32_t *__restrict__ row_has_num) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const unsigned long long key = keys[i];
    const int row = static_cast<int>(key >> 32);
    if (row >= total_rows) { seg_sum[i] = 0.0f; continue; }
    if (i > 0 && keys[i - 1] == key) { seg_sum[i] = 0.0f; continue; }   // not a head
    int64_t e = i + 1;
    while (e < n && keys[e] == key) e++;
    float sum = 0.0f;
    bool first = true;
    int last_tid = 0;
    for (;;) {
      int tid = 0x7fffffff;
      for (int64_t j = i; j < e; j++) {
        const int tj = ilabel[vals[j]];
        if (tj > last_tid && tj < tid) tid = tj;
      }
      if (tid == 0x7fffffff) break;
      float s = 0.0f;
      bool f2 = true;
      for (int64_t j = i; j < e; j++)
        if (ilabel[vals[j]] == tid) {
          const float v = arc_post[vals[j]];
          s = f2 ? v : s + v;
          f2 = false;
        }
      if (s != 0.0f) {
        const float w = sign * s;
        sum = first ? w : sum + w;
        first = false;
      }
      last_tid = tid;
    }
    seg_sum[i] = sum;
    if (ali_pdf != nullptr && sum != 0.0f && static_cast<int>(key & 0xffffffffu) == ali_pdf[row]) row_has_num[row] = 1;
  }
}

__device__ __forceinline__ int EgOfRow(const int32_t *__restrict__ row_off, int n_egs, int row) {
  int lo = 0, hi = n_egs;   // row_off[lo] <= row < row_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_off[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// the merged, weighted posterior of every (row, pdf) -> CompObjfAndDeriv.  Entries [0, n) are the segment heads,
// entries [n, n + total_rows) the numerator of rows whose pdf is not in the denominator (MMI only).
// partials[block] = {objf, weight, positive count}.
__global__ void __launch_bounds__(kThreads)
EmitKernel(int64_t n, const unsigned long long *__restrict__ keys, const float *__restrict__ seg_sum, int total_rows, int is_mmi,
           int drop_frames, const int32_t *__restrict__ ali_pdf, const int32_t *__restrict__ row_has_num,
           const int32_t *__restrict__ row_off, int n_egs, const float *__restrict__ eg_weights, const float *__restrict__ output,
           int out_stride, float *__restrict__ deriv, int deriv_stride, double *__restrict__ partials) {
  __shared__ double s_part[3][kThreads];
  double objf = 0.0, wsum = 0.0, pos = 0.0;
  const int64_t total = n + (is_mmi ? total_rows : 0);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(kThreads) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kThreads) {
    int row, pdf;
    float w;
    if (i < n) {
      w = seg_sum[i];
      if (w == 0.0f) continue;                       // not a head, or an entry MergePairVectorSumming dropped
      row = static_cast<int>(keys[i] >> 32);
      pdf = static_cast<int>(keys[i] & 0xffffffffu);
      if (is_mmi) {
        if (drop_frames && !row_has_num[row]) continue;          // MergePosteriors :266-270
        if (pdf == ali_pdf[row]) w = 1.0f + w;                   // numerator first, then the denominator's entry
        if (w == 0.0f) continue;                                 // cancelled exactly
      }
    } else {
      row = static_cast<int>(i - n);
      if (row_has_num[row] || drop_frames) continue;
      pdf = ali_pdf[row];
      w = 1.0f;
    }
    w *= eg_weights[EgOfRow(row_off, n_egs, row)];               // ScalePosterior(eg.weight) :283
    const float p = output[static_cast<size_t>(row) * out_stride + pdf];
    objf += static_cast<double>(w * logf(p));
    wsum += static_cast<double>(w);
    if (w > 0.0f) pos += static_cast<double>(w);
    deriv[static_cast<size_t>(row) * deriv_stride + pdf] += w / p;   // (row, pdf) is unique here
  }
  s_part[0][threadIdx.x] = objf; s_part[1][threadIdx.x] = wsum; s_part[2][threadIdx.x] = pos;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w)
      for (int k = 0; k < 3; k++) s_part[k][threadIdx.x] += s_part[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; k++) partials[3 * blockIdx.x + k] = s_part[k][0];
}

__global__ void AliPdfKernel(int n, const int32_t *__restrict__ ali, const int32_t *__restrict__ tid2pdf, int32_t *__restrict__ ali_pdf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) ali_pdf[i] = tid2pdf[ali[i]];
}

}  // namespace

namespace {
// What does not depend on the network output - the upload and preparation of the lattices, the (row, pdf) keys of their arcs
// and the sort of those keys - runs on a stream of its own, beside the forward pass that is still on the library's stream
// when the call arrives (the caller launches the forward pass and does not wait for it): of the call's device work only the
// pseudo log-likelihoods, the sweeps and the two kernels of the posterior algebra are left behind the forward pass.
hipStream_t SideStream(int k = 0) {   // two of them: calls in flight together (begin / end) alternate
  static hipStream_t side[2] = {nullptr, nullptr};
  static std::once_flag once;
  std::call_once(once, [] {
    for (int i = 0; i < 2; i++)
      if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side[i] = nullptr; }
  });
  return side[k & 1];
}
hipEvent_t SideEvent() {
  static hipEvent_t ev = [] {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    return e;
  }();
  return ev;
}
// Small pinned blocks for the results of the discriminative calls, kept for reuse: hipHostFree waits for the whole device -
// freed per call it made _end(i) wait for the forward pass of batch i + 1.
struct PinnedResults {
  std::mutex mu;
  struct Slab { double *p; size_t n; bool used; };
  std::vector<Slab> slabs;
  double *Take(size_t n) {
    std::lock_guard<std::mutex> l(mu);
    for (Slab &s : slabs)
      if (!s.used && s.n >= n) { s.used = true; return s.p; }
    double *p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&p), sizeof(double) * n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    slabs.push_back(Slab{p, n, true});
    return p;
  }
  void Give(double *p) {
    std::lock_guard<std::mutex> l(mu);
    for (Slab &s : slabs)
      if (s.p == p) s.used = false;
  }
};
PinnedResults g_results;
}  // namespace

// One call of the discriminative lattice computations: what it holds on the device and on the host until its results
// have been read.  (kh_discriminative_lattice_computations_begin / _end keep it between the two calls.)
struct KhDiscCall {
  LatBatch B;
  DevArr<int32_t> d_ali, d_row_off, d_t2pdf, d_t2ph, d_sil, d_ali_pdf, d_has_num, d_vals, d_vals2;
  DevArr<float> d_w, d_pri, d_post, d_seg;
  DevArr<double> d_alpha, d_beta, d_tot, d_ac, d_num, d_as, d_bs, d_score, d_bscore, d_part;
  DevArr<unsigned long long> d_keys, d_keys2;
  DevArr<uint32_t> d_sort;
  std::vector<float> w;              // the examples' weights (the statistics are weighted sums over the lattices)
  double *pinned = nullptr;          // results of the device: 3 x 256 partial sums + five values per lattice
  hipEvent_t ev_fwd = nullptr;       // (overlapped) everything queued on the library's stream when the call began
  hipStream_t side = nullptr, tail = nullptr;   // preparation; the steps behind the network's output
  int n_lats = 0, is_mmi = 0;
  bool timing = false;
  std::chrono::steady_clock::time_point t_in;
  double tm_build = 0, tm_side = 0, tm_launched = 0;
  // no device array of the call is freed while either stream may still use it
  ~KhDiscCall() {
    if (side) (void)hipStreamSynchronize(side);
    if (tail && tail != side) (void)hipStreamSynchronize(tail);
    if (ev_fwd) (void)hipEventDestroy(ev_fwd);
    if (pinned) g_results.Give(pinned);
    (void)hipGetLastError();
  }
};

namespace {
constexpr int kEmitBlocks = 256;

// Everything up to the copies of the results.  overlap: the steps behind the network's output run on the call's side stream
// too, behind an event recorded on the library's stream NOW - what the caller queues there afterwards (the next batch's
// forward pass) runs beside them.
int DiscBegin(KhDiscCall &C,
    int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets, const int32_t *arc_ilabel,
    const int32_t *arc_nextstate, const float *arc_graph, const float *arc_acoustic, const float *state_final,
    const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights, const int32_t *tid2pdf,
    const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion, float acoustic_scale,
    int drop_frames, int one_silence_class, const float *priors, const float *posteriors, KhMatrixDim d_posteriors,
    float *deriv, KhMatrixDim d_deriv, bool labels_checked, bool overlap) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n_lats > 0 && lat_state_offsets && arc_offsets && arc_ilabel && arc_nextstate && arc_graph && arc_acoustic &&
               state_final && num_ali && num_ali_offsets && eg_weights && tid2pdf && num_tids > 0 && priors && posteriors &&
               deriv && criterion >= 0 && criterion <= 2 && (criterion == 0 || tid2phone) && n_sil >= 0 &&
               (n_sil == 0 || silence_phones) && d_posteriors.rows == d_deriv.rows && d_posteriors.cols == d_deriv.cols);
  const int is_mmi = criterion == 0;
  const int total_rows = num_ali_offsets[n_lats], P = d_posteriors.cols;
  if (total_rows != d_posteriors.rows) {
    SetError("KALDI_ASSERT(posteriors.NumRows() == num_frames) nnet-compute-discriminative.cc:194: %d rows, %d frames",
             d_posteriors.rows, total_rows);
    return KH_EINVAL;
  }
  for (int t = 1; t <= num_tids; t++)
    if (tid2pdf[t] < 0 || tid2pdf[t] >= P) {
      SetError("transition-id %d maps to pdf %d, outside the %d columns of the network output", t, tid2pdf[t], P);
      return KH_EINVAL;
    }
  for (int i = 0; i < total_rows; i++) KH_CHECK_ARG(num_ali[i] > 0 && num_ali[i] <= num_tids);
  // KH_LATTICE_TIMING: the host's view of the call on stderr (ms since its entry)
  C.timing = getenv("KH_LATTICE_TIMING") != nullptr;
  C.t_in = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C.t_in).count(); };
  double &tm_build = C.tm_build, &tm_side = C.tm_side, &tm_launched = C.tm_launched;
  hipStream_t st = Stream();
  static std::atomic<int> n_calls{0};
  hipStream_t side = SideStream(overlap ? n_calls.fetch_add(1) : 0);
  hipEvent_t side_ev = SideEvent();
  if (side == nullptr || side_ev == nullptr || getenv("KH_LATTICE_ONE_STREAM") != nullptr) {
    if (overlap) { SetError("kh_discriminative_lattice_computations_begin: no second stream"); return KH_EDEVICE; }
    side = st;
  }
  hipStream_t ts = overlap ? side : st;   // where the steps behind the network's output run
  C.side = side; C.tail = ts; C.n_lats = n_lats; C.is_mmi = is_mmi;
  if (overlap) {
    KH_HIP(hipEventCreateWithFlags(&C.ev_fwd, hipEventDisableTiming));
    KH_HIP(hipEventRecord(C.ev_fwd, st));
  }
  C.pinned = g_results.Take(3 * kEmitBlocks + 5 * static_cast<size_t>(n_lats));
  if (C.pinned == nullptr) { SetError("kh_discriminative_lattice_computations: cannot allocate pinned host memory"); return KH_ENOMEM; }
  memset(C.pinned, 0, sizeof(double) * (3 * kEmitBlocks + 5 * static_cast<size_t>(n_lats)));
  C.w.assign(eg_weights, eg_weights + n_lats);
  LatBatch &B = C.B;
  // (MMI needs LatticeForwardBackward only: the dataflow preparation and sweeps; the MPE / sMBR kernels sweep by level)
  rc = B.Build(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, side,
               !is_mmi || getenv("KH_LATTICE_LEVELS") != nullptr);
  if (rc) return rc;
  tm_build = since();
  if (!labels_checked)
    for (int64_t a = 0; a < B.total_arcs; a++) KH_CHECK_ARG(arc_ilabel[a] >= 0 && arc_ilabel[a] <= num_tids);
  for (int l = 0; l < n_lats; l++)
    if (B.descs[l].max_time != num_ali_offsets[l + 1] - num_ali_offsets[l]) {
      SetError("example %d: lattice of %d frames, alignment of %d (KALDI_ASSERT(T == num_frames) nnet-compute-discriminative.cc:220)",
               l, B.descs[l].max_time, num_ali_offsets[l + 1] - num_ali_offsets[l]);
      return KH_EINVAL;
    }
  const int64_t A = B.total_arcs;
  if (A >= (1ll << 31)) { SetError("kh_discriminative_lattice_computations: %lld arcs in one batch (the sort takes < 2^31)", static_cast<long long>(A)); return KH_EINVAL; }
  auto &d_ali = C.d_ali; auto &d_row_off = C.d_row_off; auto &d_t2pdf = C.d_t2pdf; auto &d_t2ph = C.d_t2ph; auto &d_sil = C.d_sil;
  auto &d_ali_pdf = C.d_ali_pdf; auto &d_has_num = C.d_has_num; auto &d_vals = C.d_vals; auto &d_vals2 = C.d_vals2;
  auto &d_w = C.d_w; auto &d_pri = C.d_pri; auto &d_post = C.d_post; auto &d_seg = C.d_seg;
  auto &d_alpha = C.d_alpha; auto &d_beta = C.d_beta; auto &d_tot = C.d_tot; auto &d_ac = C.d_ac; auto &d_num = C.d_num;
  auto &d_as = C.d_as; auto &d_bs = C.d_bs; auto &d_score = C.d_score; auto &d_bscore = C.d_bscore; auto &d_part = C.d_part;
  auto &d_keys = C.d_keys; auto &d_keys2 = C.d_keys2; auto &d_sort = C.d_sort;
  std::vector<int32_t> h_ali(num_ali, num_ali + total_rows), h_row_off(num_ali_offsets, num_ali_offsets + n_lats + 1),
      h_t2pdf(tid2pdf, tid2pdf + num_tids + 1);
  std::vector<float> h_w(eg_weights, eg_weights + n_lats), h_pri(priors, priors + P);
  if ((rc = d_ali.Upload(h_ali, side)) || (rc = d_row_off.Upload(h_row_off, side)) || (rc = d_t2pdf.Upload(h_t2pdf, side)) ||
      (rc = d_w.Upload(h_w, side)) || (rc = d_pri.Upload(h_pri, side)))
    return rc;
  if (!is_mmi) {
    std::vector<int32_t> h_t2ph(tid2phone, tid2phone + num_tids + 1), h_sil(silence_phones, silence_phones + n_sil);
    if ((rc = d_t2ph.Upload(h_t2ph, side)) || (rc = d_sil.Upload(h_sil, side))) return rc;
    KH_HIP(hipStreamSynchronize(side));   // (the vectors go out of scope)
  }
  if (d_ali_pdf.Alloc(total_rows) || d_has_num.Alloc(total_rows) || d_vals.Alloc(A) || d_vals2.Alloc(A) || d_keys.Alloc(A) ||
      d_keys2.Alloc(A) || d_post.Alloc(A) || d_seg.Alloc(A) || d_tot.Alloc(n_lats) || d_ac.Alloc(n_lats) || d_num.Alloc(n_lats) ||
      d_part.Alloc(3 * kEmitBlocks) ||
      d_sort.Alloc(static_cast<size_t>(kSortBins) * ((A + kSortTile - 1) / kSortTile + 1) + kSortBins))
    return KH_ENOMEM;
  // ---- beside the forward pass (side stream): the numerator's pdfs, the arcs' (row, pdf) keys and their stable sort;
  // arcs without a transition-id carry row = total_rows and end up last.  Only the bits that can be set take part: the
  // pdf's (the low word) and the row's (the high word).
  KH_HIP(hipMemsetAsync(d_has_num.p, 0, sizeof(int32_t) * (total_rows ? total_rows : 1), side));
  hipLaunchKernelGGL(AliPdfKernel, dim3(std::max(1, std::min(1024, (total_rows + 255) / 256))), dim3(256), 0, side, total_rows,
                     d_ali.p, d_t2pdf.p, d_ali_pdf.p);
  hipLaunchKernelGGL(PseudoLikeKernel, dim3(n_lats), dim3(kThreads), 0, side, B.d_descs.p, B.d_arc_off.p, B.d_ilabel.p, B.d_times.p,
                     d_row_off.p, d_t2pdf.p, posteriors, d_posteriors.stride, d_pri.p, acoustic_scale, total_rows, B.d_a.p,
                     d_keys.p, d_vals.p, 1);
  KH_LAUNCH_CHECK();
  int hi_bits = 1, lo_bits = 1;
  while (hi_bits < 31 && (static_cast<unsigned long long>(total_rows) >> hi_bits) != 0) hi_bits++;
  while (lo_bits < 31 && (static_cast<unsigned long long>(std::max(1, d_posteriors.cols - 1)) >> lo_bits) != 0) lo_bits++;
  int in_second = 0;
  if ((rc = SortPairs64(d_keys.p, d_vals.p, d_keys2.p, d_vals2.p, A, lo_bits, hi_bits, d_sort.p, side, &in_second))) return rc;
  if (overlap) {
    KH_HIP(hipStreamWaitEvent(side, C.ev_fwd, 0));
  } else if (side != st) {
    KH_HIP(hipEventRecord(side_ev, side));
    KH_HIP(hipStreamWaitEvent(st, side_ev, 0));
  }
  tm_side = since();
  // ---- behind the forward pass (the library's stream; overlapped: the side stream, behind the event)
  for (int r = 0; r < d_deriv.rows && d_deriv.stride != d_deriv.cols; r++)
    KH_HIP(hipMemsetAsync(deriv + static_cast<size_t>(r) * d_deriv.stride, 0, sizeof(float) * d_deriv.cols, ts));
  if (d_deriv.stride == d_deriv.cols)
    KH_HIP(hipMemsetAsync(deriv, 0, sizeof(float) * static_cast<size_t>(d_deriv.rows) * d_deriv.cols, ts));
  hipLaunchKernelGGL(PseudoLikeKernel, dim3(n_lats), dim3(kThreads), 0, ts, B.d_descs.p, B.d_arc_off.p, B.d_ilabel.p, B.d_times.p,
                     d_row_off.p, d_t2pdf.p, posteriors, d_posteriors.stride, d_pri.p, acoustic_scale, total_rows, B.d_a.p,
                     d_keys.p, d_vals.p, 2);
  KH_LAUNCH_CHECK();
  if (is_mmi) {
    hipLaunchKernelGGL(NumLikeKernel, dim3(n_lats), dim3(kThreads), 0, ts, d_row_off.p, d_ali.p, d_t2pdf.p, posteriors,
                       d_posteriors.stride, d_pri.p, acoustic_scale, d_num.p);
    if (d_alpha.Alloc(B.total_states) || d_beta.Alloc(B.total_states)) return KH_ENOMEM;
    if (B.has_levels) {
      hipLaunchKernelGGL(ForwardBackwardKernel, dim3(n_lats), dim3(kThreads), 0, ts, B.d_descs.p, B.d_arc_off.p, B.d_next.p,
                         B.d_g.p, B.d_a.p, B.d_fin.p, B.d_level_off.p, B.d_level_states.p, B.d_in_off.p, B.d_in_arc.p, B.d_in_src.p,
                         B.d_final_list.p, d_alpha.p, d_beta.p, d_post.p, d_tot.p, d_ac.p, log(DBL_EPSILON));
      KH_LAUNCH_CHECK();
    } else if ((rc = LaunchForwardBackwardDF(B, d_alpha.p, d_beta.p, d_post.p, d_tot.p, d_ac.p, log(DBL_EPSILON), ts))) {
      return rc;
    }
  } else {
    if ((rc = RunAlphaBeta(B, d_alpha, d_beta, d_tot, 0, ts))) return rc;
    if (d_as.Alloc(B.total_states) || d_bs.Alloc(B.total_states) || d_score.Alloc(n_lats) || d_bscore.Alloc(n_lats)) return KH_ENOMEM;
    MpeArgs m;
    m.tid2phone = d_t2ph.p; m.tid2pdf = d_t2pdf.p; m.sil = d_sil.p; m.num_ali = d_ali.p; m.ali_off = d_row_off.p;
    m.times = B.d_times.p; m.ilabel = B.d_ilabel.p; m.n_sil = n_sil; m.is_mpfe = criterion == 2; m.one_silence_class = one_silence_class;
    hipLaunchKernelGGL(MpeKernel, dim3(n_lats), dim3(kThreads), 0, ts, B.d_descs.p, B.d_arc_off.p, B.d_next.p, B.d_g.p, B.d_a.p,
                       B.d_fin.p, B.d_level_off.p, B.d_level_states.p, B.d_in_off.p, B.d_in_arc.p, B.d_in_src.p, B.d_final_list.p,
                       d_alpha.p, d_beta.p, d_tot.p, d_as.p, d_bs.p, m, d_post.p, d_score.p, d_bscore.p);
    KH_LAUNCH_CHECK();
  }
  const unsigned long long *s_keys = in_second ? d_keys2.p : d_keys.p;
  const int32_t *s_vals = in_second ? d_vals2.p : d_vals.p;
  const int seg_blocks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(4096, (A + 255) / 256)));
  hipLaunchKernelGGL(SegmentKernel, dim3(seg_blocks), dim3(256), 0, ts, A, s_keys, s_vals, B.d_ilabel.p, d_post.p, total_rows,
                     is_mmi ? -1.0f : 1.0f, is_mmi ? d_ali_pdf.p : nullptr, d_seg.p, d_has_num.p);
  KH_LAUNCH_CHECK();
  hipLaunchKernelGGL(EmitKernel, dim3(kEmitBlocks), dim3(kThreads), 0, ts, A, s_keys, d_seg.p, total_rows, is_mmi, drop_frames,
                     d_ali_pdf.p, d_has_num.p, d_row_off.p, n_lats, d_w.p, posteriors, d_posteriors.stride, deriv, d_deriv.stride,
                     d_part.p);
  KH_LAUNCH_CHECK();
  tm_launched = since();
  // (the copies to the host wait for the device: issued behind the last launch, into pinned memory - the call does not)
  double *h_part = C.pinned, *h_num = h_part + 3 * kEmitBlocks, *h_tot = h_num + n_lats, *h_beta0 = h_tot + n_lats,
         *h_fwd = h_beta0 + n_lats, *h_bscore = h_fwd + n_lats;
  KH_HIP(hipMemcpyAsync(h_part, d_part.p, sizeof(double) * 3 * kEmitBlocks, hipMemcpyDeviceToHost, ts));
  if (is_mmi) {
    KH_HIP(hipMemcpyAsync(h_num, d_num.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, ts));
    KH_HIP(hipMemcpyAsync(h_tot, d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, ts));
  } else {   // + the reference's forward / backward agreement checks (lattice-functions.cc:808, :909)
    KH_HIP(hipMemcpyAsync(h_tot, d_score.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, ts));
    KH_HIP(hipMemcpyAsync(h_fwd, d_tot.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, ts));
    KH_HIP(hipMemcpyAsync(h_bscore, d_bscore.p, sizeof(double) * n_lats, hipMemcpyDeviceToHost, ts));
    for (int l = 0; l < n_lats; l++)
      KH_HIP(hipMemcpyAsync(&h_beta0[l], d_beta.p + lat_state_offsets[l], sizeof(double), hipMemcpyDeviceToHost, ts));
  }
  return KH_OK;
}

// Waits for the call's device work and turns its results into the five statistics.
int DiscEnd(KhDiscCall &C, double *stats) {
  const int n_lats = C.n_lats;
  const double *h_part = C.pinned, *h_num = h_part + 3 * kEmitBlocks, *h_tot = h_num + n_lats, *h_beta0 = h_tot + n_lats,
               *h_fwd = h_beta0 + n_lats, *h_bscore = h_fwd + n_lats;
  KH_HIP(hipStreamSynchronize(C.tail));
  if (C.timing)
    fprintf(stderr, "[kh_lattice timing] lattices prepared %.2f, side stream loaded %.2f, everything launched %.2f, device done %.2f ms\n",
            C.tm_build, C.tm_side, C.tm_launched,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C.t_in).count());
  if (!C.is_mmi)
    for (int l = 0; l < n_lats; l++) {
      if (!ApproxEqualD(h_fwd[l], h_beta0[l], 1e-6)) {
        SetError("lattice %d: Total forward probability over lattice = %g, while total backward probability = %g", l, h_fwd[l], h_beta0[l]);
        return KH_ESTATE;
      }
      if (!ApproxEqualD(h_tot[l], h_bscore[l], 1e-4)) {
        SetError("lattice %d: Total forward score over lattice = %g, while total backward score = %g", l, h_tot[l], h_bscore[l]);
        return KH_ESTATE;
      }
    }
  double objf = 0.0, wsum = 0.0, pos = 0.0, num_objf = 0.0, den_objf = 0.0;
  for (int b = 0; b < kEmitBlocks; b++) { objf += h_part[3 * b]; wsum += h_part[3 * b + 1]; pos += h_part[3 * b + 2]; }
  for (int l = 0; l < n_lats; l++) {
    num_objf += static_cast<double>(C.w[l]) * h_num[l];     // :255
    den_objf += static_cast<double>(C.w[l]) * h_tot[l];     // :285
  }
  stats[0] = pos;        // tot_num_count :291-299
  stats[1] = num_objf;   // tot_num_objf (MMI)
  stats[2] = den_objf;   // tot_den_objf
  stats[3] = objf;       // CompObjfAndDeriv's tot_objf
  stats[4] = wsum;       // ... tot_weight
  return KH_OK;
}

int DiscriminativeImpl(
    int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets, const int32_t *arc_ilabel,
    const int32_t *arc_nextstate, const float *arc_graph, const float *arc_acoustic, const float *state_final,
    const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights, const int32_t *tid2pdf,
    const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion, float acoustic_scale,
    int drop_frames, int one_silence_class, const float *priors, const float *posteriors, KhMatrixDim d_posteriors,
    float *deriv, KhMatrixDim d_deriv, double *stats, bool labels_checked) {
  KH_CHECK_ARG(stats);
  KhDiscCall C;
  int rc = DiscBegin(C, n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, num_ali,
                     num_ali_offsets, eg_weights, tid2pdf, tid2phone, num_tids, silence_phones, n_sil, criterion, acoustic_scale,
                     drop_frames, one_silence_class, priors, posteriors, d_posteriors, deriv, d_deriv, labels_checked, false);
  if (!rc) rc = DiscEnd(C, stats);
  return rc;
}

// pinned staging of kh_discriminative_lattice_computations_parts (kept across calls, grown on demand)
struct PinnedStage {
  std::mutex mu;
  char *p = nullptr;
  size_t bytes = 0;
  int Reserve(size_t want) {
    if (want <= bytes) return KH_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
    want += want / 4;
    KH_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault));
    bytes = want;
    return KH_OK;
  }
};
PinnedStage g_stage;

}  // namespace

extern "C" int kh_discriminative_lattice_computations(
    int n_lats, const int32_t *lat_state_offsets, const int64_t *arc_offsets, const int32_t *arc_ilabel,
    const int32_t *arc_nextstate, const float *arc_graph, const float *arc_acoustic, const float *state_final,
    const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights, const int32_t *tid2pdf,
    const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion, float acoustic_scale,
    int drop_frames, int one_silence_class, const float *priors, const float *posteriors, KhMatrixDim d_posteriors,
    float *deriv, KhMatrixDim d_deriv, double *stats) {
  return DiscriminativeImpl(n_lats, lat_state_offsets, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final,
                            num_ali, num_ali_offsets, eg_weights, tid2pdf, tid2phone, num_tids, silence_phones, n_sil, criterion,
                            acoustic_scale, drop_frames, one_silence_class, priors, posteriors, d_posteriors, deriv, d_deriv, stats,
                            false);
}

// The lattices as they sit in the examples, one set of arrays per lattice (NnetDiscriminativeUpdater gets its examples one by
// one, nnet-compute-discriminative.cc:150-175) -> the batch arrays, assembled by a few host threads straight into pinned
// memory while the forward pass the caller has launched runs (round 5: 6 ms of numpy concatenation per 256 lattices,
// between a 10 ms forward pass and 3.5 ms of lattice work).  The caller holds g_stage.mu until the uploads have completed.
namespace {
struct BatchArrays {
  int32_t *soff, *il, *ns;
  int64_t *aoff;
  float *g, *a, *fin;
};
int AssembleParts(int n_lats, const int32_t *n_states, const int64_t *const *arc_offsets, const int32_t *const *arc_ilabel,
                  const int32_t *const *arc_nextstate, const float *const *arc_graph, const float *const *arc_acoustic,
                  const float *const *state_final, int num_tids, BatchArrays *out) {
  const auto t_parts = std::chrono::steady_clock::now();
  KH_CHECK_ARG(n_lats > 0 && n_states && arc_offsets && arc_ilabel && arc_nextstate && arc_graph && arc_acoustic && state_final);
  std::vector<int64_t> sbase(n_lats + 1, 0), abase(n_lats + 1, 0);
  for (int l = 0; l < n_lats; l++) {
    KH_CHECK_ARG(n_states[l] > 0 && arc_offsets[l] && arc_ilabel[l] && arc_nextstate[l] && arc_graph[l] && arc_acoustic[l] &&
                 state_final[l] && arc_offsets[l][0] == 0 && arc_offsets[l][n_states[l]] >= 0);
    sbase[l + 1] = sbase[l] + n_states[l];
    abase[l + 1] = abase[l] + arc_offsets[l][n_states[l]];
  }
  const int64_t S = sbase[n_lats], A = abase[n_lats];
  KH_CHECK_ARG(S < (1ll << 31));
  auto up = [](size_t b) { return (b + 63) & ~static_cast<size_t>(63); };
  const size_t o_soff = 0, o_aoff = o_soff + up(sizeof(int32_t) * (n_lats + 1)), o_il = o_aoff + up(sizeof(int64_t) * (S + 1)),
               o_ns = o_il + up(sizeof(int32_t) * A), o_g = o_ns + up(sizeof(int32_t) * A), o_a = o_g + up(sizeof(float) * A),
               o_fin = o_a + up(sizeof(float) * A), total = o_fin + up(sizeof(float) * S);
  int rc = g_stage.Reserve(total);
  if (rc) return rc;
  int32_t *soff = reinterpret_cast<int32_t *>(g_stage.p + o_soff), *il = reinterpret_cast<int32_t *>(g_stage.p + o_il),
          *ns = reinterpret_cast<int32_t *>(g_stage.p + o_ns);
  int64_t *aoff = reinterpret_cast<int64_t *>(g_stage.p + o_aoff);
  float *g = reinterpret_cast<float *>(g_stage.p + o_g), *a = reinterpret_cast<float *>(g_stage.p + o_a),
        *fin = reinterpret_cast<float *>(g_stage.p + o_fin);
  for (int l = 0; l <= n_lats; l++) soff[l] = static_cast<int32_t>(sbase[l]);
  aoff[0] = 0;
  std::atomic<int> next{0}, bad{-1};
  auto work = [&]() {
    for (;;) {
      const int l = next.fetch_add(1);
      if (l >= n_lats) break;
      const int64_t s0 = sbase[l], a0 = abase[l], na = abase[l + 1] - abase[l];
      const int nsl = n_states[l];
      const int64_t *o = arc_offsets[l];
      bool ok = true;
      for (int i = 1; i <= nsl; i++) { aoff[s0 + i] = a0 + o[i]; ok &= o[i] >= o[i - 1] && o[i] <= na; }
      const int32_t *li = arc_ilabel[l];
      for (int64_t k = 0; k < na; k++) { const int32_t v = li[k]; il[a0 + k] = v; ok &= v >= 0 && v <= num_tids; }
      memcpy(ns + a0, arc_nextstate[l], sizeof(int32_t) * na);
      memcpy(g + a0, arc_graph[l], sizeof(float) * na);
      memcpy(a + a0, arc_acoustic[l], sizeof(float) * na);
      memcpy(fin + s0, state_final[l], sizeof(float) * nsl);
      if (!ok) bad.store(l);
    }
  };
  const int n_threads = std::max(1, std::min({4, n_lats, static_cast<int>(std::thread::hardware_concurrency())}));
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  if (bad.load() >= 0) {
    SetError("kh_discriminative_lattice_computations_parts: lattice %d: arc offsets not ascending, or an input label outside [0, %d]",
             bad.load(), num_tids);
    return KH_EINVAL;
  }
  if (getenv("KH_LATTICE_TIMING") != nullptr)
    fprintf(stderr, "[kh_lattice timing] batch assembled in pinned memory: %.2f ms (%d threads)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_parts).count(), n_threads);
  *out = BatchArrays{soff, il, ns, aoff, g, a, fin};
  return KH_OK;
}
}  // namespace

extern "C" int kh_discriminative_lattice_computations_parts(
    int n_lats, const int32_t *n_states, const int64_t *const *arc_offsets, const int32_t *const *arc_ilabel,
    const int32_t *const *arc_nextstate, const float *const *arc_graph, const float *const *arc_acoustic,
    const float *const *state_final, const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights,
    const int32_t *tid2pdf, const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion,
    float acoustic_scale, int drop_frames, int one_silence_class, const float *priors, const float *posteriors,
    KhMatrixDim d_posteriors, float *deriv, KhMatrixDim d_deriv, double *stats) {
  int rc = EnsureDevice();
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_stage.mu);
  BatchArrays b;
  if ((rc = AssembleParts(n_lats, n_states, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, num_tids, &b)))
    return rc;
  return DiscriminativeImpl(n_lats, b.soff, b.aoff, b.il, b.ns, b.g, b.a, b.fin, num_ali, num_ali_offsets, eg_weights, tid2pdf, tid2phone,
                            num_tids, silence_phones, n_sil, criterion, acoustic_scale, drop_frames, one_silence_class, priors,
                            posteriors, d_posteriors, deriv, d_deriv, stats, true);
}

// The same in two halves.  _begin: the batch is assembled, uploaded and prepared, and EVERY device step of the call is queued
// on a stream of the call's own - the steps that read `posteriors` behind an event recorded on the library's stream at the
// moment of the call (so: behind the forward pass the caller queued just before); the call returns without waiting.  What
// the caller queues on the library's stream afterwards - the NEXT batch's forward pass - runs beside this batch's sweeps.
// _end waits for the call's stream, fills stats[5] and destroys the call (also when it returns an error).  posteriors and
// deriv must stay valid until then; at most two calls should be in flight (they alternate between two streams).
extern "C" int kh_discriminative_lattice_computations_begin(
    int n_lats, const int32_t *n_states, const int64_t *const *arc_offsets, const int32_t *const *arc_ilabel,
    const int32_t *const *arc_nextstate, const float *const *arc_graph, const float *const *arc_acoustic,
    const float *const *state_final, const int32_t *num_ali, const int32_t *num_ali_offsets, const float *eg_weights,
    const int32_t *tid2pdf, const int32_t *tid2phone, int num_tids, const int32_t *silence_phones, int n_sil, int criterion,
    float acoustic_scale, int drop_frames, int one_silence_class, const float *priors, const float *posteriors,
    KhMatrixDim d_posteriors, float *deriv, KhMatrixDim d_deriv, KhDiscCall **call) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(call);
  *call = nullptr;
  std::lock_guard<std::mutex> lock(g_stage.mu);
  BatchArrays b;
  if ((rc = AssembleParts(n_lats, n_states, arc_offsets, arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final, num_tids, &b)))
    return rc;
  KhDiscCall *C = new KhDiscCall();
  rc = DiscBegin(*C, n_lats, b.soff, b.aoff, b.il, b.ns, b.g, b.a, b.fin, num_ali, num_ali_offsets, eg_weights, tid2pdf, tid2phone,
                 num_tids, silence_phones, n_sil, criterion, acoustic_scale, drop_frames, one_silence_class, priors, posteriors,
                 d_posteriors, deriv, d_deriv, true, true);
  if (rc) { delete C; return rc; }
  *call = C;
  return KH_OK;
}

extern "C" int kh_discriminative_lattice_computations_end(KhDiscCall *call, double *stats) {
  KH_CHECK_ARG(call && stats);
  const int rc = DiscEnd(*call, stats);
  delete call;
  return rc;
}
