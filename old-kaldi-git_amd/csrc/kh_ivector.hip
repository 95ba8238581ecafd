// kh_ivector.hip — online iVector extraction on gfx950.
//
// Replaces OnlineIvectorFeature (online2/online-ivector-feature.{h,cc}) in its deterministic
// mode — no silence weighting, use_most_recent_ivector = false, a fresh adaptation state per
// utterance — for a BATCH of utterances whose base features are resident in HBM:
//
//   base --------------------> splice (:383-398) -> LDA (:400-422) = lda_
//   base -> OnlineCmvn (feat/online-feature.cc:228-331, global stats) -> splice -> LDA = lda_normalized_
//   per frame:  DiagGmm::LogLikelihoods(lda_normalized_)  ->  VectorToPosteriorEntry
//               (hmm/posterior.cc:427-466) x posterior_scale  ->
//               OnlineIvectorEstimationStats::AccStats(lda_)  (ivector/ivector-extractor.cc:522-568)
//   every ivector_period frames: GetIvector = LinearCgd (:631-655, matrix/optimization.cc:453-565)
//   feature row t = iVector of estimation point t / period, first dimension - PriorOffset
//   (online-ivector-feature.cc:286-299).
//
// Mapping: the UBM scores are the batched DiagGmm kernel (MFMA GEMM), the LDA the MFMA GEMM with
// the offset column as bias; the sliding-window CMVN is one wave per utterance (a lane per
// dimension, running sums in double as the reference); the posterior entry is one wave per
// frame (softmax, num_gselect rounds of a wave arg-max, pruning by one lane); the statistics and
// the conjugate-gradient solves are sequential in time, so one workgroup per utterance keeps
// the quadratic term (S (S + 1) / 2 doubles = 40 KB for S = 100) in LDS and streams the
// per-Gaussian U_g / Sigma_g^-1 M_g rows (double, L2 / Infinity-Cache resident: 37 MB) through.
#include <cfloat>
#include <cmath>
#include <vector>

#include <functional>
#include <memory>
#include <queue>

#include "kh_common.h"

using namespace kh;

struct KhIvectorExtractor {
  KhIvectorConfig cfg;
  int sdim = 0, qdim = 0;
  float *lda = nullptr;        // [feat_dim][sdim] linear part, row-major
  float *lda_off = nullptr;    // [feat_dim]
  double *gstats = nullptr;    // [2][base_dim + 1]
  float *ubm_g = nullptr, *ubm_mi = nullptr, *ubm_iv = nullptr;
  double *U = nullptr;         // [I][qdim] packed lower triangle by rows (SpMatrix order)
  double *SiM = nullptr;       // [I][feat_dim][ivector_dim]
  int *n_exact = nullptr;      // LinearCgd fall-backs to the exact solve (device counter)
  double *eig_scratch = nullptr;   // kEigSlots x 2 x S x S: the fall-back's eigen-decomposition (IvSolveQuadraticProblem)
  int *eig_locks = nullptr;
};

namespace {

// ---- sliding-window CMVN with global-stats smoothing: one wave per utterance, lane = dimension
__global__ void __launch_bounds__(64)
IvCmvnKernel(const float *__restrict__ x, int x_stride, const int32_t *__restrict__ utt_off, int D,
             const double *__restrict__ gstats, int cmn_window, int speaker_frames, int global_frames, int norm_mean,
             int norm_var, float *__restrict__ y, int y_stride, const double *__restrict__ state_in,
             double *__restrict__ state_out, int state_dim) {
  const int u = blockIdx.x, d = threadIdx.x;
  const int b = utt_off[u], e = utt_off[u + 1];
  if (d >= D) return;
  const double g0 = gstats[d], g1 = gstats[D + 1 + d], gcount = gstats[D];
  // OnlineCmvnState::speaker_cmvn_stats of the speaker's previous utterances (adaptation state)
  double p0 = 0.0, p1 = 0.0, pcount = 0.0;
  if (state_in != nullptr) {
    const double *sp = state_in + static_cast<size_t>(u) * state_dim;
    p0 = sp[d]; p1 = sp[D + 1 + d]; pcount = sp[D];
  }
  double s0 = 0.0, s1 = 0.0, count = 0.0, t0 = 0.0, t1 = 0.0;
  for (int t = b; t < e; t++) {
    const float xf = x[static_cast<size_t>(t) * x_stride + d];
    const double xd = static_cast<double>(xf);
    s0 += xd;                       // ComputeStatsForFrame :238-255
    s1 += xd * xd;
    count += 1.0;
    t0 += xd;                       // GetState :333-356: every frame of the utterance
    t1 += xd * xd;
    const int prev = t - cmn_window;
    if (prev >= b) {
      const double pd = static_cast<double>(x[static_cast<size_t>(prev) * x_stride + d]);
      s0 -= pd;
      s1 -= pd * pd;
      count -= 1.0;
    }
    double a0 = s0, a1 = s1, c = count;
    if (c < cmn_window && pcount > 0.0) {   // SmoothOnlineCmvnStats :263-298: the speaker's stats first
      double from_speaker = cmn_window - c;
      if (from_speaker > speaker_frames) from_speaker = speaker_frames;
      if (from_speaker > pcount) from_speaker = pcount;
      if (from_speaker > 0.0) {
        const double f = from_speaker / pcount;
        a0 += f * p0;
        a1 += f * p1;
        c += f * pcount;
      }
    }
    if (c < cmn_window) {           // ... then the global ones
      double from_global = cmn_window - c;
      if (from_global > global_frames) from_global = global_frames;
      if (from_global > 0.0) {
        const double f = from_global / gcount;
        a0 += f * g0;
        a1 += f * g1;
        c += f * gcount;
      }
    }
    float out = xf;
    if (norm_mean) {                // ApplyCmvn transform/cmvn.cc:64-113
      const double mean = a0 / c;
      double scale = 1.0, offset = -mean;
      if (norm_var) {
        double var = a1 / c - mean * mean;
        if (var < 1.0e-20) var = 1.0e-20;
        scale = 1.0 / sqrt(var);
        offset = -(mean * scale);
      }
      out = xf * static_cast<float>(scale) + static_cast<float>(offset);
    }
    y[static_cast<size_t>(t) * y_stride + d] = out;
  }
  if (state_out != nullptr) {
    double *so = state_out + static_cast<size_t>(u) * state_dim;
    so[d] = p0 + t0;
    so[D + 1 + d] = p1 + t1;
    if (d == 0) { so[D] = pcount + (e - b); so[2 * D + 1] = 0.0; }
  }
}

// ---- OnlineSpliceFrames for a batch: rows clamp to their own utterance
__global__ void IvSpliceKernel(const float *__restrict__ x, int x_stride, const int32_t *__restrict__ row_utt,
                               const int32_t *__restrict__ utt_off, int rows, int D, int left, int right,
                               float *__restrict__ y, int y_stride) {
  const int nctx = left + right + 1;
  for (int t = blockIdx.x; t < rows; t += gridDim.x) {
    const int u = row_utt[t], b = utt_off[u], e = utt_off[u + 1];
    for (int c = threadIdx.x; c < nctx * D; c += blockDim.x) {
      const int k = c / D, d = c - k * D;
      int s = t + k - left;
      s = s < b ? b : (s >= e ? e - 1 : s);
      y[static_cast<size_t>(t) * y_stride + c] = x[static_cast<size_t>(s) * x_stride + d];
    }
  }
}

// ---- VectorToPosteriorEntry: one wave per frame
constexpr int kMaxPerLane = 32;   // num_gauss <= 2048
constexpr int kMaxGselect = 16;
__global__ void __launch_bounds__(256)
IvPosteriorKernel(const float *__restrict__ ll, int ll_stride, int rows, int num_gauss, int num_gselect, float min_post,
                  float posterior_scale, int32_t *__restrict__ post_idx, float *__restrict__ post_w) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= rows) return;
  const float *row = ll + static_cast<size_t>(t) * ll_stride;
  float v[kMaxPerLane];
  const int per = (num_gauss + 63) / 64;
  float mx = -INFINITY;
#pragma unroll 4
  for (int j = 0; j < kMaxPerLane; j++) {
    if (j >= per) break;
    const int g = j * 64 + lane;
    v[j] = g < num_gauss ? row[g] : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = kh_wave_max(mx);
  // ApplySoftMax (kaldi-vector.cc): sum += (data[i] = Exp(data[i] - max)) sequentially in the
  // reference; here per lane then over lanes (float, 1e-7 relative)
  float sum = 0.f;
#pragma unroll 4
  for (int j = 0; j < kMaxPerLane; j++) {
    if (j >= per) break;
    v[j] = (j * 64 + lane) < num_gauss ? expf(v[j] - mx) : -1.f;
    if (v[j] > 0.f) sum += v[j];
  }
  sum = kh_wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll 4
  for (int j = 0; j < kMaxPerLane; j++) {
    if (j >= per) break;
    if (v[j] >= 0.f) v[j] *= inv;
  }
  // the num_gselect largest, in decreasing order (nth_element + sort in the reference)
  const int G = num_gselect < num_gauss ? num_gselect : num_gauss;
  float sel_w[kMaxGselect];
  int sel_g[kMaxGselect];
  for (int k = 0; k < G; k++) {
    float best = -1.f;
    int best_g = 0x7fffffff;
#pragma unroll 4
    for (int j = 0; j < kMaxPerLane; j++) {
      if (j >= per) break;
      if (v[j] > best) { best = v[j]; best_g = j * 64 + lane; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int og = __shfl_xor(best_g, o, 64);
      if (ob > best || (ob == best && og < best_g)) { best = ob; best_g = og; }
    }
    sel_w[k] = best;
    sel_g[k] = best_g;
    if ((best_g & 63) == lane) v[best_g >> 6] = -1.f;   // taken
  }
  if (lane == 0) {
    int n = G;
    while (n > 1 && sel_w[n - 1] < min_post) n--;      // :452-453
    float tot = 0.0f;
    for (int k = 0; k < n; k++) tot += sel_w[k];
    const float inv_tot = 1.0f / tot;
    for (int k = 0; k < num_gselect; k++) {
      float w = 0.f;
      int g = 0;
      if (k < n) { w = (sel_w[k] * inv_tot) * posterior_scale; g = sel_g[k]; }   // :458-459, online-ivector-feature.cc:195-196
      post_idx[static_cast<size_t>(t) * num_gselect + k] = g;
      post_w[static_cast<size_t>(t) * num_gselect + k] = w;
    }
  }
}

// ---- statistics + conjugate gradient: one workgroup per utterance
template <int kNW = 4>
__device__ __forceinline__ double BlockSumD(double v, double *red) {
  v = kh_wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 64 * kNW) red[threadIdx.x >> 6] = v;   // callers: zero beyond thread S <= 64 kNW
  __syncthreads();
  double sum = red[0];
#pragma unroll
  for (int w = 1; w < kNW; w++) sum += red[w];
  return sum;
}

// two sums in one round (their shuffles interleave: a solve is a chain of these)
template <int kNW = 4>
__device__ __forceinline__ void BlockSum2D(double &a, double &b, double *red) {
  // (the same butterfly as before, lane exchanges on DPP / ds_swizzle: kh_common.h)
  a += kh_lane_xor_d<32>(a); b += kh_lane_xor_d<32>(b);
  a += kh_lane_xor_d<16>(a); b += kh_lane_xor_d<16>(b);
  a += kh_lane_xor_d<8>(a);  b += kh_lane_xor_d<8>(b);
  a += kh_lane_xor_d<4>(a);  b += kh_lane_xor_d<4>(b);
  a += kh_lane_xor_d<2>(a);  b += kh_lane_xor_d<2>(b);
  a += kh_lane_xor_d<1>(a);  b += kh_lane_xor_d<1>(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 64 * kNW) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = red[0];
  b = red[4];
#pragma unroll
  for (int w = 1; w < kNW; w++) { a += red[w]; b += red[4 + w]; }
}

// What a workgroup needs for LinearCgd's exact fall-back: the counter of the events (reported like the reference's
// KALDI_WARN) and a few scratch slots in device memory, one S x S matrix pair each, taken by whichever workgroup falls back
// (the event is rare: a slot per workgroup of a 2620-utterance launch would be 420 MB that nothing uses).
struct IvExact {
  int *n_fallback;
  double *scratch;   // [n_slots][2][S][S]
  int *locks;        // [n_slots]
  int n_slots;
};

// SolveQuadraticProblem<double> (matrix/sp-matrix.cc:659-734) as LinearCgd calls it (optimization.cc:546-563:
// SolverOptions("called-from-linearCGD"): K = 1e4, eps = 1e-40, optimize_delta, no diagonal preconditioning), x = x_orig on
// entry: H = U L U^T, eigenvalues floored at max(eps, l_max / K) (negative ones at 0 first: SymPosSemiDefEig), the step
// delta = U L~^-1 U^T (g - H x) is taken only if the auxiliary function g.x - x^T H x / 2 does not decrease.
// Eigen-decomposition: cyclic Jacobi with the round-robin ("tournament") ordering - n / 2 disjoint rotations per round, all
// lanes busy - on the full matrix in a scratch slot (device memory, L2-resident; LDS holds the packed statistics).
constexpr int kEigSlots = 16;
inline IvExact Exact(const KhIvectorExtractor *x) { return IvExact{x->n_exact, x->eig_scratch, x->eig_locks, kEigSlots}; }

template <int kNW = 4>
__device__ void IvSolveQuadraticProblem(const double *quad, const double *lin, double *xv, double *rv, double *pv, int S, double *red,
                                        const IvExact &ex) {
  __shared__ int s_slot;
  __shared__ double jc[128], js[128];
  __shared__ int jp[128], jq[128];
  const int t_id = threadIdx.x, nt = blockDim.x;
  if (t_id == 0) {
    int slot = -1;
    for (int tries = 0; slot < 0; tries++) {
      const int i = (blockIdx.x + tries) % ex.n_slots;
      if (atomicCAS(&ex.locks[i], 0, 1) == 0) slot = i;
      else if (tries % ex.n_slots == ex.n_slots - 1) __builtin_amdgcn_s_sleep(64);
    }
    s_slot = slot;
  }
  __syncthreads();
  double *A = ex.scratch + static_cast<size_t>(s_slot) * 2 * S * S, *U = A + static_cast<size_t>(S) * S;
  for (int i = t_id; i < S * S; i += nt) {
    const int r = i / S, c = i - r * S;
    A[i] = quad[r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r];
    U[i] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  const int n = (S + 1) & ~1, half = n >> 1;      // an odd dimension plays with a dummy index that never rotates
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, tot = 0.0;
    for (int i = t_id; i < S * S; i += nt) {
      const int r = i / S, c = i - r * S;
      const double v = A[i] * A[i];
      tot += v;
      if (r != c) off += v;
    }
    BlockSum2D<kNW>(off, tot, red);
    if (!(off > 1.0e-30 * tot)) break;
    for (int round = 0; round < n - 1; round++) {
      if (t_id < half) {
        int p = t_id == 0 ? n - 1 : (round + t_id) % (n - 1), q = t_id == 0 ? round : (round - t_id + (n - 1)) % (n - 1);
        if (p > q) { const int tmp = p; p = q; q = tmp; }
        double c = 1.0, sn = 0.0;
        if (q < S) {
          const double apq = A[p * S + q];
          if (apq != 0.0) {
            const double tau = (A[q * S + q] - A[p * S + p]) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            sn = t * c;
          }
        }
        jp[t_id] = p; jq[t_id] = q; jc[t_id] = c; js[t_id] = sn;
      }
      __syncthreads();
      for (int i = t_id; i < S * half; i += nt) {          // A <- A J, U <- U J (columns p, q of every row)
        const int k = i / half, pr = i - k * half, p = jp[pr], q = jq[pr];
        if (q >= S || js[pr] == 0.0) continue;
        const double c = jc[pr], sn = js[pr];
        const double ap = A[k * S + p], aq = A[k * S + q];
        A[k * S + p] = c * ap - sn * aq;
        A[k * S + q] = sn * ap + c * aq;
        const double up = U[k * S + p], uq = U[k * S + q];
        U[k * S + p] = c * up - sn * uq;
        U[k * S + q] = sn * up + c * uq;
      }
      __syncthreads();
      for (int i = t_id; i < S * half; i += nt) {          // A <- J^T A (rows p, q of every column)
        const int k = i / half, pr = i - k * half, p = jp[pr], q = jq[pr];
        if (q >= S || js[pr] == 0.0) continue;
        const double c = jc[pr], sn = js[pr];
        const double ap = A[p * S + k], aq = A[q * S + k];
        A[p * S + k] = c * ap - sn * aq;
        A[q * S + k] = sn * ap + c * aq;
      }
      __syncthreads();
    }
  }
  // eigenvalues on the diagonal; floor; gbar = g - H x; delta = U L~^-1 U^T gbar
  double lmax = 0.0;
  for (int i = 0; i < S; i++) lmax = fmax(lmax, A[i * S + i]);
  const double fl = fmax(static_cast<double>(1.0e-40f), lmax / 1.0e4);
  auto spmv = [&](const double *vec, int s_) {
    double acc = 0.0;
    const int rs = s_ * (s_ + 1) / 2;
    for (int c = 0; c < S; c++) acc += quad[c <= s_ ? rs + c : c * (c + 1) / 2 + s_] * vec[c];
    return acc;
  };
  if (t_id < S) rv[t_id] = lin[t_id] - spmv(xv, t_id);     // gbar
  __syncthreads();
  if (t_id < S) {
    double acc = 0.0;
    for (int k = 0; k < S; k++) acc += U[k * S + t_id] * rv[k];
    double l = A[t_id * S + t_id];
    if (l < 0.0) l = 0.0;
    if (l < fl) l = fl;
    pv[t_id] = acc / l;
  }
  __syncthreads();
  double xh = 0.0;
  if (t_id < S) {
    double acc = 0.0;
    for (int i = 0; i < S; i++) acc += U[t_id * S + i] * pv[i];
    xh = xv[t_id] + acc;                                     // xhat = x + delta
  }
  __syncthreads();
  if (t_id < S) rv[t_id] = xh;
  __syncthreads();
  // auxf(x) = g.x - 0.5 x^T H x before and after
  double b1 = 0.0, b2 = 0.0;
  if (t_id < S) {
    b1 = lin[t_id] * xv[t_id] - 0.5 * xv[t_id] * spmv(xv, t_id);
    b2 = lin[t_id] * rv[t_id] - 0.5 * rv[t_id] * spmv(rv, t_id);
  }
  BlockSum2D<kNW>(b1, b2, red);
  if (!(b2 < b1) && t_id < S) xv[t_id] = rv[t_id];           // "Reject change" otherwise: x stays x_orig
  __syncthreads();
  if (t_id == 0) atomicExch(&ex.locks[s_slot], 0);
}

// GetIvector :631-655 -> LinearCgd matrix/optimization.cc:453-565 (max_error 0, recompute factor 0.01)
// on the workgroup's LDS copy of the statistics: quad (packed lower triangle by rows), lin; xv = the
// previous estimate on entry (current_ivector_), the new one on return.
// (forced inline: as a called function its LDS arguments are generic pointers and every access a FLAT
// instruction - the solve ran at a third of the speed)
template <int kNW = 4>
__device__ __forceinline__ void IvGetIvector(const double *quad, const double *lin, double *xv, double *rv, double *pv, double *x0, int S,
                             int cg_iters, double prior_offset, bool have_frames, double *red, const IvExact &ex) {
  const int t_id = threadIdx.x;
  // (A vec)[s], A = quad (symmetric, packed lower triangle).  ONE loop of S trips for every lane - element
  // (s, c) sits at tri(max) + min - with four independent LDS reads in flight: as two loops of s + 1 and
  // S - s - 1 trips the lanes of a wave diverged (a wave ran ~160 trips) and every trip waited for its own
  // read, 8 us per product where the solve is a chain of them.  Same products, same order of additions.
  auto spmv = [&](const double *vec, int s) {
    double acc = 0.0;
    const int rs = s * (s + 1) / 2;
    int c = 0;
    for (; c + 4 <= S; c += 4) {
      double q[4], v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int cc = c + j;
        q[j] = quad[cc <= s ? rs + cc : cc * (cc + 1) / 2 + s];
        v[j] = vec[cc];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) acc += q[j] * v[j];
    }
    for (; c < S; c++) acc += quad[c <= s ? rs + c : c * (c + 1) / 2 + s] * vec[c];
    return acc;
  };
      if (have_frames) {
        if (t_id == 0 && xv[0] == 0.0) xv[0] = prior_offset;
        __syncthreads();
        if (t_id < S) x0[t_id] = xv[t_id];
        // pass 0: LinearCgd with max_iters = num_cg_iters.  If the squared residual got worse
        // (:546-547: "Will do an exact optimization"): SolveQuadraticProblem from x_orig
        // (IvSolveQuadraticProblem above; round 2 ran CG to convergence here - the same solution only when
        // no eigenvalue is floored, cond(A) <= 1e4 - which remains the path of models wider than 256).
        for (int pass = 0; pass < 2; pass++) {
          const int max_iters = pass == 0 ? cg_iters : -1;
          double my_p = 0.0, my_r = 0.0;
          if (t_id < S) {
            my_p = lin[t_id] - spmv(xv, t_id);   // p_0 = b - A x_0
            my_r = -my_p;
          }
          __syncthreads();
          if (t_id < S) { pv[t_id] = my_p; rv[t_id] = my_r; }
          double r_cur = BlockSumD<kNW>(t_id < S ? my_r * my_r : 0.0, red);
          const double r_init = r_cur;
          double r_recompute = r_cur;
          const double rf = 0.01 * 0.01;
          for (int k = 0; k < S + 5 && k != max_iters; k++) {
            double ap = 0.0;
            if (t_id < S) ap = spmv(pv, t_id);
            double p_r = t_id < S ? pv[t_id] * rv[t_id] : 0.0, p_ap = t_id < S ? pv[t_id] * ap : 0.0;
            BlockSum2D<kNW>(p_r, p_ap, red);
            const double alpha = -p_r / p_ap;
            if (t_id < S) {
              xv[t_id] += alpha * pv[t_id];
              rv[t_id] += alpha * ap;
            }
            double r_next = BlockSumD<kNW>(t_id < S ? rv[t_id] * rv[t_id] : 0.0, red);
            if (r_next < rf * r_recompute || r_next > r_recompute / rf) {
              double nr = 0.0;
              if (t_id < S) nr = spmv(xv, t_id) - lin[t_id];
              __syncthreads();
              if (t_id < S) rv[t_id] = nr;
              r_next = BlockSumD<kNW>(t_id < S ? nr * nr : 0.0, red);
              r_recompute = r_next;
            }
            if (r_next <= DBL_MIN) break;
            const double beta = r_next / r_cur;
            if (t_id < S) pv[t_id] = -rv[t_id] + beta * pv[t_id];
            r_cur = r_next;
            __syncthreads();
          }
          if (pass == 1 || !(r_cur > r_init)) break;
          const double bb = BlockSumD<kNW>(t_id < S ? lin[t_id] * lin[t_id] : 0.0, red);
          if (!(r_cur > r_init + 1.0e-10 * bb)) break;
          __syncthreads();
          if (t_id < S) xv[t_id] = x0[t_id];
          if (t_id == 0 && ex.n_fallback) atomicAdd(ex.n_fallback, 1);
          __syncthreads();
          if (ex.scratch != nullptr && S <= 256) {   // the reference's exact optimisation (wider models: CG to convergence below)
            IvSolveQuadraticProblem<kNW>(quad, lin, xv, rv, pv, S, red, ex);
            break;
          }
        }
      } else if (t_id < S) {
        xv[t_id] = t_id == 0 ? prior_offset : 0.0;
      }
}

constexpr int kIvThreads = 256;
__global__ void __launch_bounds__(kIvThreads)
IvStatsKernel(const float *__restrict__ F, int f_stride, const int32_t *__restrict__ utt_off,
              const int32_t *__restrict__ post_idx, const float *__restrict__ post_w, int num_gselect, int D, int S, int qdim,
              const double *__restrict__ U, const double *__restrict__ SiM, double prior_offset, double max_count,
              int period, int cg_iters, float *__restrict__ out, int out_stride, IvExact n_fallback) {
  extern __shared__ double lds[];
  double *quad = lds;             // [qdim] packed lower triangle by rows
  double *lin = quad + qdim;      // [S]
  double *xv = lin + S, *rv = xv + S, *pv = rv + S, *x0 = pv + S, *feat = x0 + S, *part = feat + D;   // [S] x 4, [D], [num_gselect][S]
  __shared__ double red[8];
  const int u = blockIdx.x, t_id = threadIdx.x;
  const int b = utt_off[u], e = utt_off[u + 1];
  // OnlineIvectorEstimationStats ctor :685-694
  for (int k = t_id; k < qdim; k += kIvThreads) quad[k] = 0.0;
  __syncthreads();
  if (t_id < S) {
    quad[t_id * (t_id + 1) / 2 + t_id] = 1.0;
    lin[t_id] = t_id == 0 ? prior_offset : 0.0;
    xv[t_id] = t_id == 0 ? prior_offset : 0.0;   // current_ivector_ :358-359
  }
  double num_frames = 0.0;
  __syncthreads();
  for (int t = b; t < e; t++) {
    if (t_id < D) feat[t_id] = static_cast<double>(F[static_cast<size_t>(t) * f_stride + t_id]);
    __syncthreads();
    // AccStats :522-568: the frame's postings together, the loads of the U_g / Sigma_g^-1 M_g rows
    // independent of one another (kIvThreads x num_gselect in flight)
    double tot_weight = 0.0;
    const int32_t *pi = post_idx + static_cast<size_t>(t) * num_gselect;
    const float *pw = post_w + static_cast<size_t>(t) * num_gselect;
    for (int idx = t_id; idx < num_gselect * S; idx += kIvThreads) {
      const int k = idx / S, s = idx - k * S;
      const double w = static_cast<double>(pw[k]);
      double acc = 0.0;
      if (w != 0.0) {
        const double *m = SiM + (static_cast<size_t>(pi[k]) * D) * S + s;
        for (int d = 0; d < D; d++) acc += m[static_cast<size_t>(d) * S] * feat[d];
      }
      part[idx] = w * acc;
    }
    for (int q = t_id; q < qdim; q += kIvThreads) {
      double acc = quad[q];
      for (int k = 0; k < num_gselect; k++) {
        const double w = static_cast<double>(pw[k]);
        if (w != 0.0) acc += w * U[static_cast<size_t>(pi[k]) * qdim + q];
      }
      quad[q] = acc;
    }
    for (int k = 0; k < num_gselect; k++) tot_weight += static_cast<double>(pw[k]);
    __syncthreads();
    if (t_id < S) {
      double acc = lin[t_id];
      for (int k = 0; k < num_gselect; k++)
        if (pw[k] != 0.f) acc += part[k * S + t_id];
      lin[t_id] = acc;
    }
    if (max_count > 0.0) {
      const double old_scale = fmax(num_frames, max_count) / max_count,
                   new_scale = fmax(num_frames + tot_weight, max_count) / max_count, change = new_scale - old_scale;
      if (change != 0.0) {
        __syncthreads();
        if (t_id == 0) lin[0] += prior_offset * change;
        if (t_id < S) quad[t_id * (t_id + 1) / 2 + t_id] += change;
      }
    }
    num_frames += tot_weight;
    __syncthreads();
    if ((t - b) % period == 0) {
      IvGetIvector(quad, lin, xv, rv, pv, x0, S, cg_iters, prior_offset, num_frames > 0.0, red, n_fallback);
      __syncthreads();
      // the rows of this estimation point: frames [t, t + period)
      const int last = (t + period < e) ? t + period : e;
      for (int i = t_id; i < (last - t) * S; i += kIvThreads) {
        const int row = t + i / S, s = i - (i / S) * S;
        out[static_cast<size_t>(row) * out_stride + s] = static_cast<float>(xv[s] - (s == 0 ? prior_offset : 0.0));
      }
      __syncthreads();
    }
  }
}


// =====================================================================================
// Batched statistics (the default path).  The quadratic term of an estimation point is LINEAR in
// the per-Gaussian counts accumulated so far, quad = I + sum_g gamma_g U_g, and the linear term
// is a sum over the postings of Sigma_g^-1 M_g^T f_t — so instead of streaming a U_g row (40 KB)
// and a Sigma_g^-1 M_g block (32 KB) per posting through one workgroup per utterance (360 KB per
// frame out of L2 / Infinity Cache: that kernel was bound by it at 3.3 TB/s), the batch does
//   (1) IvGammaKernel: cumulative counts per estimation point  Gc [points x I]       (double)
//   (2) IvGemmF64Kernel: Quad [points x S(S+1)/2] = Gc x U      — ONE fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64)
//   (3) postings grouped by Gaussian (histogram + scan + scatter) and IvLinKernel: a workgroup per
//       Gaussian keeps Sigma_g^-1 M_g in LDS and writes y = w Sigma_g^-1 M_g^T f for its postings
//   (4) IvSolveKernel: one workgroup per utterance walks its estimation points: loads Quad, adds the
//       y rows of the period to the linear term, runs the conjugate gradient (sequential in time:
//       each solve starts from the previous estimate).
// Sums run in a different order from AccStats (double: 1e-16 relative).
constexpr int kGemmM = 64, kGemmN = 128, kGemmK = 16;
typedef double KhDouble4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
IvGammaKernel(const int32_t *__restrict__ utt_off, const int32_t *__restrict__ point_off, const int32_t *__restrict__ post_idx,
              const float *__restrict__ post_w, int G, int I, int period, double *__restrict__ Gc,
              const double *__restrict__ gamma_in, double *__restrict__ gamma_out, int state_dim) {
  __shared__ double gam[64 * kMaxPerLane];
  const int u = blockIdx.x, b = utt_off[u], e = utt_off[u + 1];
  // (the counts the speaker's previous utterances left in the adaptation state)
  for (int i = threadIdx.x; i < I; i += 256) gam[i] = gamma_in ? gamma_in[static_cast<size_t>(u) * state_dim + i] : 0.0;
  __syncthreads();
  int prev = b - 1;
  // period <= 0: use_most_recent_ivector + greedy_ivector_extractor (--online=false): ONE estimation
  // point per utterance, at its last frame
  const int step = period > 0 ? period : (e - b);
  for (int t = period > 0 ? b : e - 1, p = point_off[u]; t < e; t += step, p++) {
    // the postings of frames (prev, t]
    const long long n = static_cast<long long>(t - prev) * G;
    for (long long j = threadIdx.x; j < n; j += 256) {
      const long long q = static_cast<long long>(prev + 1) * G + j;
      const float w = post_w[q];
      if (w != 0.f) atomicAdd(&gam[post_idx[q]], static_cast<double>(w));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < I; i += 256) Gc[static_cast<size_t>(p) * I + i] = gam[i];
    __syncthreads();
    prev = t;
  }
  if (gamma_out != nullptr) {   // the frames behind the last estimation point belong to the state too
    const long long n = static_cast<long long>(e - 1 - prev) * G;
    for (long long j = threadIdx.x; j < n; j += 256) {
      const long long q = static_cast<long long>(prev + 1) * G + j;
      const float w = post_w[q];
      if (w != 0.f) atomicAdd(&gam[post_idx[q]], static_cast<double>(w));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < I; i += 256) gamma_out[static_cast<size_t>(u) * state_dim + i] = gam[i];
  }
}

// C [M x N] = A [M x K] * B [K x N], row-major doubles; 256 threads = 4 waves, block tile 64 x 128,
// a wave owns 64 x 32 = 4 x 2 MFMA tiles of 16 x 16 (lane l: A[row l & 15][k = l >> 4],
// B[k = l >> 4][col l & 15]; result register r: row (l >> 4) + 4 r, col l & 15).
__global__ void __launch_bounds__(256)
IvGemmF64Kernel(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C, int M, int N, int K) {
  __shared__ double sa[kGemmK][kGemmM + 1];   // [k][m]
  __shared__ double sb[kGemmK][kGemmN + 1];   // [k][n]
  const int bm = blockIdx.y * kGemmM, bn = blockIdx.x * kGemmN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // rows of A and B start on 16-byte boundaries (even K and N, aligned bases): pairs of doubles can be loaded at once
  const bool wide = (K & 1) == 0 && (N & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  KhDouble4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = KhDouble4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += kGemmK) {
    // A tile 64 x 16: thread -> (m = tid / 4, 4 consecutive k); B tile 16 x 128: thread -> (k = tid / 16, 8 consecutive n).
    // Interior pieces come as 16-byte loads (2 + 4 per thread instead of 12 eight-byte ones: a vector-memory instruction
    // costs the SIMD ~70 cycles of issue in which no MFMA starts, tools/gemm_lab.hip).
    typedef double KhDouble2 __attribute__((ext_vector_type(2)));
    {
      const int m = tid >> 2, kk = (tid & 3) * 4;
      const int gm = bm + m, gk0 = k0 + kk;
      if (wide && gm < M && gk0 + 3 < K) {
        const KhDouble2 *p2 = reinterpret_cast<const KhDouble2 *>(A + static_cast<size_t>(gm) * K + gk0);
        const KhDouble2 v0 = p2[0], v1 = p2[1];
        sa[kk + 0][m] = v0.x; sa[kk + 1][m] = v0.y; sa[kk + 2][m] = v1.x; sa[kk + 3][m] = v1.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int gk = gk0 + j;
          sa[kk + j][m] = (gm < M && gk < K) ? A[static_cast<size_t>(gm) * K + gk] : 0.0;
        }
      }
    }
    {
      const int kk = tid >> 4, n = (tid & 15) * 8;
      const int gk = k0 + kk, gn0 = bn + n;
      if (wide && gk < K && gn0 + 7 < N) {
        const KhDouble2 *p2 = reinterpret_cast<const KhDouble2 *>(B + static_cast<size_t>(gk) * N + gn0);
        KhDouble2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = p2[j];
#pragma unroll
        for (int j = 0; j < 4; j++) { sb[kk][n + 2 * j] = v[j].x; sb[kk][n + 2 * j + 1] = v[j].y; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int gn = gn0 + j;
          sb[kk][n + j] = (gk < K && gn < N) ? B[static_cast<size_t>(gk) * N + gn] : 0.0;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGemmK; ks += 4) {
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
        if (gm < M && gn < N) C[static_cast<size_t>(gm) * N + gn] = acc[i][j][r];
      }
}

// postings grouped by Gaussian: counts, then positions
__global__ void __launch_bounds__(256)
IvCountKernel(const int32_t *__restrict__ post_idx, const float *__restrict__ post_w, long long n, int I, int *__restrict__ count) {
  __shared__ int h[64 * kMaxPerLane];
  for (int i = threadIdx.x; i < I; i += 256) h[i] = 0;
  __syncthreads();
  const long long per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (long long q = lo + threadIdx.x; q < hi; q += 256)
    if (post_w[q] != 0.f) atomicAdd(&h[post_idx[q]], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < I; i += 256)
    if (h[i]) atomicAdd(&count[i], h[i]);
}
// positions of the groups + the work items of IvLinKernel: a Gaussian's postings in pieces of kLinItem
// (the postings concentrate on few Gaussians: a workgroup per Gaussian would leave the chip idle)
constexpr int kLinItem = 512;
__global__ void __launch_bounds__(256) IvScanKernel(const int *__restrict__ count, int I, int *__restrict__ start, int *__restrict__ cursor,
                                                    int32_t *__restrict__ item_g, int32_t *__restrict__ item_b, int *__restrict__ n_items) {
  if (threadIdx.x == 0) {   // I <= 2048: a serial scan is nothing
    int run = 0, ni = 0;
    for (int i = 0; i < I; i++) {
      start[i] = run; cursor[i] = run;
      for (int o = 0; o < count[i]; o += kLinItem) { item_g[ni] = i; item_b[ni] = run + o; ni++; }
      run += count[i];
    }
    start[I] = run;
    *n_items = ni;
  }
}
__global__ void __launch_bounds__(256)
IvScatterKernel(const int32_t *__restrict__ post_idx, const float *__restrict__ post_w, long long n, int I, int *__restrict__ cursor,
                int32_t *__restrict__ sorted) {
  __shared__ int h[64 * kMaxPerLane], base[64 * kMaxPerLane];
  for (int i = threadIdx.x; i < I; i += 256) h[i] = 0;
  __syncthreads();
  const long long per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (long long q = lo + threadIdx.x; q < hi; q += 256)
    if (post_w[q] != 0.f) atomicAdd(&h[post_idx[q]], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < I; i += 256) { base[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0; h[i] = 0; }
  __syncthreads();
  for (long long q = lo + threadIdx.x; q < hi; q += 256)
    if (post_w[q] != 0.f) {
      const int g = post_idx[q];
      sorted[base[g] + atomicAdd(&h[g], 1)] = static_cast<int32_t>(q);   // (the order inside a group does not matter)
    }
}

// y[posting] = w * (Sigma_g^-1 M_g)^T f_t; a workgroup = one work item (up to kLinItem postings of one Gaussian)
constexpr int kLinTile = 32;
__global__ void __launch_bounds__(256)
IvLinKernel(const float *__restrict__ F, int f_stride, const float *__restrict__ post_w, int G, int D, int S,
            const double *__restrict__ SiM, const int *__restrict__ start, const int32_t *__restrict__ item_g,
            const int32_t *__restrict__ item_b, const int *__restrict__ n_items, const int32_t *__restrict__ sorted,
            double *__restrict__ Y) {
  extern __shared__ double lds[];
  double *m = lds;                 // [D][S]
  double *f = m + D * S;           // [kLinTile][D]
  __shared__ int32_t ids[kLinTile];
  __shared__ double ws[kLinTile];
  if (static_cast<int>(blockIdx.x) >= *n_items) return;
  const int g = item_g[blockIdx.x], b = item_b[blockIdx.x], e = min(b + kLinItem, start[g + 1]);
  for (int i = threadIdx.x; i < D * S; i += 256) m[i] = SiM[static_cast<size_t>(g) * D * S + i];
  for (int t0 = b; t0 < e; t0 += kLinTile) {
    const int nt = min(kLinTile, e - t0);
    __syncthreads();
    if (threadIdx.x < nt) {
      const int32_t q = sorted[t0 + threadIdx.x];
      ids[threadIdx.x] = q;
      ws[threadIdx.x] = static_cast<double>(post_w[q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * D; i += 256) {
      const int pp = i / D, d = i - pp * D;
      f[i] = static_cast<double>(F[static_cast<size_t>(ids[pp] / G) * f_stride + d]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * S; i += 256) {
      const int pp = i / S, sidx = i - pp * S;
      double acc = 0.0;
      for (int d = 0; d < D; d++) acc += m[d * S + sidx] * f[pp * D + d];
      Y[static_cast<size_t>(ids[pp]) * S + sidx] = ws[pp] * acc;
    }
  }
}

// one workgroup per utterance: estimation points in order.  (Measured: expanding the quadratic term
// to the full symmetric matrix — conflict-free columns, 80 KB, one workgroup per CU — gives the same
// rate as the packed triangle at three workgroups per CU: a solve is a chain of ~35 reductions.)
__global__ void __launch_bounds__(kIvThreads)
IvSolveKernel(const int32_t *__restrict__ utt_off, const int32_t *__restrict__ point_off, const float *__restrict__ post_w, int G, int S,
              int qdim, const double *__restrict__ Quad, const double *__restrict__ Y, double prior_offset, double max_count,
              int period, int cg_iters, float *__restrict__ out, int out_stride, IvExact n_fallback,
              const double *__restrict__ state_in, double *__restrict__ state_out, int state_dim, int lin_off,
              const int32_t *__restrict__ order) {
  extern __shared__ double lds[];
  constexpr int kT = kIvThreads;
  double *quad = lds;             // [qdim] packed lower triangle by rows
  double *lin = quad + qdim;      // [S]
  double *xv = lin + S, *rv = xv + S, *pv = rv + S, *x0 = pv + S;
  __shared__ double red[8];
  constexpr int kWChunk = 16, kYU = 10;   // frames per chunk of weights (kWChunk x num_gselect <= 256 floats); rows of Y in flight per lane
  __shared__ float s_w[kWChunk * 16];
  // longest utterance first: a workgroup's time is proportional to its utterance's length, and in list
  // order the long ones that start last set the kernel's duration
  const int u = order[blockIdx.x], t_id = threadIdx.x;
  const int b = utt_off[u], e = utt_off[u + 1];
  // adaptation state (SetAdaptationState :151-160): [lin_off - 2] = num_frames, [lin_off - 1] = the prior's
  // share of the quadratic diagonal, [lin_off ...) = the linear term; the counts went into Quad with IvGammaKernel
  const double *sin = state_in ? state_in + static_cast<size_t>(u) * state_dim : nullptr;
  if (t_id < S) {
    lin[t_id] = sin ? sin[lin_off + t_id] : (t_id == 0 ? prior_offset : 0.0);   // OnlineIvectorEstimationStats ctor :685-694
    xv[t_id] = t_id == 0 ? prior_offset : 0.0;    // current_ivector_ :358-359
  }
  double num_frames = sin ? sin[lin_off - 2] : 0.0, diag = sin ? sin[lin_off - 1] : 1.0;   // diag: 1, + the max_count rescaling :557-566
  int prev = b - 1;
  const int step = period > 0 ? period : (e - b);   // period <= 0: one point at the last frame, its iVector on every row
  for (int t = period > 0 ? b : e - 1, p = point_off[u]; t < e; t += step, p++) {
    // frames (prev, t]: linear term and counts.  A frame's weights decide which rows of Y are read: as a
    // loop "weight, branch, row" per posting this was a chain of dependent HBM round trips (50 per
    // estimation point, most of the kernel's time).  The weights of a chunk of frames go to LDS in one
    // round trip; the rows are then fetched kYU at a time (independent loads) and added in posting order.
    double lin0_add = 0.0;
    double acc = t_id < S ? lin[t_id] : 0.0;
    for (int c0 = prev + 1; c0 <= t; c0 += kWChunk) {
      const int nf = min(kWChunk, t + 1 - c0), nw = nf * G;
      __syncthreads();   // (the previous chunk's readers are done)
      for (int i = t_id; i < nw; i += kT) s_w[i] = post_w[static_cast<size_t>(c0) * G + i];
      __syncthreads();
      for (int f = 0; f < nf; f++) {
        double tot_weight = 0.0;
        for (int k = 0; k < G; k++) tot_weight += static_cast<double>(s_w[f * G + k]);
        if (max_count > 0.0) {
          const double old_scale = fmax(num_frames, max_count) / max_count,
                       new_scale = fmax(num_frames + tot_weight, max_count) / max_count, change = new_scale - old_scale;
          lin0_add += prior_offset * change;
          diag += change;
        }
        num_frames += tot_weight;
      }
      if (t_id < S) {
        const double *yb = Y + static_cast<size_t>(c0) * G * S + t_id;
        for (int q0 = 0; q0 < nw; q0 += kYU) {
          double y[kYU];
#pragma unroll
          for (int j = 0; j < kYU; j++) {
            const int q = q0 + j;
            y[j] = (q < nw && s_w[q] != 0.f) ? yb[static_cast<size_t>(q) * S] : 0.0;
          }
#pragma unroll
          for (int j = 0; j < kYU; j++)
            if (q0 + j < nw && s_w[q0 + j] != 0.f) acc += y[j];
        }
      }
    }
    if (t_id < S) {
      if (t_id == 0) acc += lin0_add;
      lin[t_id] = acc;
    }
    {
      // the point's quadratic term (40 KB at S = 100): eight independent loads per lane in flight (a loop
      // "load, store to LDS" per element was 20 dependent HBM round trips per estimation point)
      const double *qsrc = Quad + static_cast<size_t>(p) * qdim;
      for (int q0 = t_id; q0 < qdim; q0 += 8 * kT) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = q0 + j * kT < qdim ? qsrc[q0 + j * kT] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (q0 + j * kT < qdim) quad[q0 + j * kT] = v[j];
      }
      __syncthreads();
      if (t_id < S) quad[t_id * (t_id + 1) / 2 + t_id] += diag;
    }
    __syncthreads();
    IvGetIvector<kT / 64>(quad, lin, xv, rv, pv, x0, S, cg_iters, prior_offset, num_frames > 0.0, red, n_fallback);
    __syncthreads();
    const int first = period > 0 ? t : b, last = period > 0 ? ((t + period < e) ? t + period : e) : e;
    for (int i = t_id; i < (last - first) * S; i += kT) {
      const int row = first + i / S, sidx = i - (i / S) * S;
      out[static_cast<size_t>(row) * out_stride + sidx] = static_cast<float>(xv[sidx] - (sidx == 0 ? prior_offset : 0.0));
    }
    __syncthreads();
    prev = t;
  }
  if (state_out != nullptr) {   // GetAdaptationState :162-171 before LimitFrames: the frames behind the last point too
    double lin0_add = 0.0;
    for (int tt = prev + 1; tt < e; tt++) {
      double tot_weight = 0.0;
      for (int k = 0; k < G; k++) tot_weight += static_cast<double>(post_w[static_cast<size_t>(tt) * G + k]);
      if (max_count > 0.0) {
        const double old_scale = fmax(num_frames, max_count) / max_count,
                     new_scale = fmax(num_frames + tot_weight, max_count) / max_count, change = new_scale - old_scale;
        lin0_add += prior_offset * change;
        diag += change;
      }
      num_frames += tot_weight;
    }
    double *so = state_out + static_cast<size_t>(u) * state_dim;
    if (t_id < S) {
      double acc = lin[t_id];
      for (int tt = prev + 1; tt < e; tt++)
        for (int k = 0; k < G; k++) {
          const size_t q = static_cast<size_t>(tt) * G + k;
          if (post_w[q] != 0.f) acc += Y[q * S + t_id];
        }
      if (t_id == 0) acc += lin0_add;
      so[lin_off + t_id] = acc;
    }
    if (t_id == 0) { so[lin_off - 2] = num_frames; so[lin_off - 1] = diag; }
  }
}




// ---- OnlineIvectorFeature with frame weights (UpdateFrameWeights / UpdateStatsUntilFrameWeighted,
// online-ivector-feature.cc:155-254): one workgroup per stream and call, the stream's statistics live in
// device memory between calls (quadratic term packed, linear term, per-Gaussian counts, current iVector).
// A call runs the stream's PROGRAM: items (frame >= 0, delta weight) = UpdateStatsForFrame(frame, weight)
// (:172-189, AccStats ivector-extractor.cc:522-568 with the posteriors scaled by posterior_scale * weight,
// negative weights allowed), items (frame < 0) = GetIvector at estimation point -frame - 1 (:243-251), in the
// order the reference's loop over t produces them.  The per-frame inputs (lda_ features, pruned UBM posteriors)
// do not depend on the weights and are computed once per utterance.
__global__ void __launch_bounds__(kIvThreads)
IvStreamKernel(const int32_t *__restrict__ streams, const int32_t *__restrict__ prog_off, const int32_t *__restrict__ item_frame,
               const float *__restrict__ item_w, const int32_t *__restrict__ utt_off, const float *__restrict__ F, int f_stride,
               const int32_t *__restrict__ post_idx, const float *__restrict__ post_u, int num_gselect, int D, int S, int qdim, int I,
               const double *__restrict__ U, const double *__restrict__ SiM, double prior_offset, double max_count, float posterior_scale,
               int period, int cg_iters, double *__restrict__ g_quad, double *__restrict__ g_lin, double *__restrict__ g_xv,
               double *__restrict__ g_scal, double *__restrict__ g_cnt, float *__restrict__ out, int out_stride, IvExact n_fallback) {
  extern __shared__ double lds[];
  double *quad = lds;             // [qdim] packed lower triangle by rows
  double *lin = quad + qdim;      // [S]
  double *xv = lin + S, *rv = xv + S, *pv = rv + S, *x0 = pv + S, *feat = x0 + S, *part = feat + D;   // [S] x 4, [D], [num_gselect][S]
  __shared__ double red[8];
  __shared__ float s_w[16];
  __shared__ int s_g[16];
  const int u = streams[blockIdx.x], t_id = threadIdx.x;
  const int b = utt_off[u], e = utt_off[u + 1];
  double *q_u = g_quad + static_cast<size_t>(u) * qdim, *cnt_u = g_cnt + static_cast<size_t>(u) * I;
  for (int k = t_id; k < qdim; k += kIvThreads) quad[k] = q_u[k];
  if (t_id < S) { lin[t_id] = g_lin[static_cast<size_t>(u) * S + t_id]; xv[t_id] = g_xv[static_cast<size_t>(u) * S + t_id]; }
  double num_frames = g_scal[2 * u], diag = g_scal[2 * u + 1];   // diag: the prior's share of the quadratic diagonal (the state's layout)
  __syncthreads();
  for (int it = prog_off[blockIdx.x]; it < prog_off[blockIdx.x + 1]; it++) {
    const int fr = item_frame[it];
    if (fr >= 0) {
      const int t = b + fr;
      const float sc = posterior_scale * item_w[it];              // "posterior[i].second *= info_.posterior_scale * weight" :186
      if (t_id < D) feat[t_id] = static_cast<double>(F[static_cast<size_t>(t) * f_stride + t_id]);
      if (t_id < num_gselect) {
        s_w[t_id] = post_u[static_cast<size_t>(t) * num_gselect + t_id] * sc;
        s_g[t_id] = post_idx[static_cast<size_t>(t) * num_gselect + t_id];
      }
      __syncthreads();
      double tot_weight = 0.0;
      for (int idx = t_id; idx < num_gselect * S; idx += kIvThreads) {
        const int k = idx / S, sidx = idx - k * S;
        const double w = static_cast<double>(s_w[k]);
        double acc = 0.0;
        if (w != 0.0) {
          const double *m = SiM + (static_cast<size_t>(s_g[k]) * D) * S + sidx;
          for (int d = 0; d < D; d++) acc += m[static_cast<size_t>(d) * S] * feat[d];
        }
        part[idx] = w * acc;
      }
      for (int q = t_id; q < qdim; q += kIvThreads) {
        double acc = quad[q];
        for (int k = 0; k < num_gselect; k++) {
          const double w = static_cast<double>(s_w[k]);
          if (w != 0.0) acc += w * U[static_cast<size_t>(s_g[k]) * qdim + q];
        }
        quad[q] = acc;
      }
      for (int k = 0; k < num_gselect; k++) tot_weight += static_cast<double>(s_w[k]);
      if (t_id == 0)
        for (int k = 0; k < num_gselect; k++)
          if (s_w[k] != 0.f) cnt_u[s_g[k]] += static_cast<double>(s_w[k]);
      __syncthreads();
      if (t_id < S) {
        double acc = lin[t_id];
        for (int k = 0; k < num_gselect; k++)
          if (s_w[k] != 0.f) acc += part[k * S + t_id];
        lin[t_id] = acc;
      }
      if (max_count > 0.0) {
        const double old_scale = fmax(num_frames, max_count) / max_count,
                     new_scale = fmax(num_frames + tot_weight, max_count) / max_count, change = new_scale - old_scale;
        if (change != 0.0) {
          __syncthreads();
          if (t_id == 0) lin[0] += prior_offset * change;
          if (t_id < S) quad[t_id * (t_id + 1) / 2 + t_id] += change;
          diag += change;
        }
      }
      num_frames += tot_weight;
      __syncthreads();
    } else {
      const int point = -fr - 1, t0 = b + point * period;
      IvGetIvector(quad, lin, xv, rv, pv, x0, S, cg_iters, prior_offset, num_frames > 0.0, red, n_fallback);
      __syncthreads();
      const int last = (t0 + period < e) ? t0 + period : e;
      for (int i = t_id; i < (last - t0) * S; i += kIvThreads) {
        const int row = t0 + i / S, sidx = i - (i / S) * S;
        out[static_cast<size_t>(row) * out_stride + sidx] = static_cast<float>(xv[sidx] - (sidx == 0 ? prior_offset : 0.0));
      }
      __syncthreads();
    }
  }
  for (int k = t_id; k < qdim; k += kIvThreads) q_u[k] = quad[k];
  if (t_id < S) { g_lin[static_cast<size_t>(u) * S + t_id] = lin[t_id]; g_xv[static_cast<size_t>(u) * S + t_id] = xv[t_id]; }
  if (t_id == 0) { g_scal[2 * u] = num_frames; g_scal[2 * u + 1] = diag; }
}

// quadratic term of an adaptation state: diag on the diagonal added to sum_g count_g U_g (IvGemmF64Kernel)
__global__ void IvStreamDiagKernel(int n, int S, int qdim, const double *__restrict__ diag, double *__restrict__ quad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * S; i += gridDim.x * blockDim.x) {
    const int u = i / S, s = i - u * S;
    quad[static_cast<size_t>(u) * qdim + s * (s + 1) / 2 + s] += diag[u];
  }
}

template <class T>
T *Upload(const T *h, size_t n) {
  T *d = static_cast<T *>(PoolMalloc(sizeof(T) * (n ? n : 1)));
  if (d && n && hipMemcpy(d, h, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) {
    PoolFree(d);
    return nullptr;
  }
  return d;
}

}  // namespace

extern "C" {

KhIvectorExtractor *kh_ivector_extractor_create(const KhIvectorConfig *cfg, const float *lda_mat,
                                                const double *global_cmvn_stats, const float *ubm_gconsts,
                                                const float *ubm_means_invvars, const float *ubm_inv_vars, const double *M,
                                                const double *Sigma_inv) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (!cfg || !lda_mat || !global_cmvn_stats || !ubm_gconsts || !ubm_means_invvars || !ubm_inv_vars || !M || !Sigma_inv) {
    SetError("kh_ivector_extractor_create: bad arguments");
    return nullptr;
  }
  const KhIvectorConfig &c = *cfg;
  const int sdim = c.base_dim * (c.splice_left + c.splice_right + 1);
  // OnlineIvectorExtractionInfo::Check online-ivector-feature.cc:70-87
  if (!(c.base_dim > 0 && c.base_dim <= 64 && c.feat_dim > 0 && c.feat_dim <= 256 && c.num_gauss > 0 && c.num_gauss <= 64 * kMaxPerLane &&
        c.ivector_dim > 0 && c.ivector_dim <= 256 && (c.lda_cols == sdim || c.lda_cols == sdim + 1) && c.ivector_period > 0 &&
        c.num_gselect > 0 && c.num_gselect <= kMaxGselect && c.min_post < 0.5f && c.posterior_scale > 0.0f &&
        c.posterior_scale <= 1.0f && c.global_frames <= c.speaker_frames && c.speaker_frames <= c.cmn_window &&
        global_cmvn_stats[c.base_dim] > 0.0)) {
    SetError("kh_ivector_extractor_create: OnlineIvectorExtractionInfo::Check() failed");
    return nullptr;
  }
  {
    // The batched statistics kernels keep a Gaussian's Sigma^-1 M block (IvLinKernel) and an estimation point's packed
    // quadratic term (IvSolveKernel) in dynamic LDS, 64 KB per workgroup without an opt-in; a model beyond that is refused
    // HERE, with the limit named, instead of failing at extract time.  The reference has no such limit: the
    // shapes of online2 recipes (feat-dim 40, iVector-dim 100) need 55 KB / 44 KB.
    const size_t D_ = c.feat_dim, S_ = c.ivector_dim;
    const size_t lin_lds = sizeof(double) * (D_ * S_ + static_cast<size_t>(kLinTile) * D_),
                 solve_lds = sizeof(double) * (S_ * (S_ + 1) / 2 + 5 * S_);
    if (lin_lds > 64 * 1024 || solve_lds > 64 * 1024) {
      SetError("kh_ivector_extractor_create: feat-dim %d x ivector-dim %d needs %zu / %zu bytes of LDS per workgroup in the statistics / "
               "solve kernels; the limit is 65536 (e.g. feat-dim 40: ivector-dim <= 172; ivector-dim 100: feat-dim <= 62)",
               c.feat_dim, c.ivector_dim, lin_lds, solve_lds);
      return nullptr;
    }
  }
  KhIvectorExtractor *x = new KhIvectorExtractor();
  x->cfg = c;
  x->sdim = sdim;
  const int D = c.feat_dim, S = c.ivector_dim, I = c.num_gauss;
  x->qdim = S * (S + 1) / 2;
  std::vector<float> lin(static_cast<size_t>(D) * sdim), off(D, 0.f);
  for (int r = 0; r < D; r++) {
    for (int k = 0; k < sdim; k++) lin[static_cast<size_t>(r) * sdim + k] = lda_mat[static_cast<size_t>(r) * c.lda_cols + k];
    if (c.lda_cols == sdim + 1) off[r] = lda_mat[static_cast<size_t>(r) * c.lda_cols + sdim];   // OnlineTransform :400-415
  }
  // IvectorExtractor::ComputeDerivedVars(i) ivector-extractor.cc:207-217
  std::vector<double> U(static_cast<size_t>(I) * x->qdim), SiM(static_cast<size_t>(I) * D * S);
  for (int i = 0; i < I; i++) {
    const double *Mi = M + static_cast<size_t>(i) * D * S, *Si = Sigma_inv + static_cast<size_t>(i) * D * D;
    double *sm = SiM.data() + static_cast<size_t>(i) * D * S;
    for (int d = 0; d < D; d++)
      for (int s = 0; s < S; s++) {
        double acc = 0.0;
        for (int k = 0; k < D; k++) acc += Si[static_cast<size_t>(d) * D + k] * Mi[static_cast<size_t>(k) * S + s];
        sm[static_cast<size_t>(d) * S + s] = acc;
      }
    double *ui = U.data() + static_cast<size_t>(i) * x->qdim;
    for (int r = 0; r < S; r++)
      for (int cc = 0; cc <= r; cc++) {
        double acc = 0.0;
        for (int d = 0; d < D; d++) acc += Mi[static_cast<size_t>(d) * S + r] * sm[static_cast<size_t>(d) * S + cc];
        ui[r * (r + 1) / 2 + cc] = acc;
      }
  }
  x->lda = Upload(lin.data(), lin.size());
  x->lda_off = Upload(off.data(), off.size());
  x->gstats = Upload(global_cmvn_stats, 2 * static_cast<size_t>(c.base_dim + 1));
  x->ubm_g = Upload(ubm_gconsts, I);
  x->ubm_mi = Upload(ubm_means_invvars, static_cast<size_t>(I) * D);
  x->ubm_iv = Upload(ubm_inv_vars, static_cast<size_t>(I) * D);
  x->U = Upload(U.data(), U.size());
  x->SiM = Upload(SiM.data(), SiM.size());
  const int zero = 0;
  x->n_exact = Upload(&zero, 1);
  {
    const int S_ = cfg->ivector_dim;
    const std::vector<int> locks(kEigSlots, 0);
    x->eig_locks = Upload(locks.data(), locks.size());
    x->eig_scratch = static_cast<double *>(PoolMalloc(sizeof(double) * kEigSlots * 2 * static_cast<size_t>(S_) * S_));
  }
  if (!x->lda || !x->lda_off || !x->gstats || !x->ubm_g || !x->ubm_mi || !x->ubm_iv || !x->U || !x->SiM || !x->n_exact || !x->eig_locks ||
      !x->eig_scratch) {
    kh_ivector_extractor_destroy(x);
    SetError("kh_ivector_extractor_create: out of device memory");
    return nullptr;
  }
  return x;
}

void kh_ivector_extractor_destroy(KhIvectorExtractor *x) {
  if (!x) return;
  PoolFree(x->lda); PoolFree(x->lda_off); PoolFree(x->gstats); PoolFree(x->ubm_g); PoolFree(x->ubm_mi);
  PoolFree(x->ubm_iv); PoolFree(x->U); PoolFree(x->SiM); PoolFree(x->n_exact); PoolFree(x->eig_scratch); PoolFree(x->eig_locks);
  delete x;
}

int kh_ivector_state_dim(const KhIvectorExtractor *x) {
  if (!x) return 0;
  return 2 * (x->cfg.base_dim + 1) + 2 + x->cfg.ivector_dim + x->cfg.num_gauss;
}

int kh_ivector_extract(const KhIvectorExtractor *x, const float *feats, int feat_stride, const int32_t *utt_row_offsets_host,
                       int n_utts, float *ivectors, int ivector_stride) {
  return kh_ivector_extract_adapt(x, feats, feat_stride, utt_row_offsets_host, n_utts, nullptr, nullptr, ivectors, ivector_stride);
}

int kh_ivector_extract_adapt(const KhIvectorExtractor *x, const float *feats, int feat_stride, const int32_t *utt_row_offsets_host,
                             int n_utts, const double *state_in_host, double *state_out_host, float *ivectors, int ivector_stride) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(x && feats && utt_row_offsets_host && n_utts > 0 && ivectors && feat_stride >= x->cfg.base_dim &&
               ivector_stride >= x->cfg.ivector_dim && utt_row_offsets_host[0] == 0);
  const KhIvectorConfig &c = x->cfg;
  const int rows = utt_row_offsets_host[n_utts];
  for (int u = 0; u < n_utts; u++) KH_CHECK_ARG(utt_row_offsets_host[u + 1] > utt_row_offsets_host[u]);
  hipStream_t st = Stream();
  const int B = c.base_dim, D = c.feat_dim, S = c.ivector_dim, I = c.num_gauss, G = c.num_gselect;
  // adaptation state per utterance (doubles): CMVN speaker stats [2 (B + 1)], num_frames, the prior's share of the
  // quadratic diagonal, linear term [S], per-Gaussian counts [I] (quadratic term = diag * I + sum_g count_g U_g)
  const int state_dim = kh_ivector_state_dim(x), lin_off = 2 * (B + 1) + 2, gamma_off = lin_off + S;
  const bool adapt = state_in_host != nullptr || state_out_host != nullptr;
  double *d_sin = nullptr, *d_sout = nullptr;
  if (state_in_host) {
    d_sin = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * state_dim));
    if (!d_sin) return KH_ENOMEM;
    if (hipMemcpy(d_sin, state_in_host, sizeof(double) * static_cast<size_t>(n_utts) * state_dim, hipMemcpyHostToDevice) != hipSuccess) {
      PoolFree(d_sin);
      return KH_EDEVICE;
    }
  }
  if (state_out_host) {
    d_sout = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * state_dim));
    if (!d_sout) { PoolFree(d_sin); return KH_ENOMEM; }
  }
  const int sstride = (x->sdim + 3) & ~3, dstride = (D + 3) & ~3, istride = (I + 3) & ~3, bstride = (B + 3) & ~3;
  std::vector<int32_t> row_utt(rows);
  for (int u = 0; u < n_utts; u++)
    for (int t = utt_row_offsets_host[u]; t < utt_row_offsets_host[u + 1]; t++) row_utt[t] = u;
  int32_t *d_off = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * (n_utts + 1)));
  int32_t *d_row_utt = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * rows));
  float *d_norm = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * bstride));
  float *d_spl = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * sstride));
  float *d_F = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * dstride));
  float *d_Fn = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * dstride));
  float *d_ll = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * istride));
  int32_t *d_pi = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * static_cast<size_t>(rows) * G));
  float *d_pw = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * G));
  auto cleanup = [&]() {
    PoolFree(d_off); PoolFree(d_row_utt); PoolFree(d_norm); PoolFree(d_spl); PoolFree(d_F); PoolFree(d_Fn); PoolFree(d_ll);
    PoolFree(d_pi); PoolFree(d_pw);
  };
  if (!d_off || !d_row_utt || !d_norm || !d_spl || !d_F || !d_Fn || !d_ll || !d_pi || !d_pw) { cleanup(); PoolFree(d_sin); PoolFree(d_sout); return KH_ENOMEM; }
  rc = KH_OK;
  do {
    if (hipMemcpyAsync(d_off, utt_row_offsets_host, sizeof(int32_t) * (n_utts + 1), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_row_utt, row_utt.data(), sizeof(int32_t) * rows, hipMemcpyHostToDevice, st) != hipSuccess) { rc = KH_EDEVICE; break; }
    const KhMatrixDim dspl{rows, x->sdim, sstride}, dlda{D, x->sdim, x->sdim}, dF{rows, D, dstride};
    const int sgrid = std::min(rows, NumCUs() * 16);
    // lda_: splice(base) -> LDA
    hipLaunchKernelGGL(IvSpliceKernel, dim3(sgrid), dim3(256), 0, st, feats, feat_stride, d_row_utt, d_off, rows, B, c.splice_left,
                       c.splice_right, d_spl, sstride);
    if ((rc = kh_affine(d_spl, dspl, x->lda, dlda, x->lda_off, d_F, dF))) break;
    // lda_normalized_: cmvn(base) -> splice -> LDA
    hipLaunchKernelGGL(IvCmvnKernel, dim3(n_utts), dim3(64), 0, st, feats, feat_stride, d_off, B, x->gstats, c.cmn_window,
                       c.speaker_frames, c.global_frames, c.normalize_mean, c.normalize_variance, d_norm, bstride, d_sin, d_sout, state_dim);
    hipLaunchKernelGGL(IvSpliceKernel, dim3(sgrid), dim3(256), 0, st, d_norm, bstride, d_row_utt, d_off, rows, B, c.splice_left,
                       c.splice_right, d_spl, sstride);
    if ((rc = kh_affine(d_spl, dspl, x->lda, dlda, x->lda_off, d_Fn, dF))) break;
    // UBM log-likelihoods, posterior entries
    if ((rc = kh_diag_gmm_loglikes(d_Fn, dF, x->ubm_g, x->ubm_mi, x->ubm_iv, I, d_ll, istride))) break;
    hipLaunchKernelGGL(IvPosteriorKernel, dim3(DivUp(rows, 4)), dim3(256), 0, st, d_ll, istride, rows, I, G, c.min_post,
                       c.posterior_scale, d_pi, d_pw);
    // statistics + solves
    const size_t lin_lds = sizeof(double) * (static_cast<size_t>(D) * S + static_cast<size_t>(kLinTile) * D);
    if (!c.greedy_most_recent && !adapt && (getenv("KH_IVECTOR_SEQUENTIAL") || lin_lds > 64 * 1024)) {   // the per-utterance accumulation (A/B reference; very wide models)
      const size_t lds = sizeof(double) * (static_cast<size_t>(x->qdim) + 5 * S + D + static_cast<size_t>(G) * S);
      hipLaunchKernelGGL(IvStatsKernel, dim3(n_utts), dim3(kIvThreads), lds, st, d_F, dstride, d_off, d_pi, d_pw, G, D, S, x->qdim, x->U,
                         x->SiM, c.prior_offset, static_cast<double>(c.max_count), c.ivector_period, c.num_cg_iters, ivectors,
                         ivector_stride, Exact(x));
    } else {
      // estimation points per utterance; chunks of utterances bound the scratch (Quad: 8 qdim bytes per
      // point, y: 8 G S bytes per frame)
      std::vector<int32_t> poff(n_utts + 1, 0);
      for (int u = 0; u < n_utts; u++)
        poff[u + 1] = poff[u] + (c.greedy_most_recent ? 1 : (utt_row_offsets_host[u + 1] - utt_row_offsets_host[u] + c.ivector_period - 1) / c.ivector_period);
      int32_t *d_poff = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * (n_utts + 1)));
      int32_t *d_order = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * (n_utts + 1)));
      std::vector<int32_t> order(n_utts);
      int *d_cnt = static_cast<int *>(PoolMalloc(sizeof(int) * (3 * static_cast<size_t>(I) + 3)));
      if (!d_poff || !d_cnt || !d_order) { PoolFree(d_poff); PoolFree(d_cnt); PoolFree(d_order); rc = KH_ENOMEM; break; }
      if (hipMemcpyAsync(d_poff, poff.data(), sizeof(int32_t) * (n_utts + 1), hipMemcpyHostToDevice, st) != hipSuccess) rc = KH_EDEVICE;
      // scratch per chunk of utterances: <= 16 GB of Quad (8 qdim B per point) + 16 GB of y (8 G S B per frame) at the default sizes
      const long long kMaxRows = getenv("KH_IVECTOR_MAX_ROWS") ? atoll(getenv("KH_IVECTOR_MAX_ROWS")) : 4000000;
      const size_t solve_lds = sizeof(double) * (static_cast<size_t>(x->qdim) + 5 * S);
      for (int u0 = 0; u0 < n_utts && !rc;) {
        int u1 = u0 + 1;
        while (u1 < n_utts && utt_row_offsets_host[u1 + 1] - utt_row_offsets_host[u0] <= kMaxRows) u1++;
        const int row0 = utt_row_offsets_host[u0], rows_c = utt_row_offsets_host[u1] - row0;
        const int p0 = poff[u0], pts = poff[u1] - p0;
        const long long n_post = static_cast<long long>(rows_c) * G;
        // the chunk's utterances (indices relative to u0), longest first: the order the solve kernel's workgroups take them in
        for (int u = u0; u < u1; u++) order[u] = u - u0;
        std::stable_sort(order.begin() + u0, order.begin() + u1, [&](int32_t a, int32_t b) {
          return utt_row_offsets_host[u0 + a + 1] - utt_row_offsets_host[u0 + a] > utt_row_offsets_host[u0 + b + 1] - utt_row_offsets_host[u0 + b];
        });
        if (hipMemcpyAsync(d_order + u0, order.data() + u0, sizeof(int32_t) * (u1 - u0), hipMemcpyHostToDevice, st) != hipSuccess) rc = KH_EDEVICE;
        double *d_gc = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(pts) * I));
        double *d_quad = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(pts) * x->qdim));
        double *d_y = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_post) * S));
        int32_t *d_sorted = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * static_cast<size_t>(n_post)));
        const int max_items = static_cast<int>(n_post / kLinItem) + I + 1;
        int32_t *d_items = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * 2 * static_cast<size_t>(max_items)));
        if (!d_gc || !d_quad || !d_y || !d_sorted || !d_items) {
          rc = KH_ENOMEM;
        } else {
          int *d_count = d_cnt, *d_start = d_cnt + I, *d_cursor = d_cnt + 2 * I + 1, *d_nitems = d_cnt + 3 * I + 2;
          const int32_t *pi_c = d_pi + static_cast<size_t>(row0) * G;
          const float *pw_c = d_pw + static_cast<size_t>(row0) * G;
          hipLaunchKernelGGL(IvGammaKernel, dim3(u1 - u0), dim3(256), 0, st, d_off + u0, d_poff + u0, d_pi, d_pw, G, I, c.greedy_most_recent ? 0 : c.ivector_period,
                             d_gc - static_cast<ptrdiff_t>(p0) * I,
                             d_sin ? d_sin + static_cast<size_t>(u0) * state_dim + gamma_off : nullptr,
                             d_sout ? d_sout + static_cast<size_t>(u0) * state_dim + gamma_off : nullptr, state_dim);
          hipLaunchKernelGGL(IvGemmF64Kernel, dim3(DivUp(x->qdim, kGemmN), DivUp(pts, kGemmM)), dim3(256), 0, st, d_gc, x->U, d_quad, pts,
                             x->qdim, I);
          (void)hipMemsetAsync(d_count, 0, sizeof(int) * I, st);
          const int sort_grid = std::max(1, std::min<int>(NumCUs() * 4, static_cast<int>((n_post + 4095) / 4096)));
          hipLaunchKernelGGL(IvCountKernel, dim3(sort_grid), dim3(256), 0, st, pi_c, pw_c, n_post, I, d_count);
          hipLaunchKernelGGL(IvScanKernel, dim3(1), dim3(256), 0, st, d_count, I, d_start, d_cursor, d_items, d_items + max_items, d_nitems);
          hipLaunchKernelGGL(IvScatterKernel, dim3(sort_grid), dim3(256), 0, st, pi_c, pw_c, n_post, I, d_cursor, d_sorted);
          hipLaunchKernelGGL(IvLinKernel, dim3(max_items), dim3(256), lin_lds, st, d_F + static_cast<size_t>(row0) * dstride, dstride, pw_c, G, D,
                             S, x->SiM, d_start, d_items, d_items + max_items, d_nitems, d_sorted, d_y);
          hipLaunchKernelGGL(IvSolveKernel, dim3(u1 - u0), dim3(kIvThreads), solve_lds, st, d_off + u0, d_poff + u0, d_pw, G, S, x->qdim,
                           d_quad - static_cast<ptrdiff_t>(p0) * x->qdim, d_y - static_cast<ptrdiff_t>(row0) * G * S, c.prior_offset,
                           static_cast<double>(c.max_count), c.greedy_most_recent ? 0 : c.ivector_period, c.num_cg_iters, ivectors, ivector_stride, Exact(x),
                             d_sin ? d_sin + static_cast<size_t>(u0) * state_dim : nullptr,
                             d_sout ? d_sout + static_cast<size_t>(u0) * state_dim : nullptr, state_dim, lin_off, d_order + u0);
          if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { SetError("kh_ivector_extract: statistics kernels failed"); rc = KH_EDEVICE; }
        }
        PoolFree(d_gc); PoolFree(d_quad); PoolFree(d_y); PoolFree(d_sorted); PoolFree(d_items);
        u0 = u1;
      }
      (void)hipStreamSynchronize(st);
      PoolFree(d_poff); PoolFree(d_cnt); PoolFree(d_order);
      if (rc) break;
    }
    if (hipGetLastError() != hipSuccess) { SetError("kh_ivector_extract: kernel launch failed"); rc = KH_EDEVICE; break; }
  } while (0);
  hipError_t e = hipStreamSynchronize(st);   // the scratch returns to the pool
  cleanup();
  if (!rc && e == hipSuccess && d_sout)
    e = hipMemcpy(state_out_host, d_sout, sizeof(double) * static_cast<size_t>(n_utts) * state_dim, hipMemcpyDeviceToHost);
  PoolFree(d_sin);
  PoolFree(d_sout);
  int n_exact = 0;
  if (!rc && e == hipSuccess && (e = hipMemcpy(&n_exact, x->n_exact, sizeof(int), hipMemcpyDeviceToHost)) == hipSuccess && n_exact > 0) {
    // KALDI_WARN of LinearCgd, optimization.cc:548-552
    fprintf(stderr, "WARNING (kh_ivector_extract): linear CGD in dimension %d: the squared residual got worse at %d estimation points; did an exact optimization there\n", S, n_exact);
    (void)hipMemset(x->n_exact, 0, sizeof(int));
  }
  if (!rc && e != hipSuccess) { SetError("kh_ivector_extract: %s", hipGetErrorString(e)); rc = KH_EDEVICE; }
  return rc;
}


// ---- KhIvectorStreams: OnlineIvectorFeature objects of n utterances side by side, with frame weights --------
struct KhIvectorStreams {
  const KhIvectorExtractor *x = nullptr;
  int n = 0, dstride = 0, ivector_stride = 0;
  std::vector<int32_t> off;
  int32_t *d_off = nullptr, *d_pi = nullptr;
  float *d_F = nullptr, *d_pu = nullptr, *ivectors = nullptr;
  double *d_quad = nullptr, *d_lin = nullptr, *d_xv = nullptr, *d_scal = nullptr, *d_cnt = nullptr;
  // host side of the reference's object, per stream
  typedef std::pair<int32_t, float> Delta;
  std::vector<std::priority_queue<Delta, std::vector<Delta>, std::greater<Delta> > > delta_weights;   // lowest frame on top (:348-350)
  std::vector<int32_t> num_frames_stats, most_recent_frame_with_weight;
  std::vector<char> delta_weights_provided, updated_with_no_delta_weights;
  std::vector<std::vector<float> > current_frame_weight_debug;
  ~KhIvectorStreams() {
    PoolFree(d_off); PoolFree(d_pi); PoolFree(d_F); PoolFree(d_pu); PoolFree(d_quad); PoolFree(d_lin); PoolFree(d_xv); PoolFree(d_scal);
    PoolFree(d_cnt);
  }
};

KhIvectorStreams *kh_ivector_streams_create(const KhIvectorExtractor *x, const float *feats, int feat_stride,
                                            const int32_t *utt_row_offsets_host, int n_utts, const double *state_in_host,
                                            float *ivectors, int ivector_stride) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (!x || !feats || !utt_row_offsets_host || n_utts <= 0 || !ivectors || feat_stride < x->cfg.base_dim ||
      ivector_stride < x->cfg.ivector_dim || utt_row_offsets_host[0] != 0) {
    SetError("kh_ivector_streams_create: bad arguments");
    return nullptr;
  }
  const KhIvectorConfig &c = x->cfg;
  if (c.greedy_most_recent) {
    SetError("kh_ivector_streams_create: --use-most-recent-ivector / --greedy-ivector-extractor are not supported with frame weights");
    return nullptr;
  }
  if (c.num_gselect > 16) { SetError("kh_ivector_streams_create: num_gselect %d > 16", c.num_gselect); return nullptr; }
  for (int u = 0; u < n_utts; u++)
    if (utt_row_offsets_host[u + 1] <= utt_row_offsets_host[u]) { SetError("kh_ivector_streams_create: empty utterance %d", u); return nullptr; }
  const int rows = utt_row_offsets_host[n_utts];
  const int B = c.base_dim, D = c.feat_dim, S = c.ivector_dim, I = c.num_gauss, G = c.num_gselect;
  const int state_dim = kh_ivector_state_dim(x), lin_off = 2 * (B + 1) + 2, gamma_off = lin_off + S;
  const int sstride = (x->sdim + 3) & ~3, dstride = (D + 3) & ~3, istride = (I + 3) & ~3, bstride = (B + 3) & ~3;
  hipStream_t st = Stream();
  std::unique_ptr<KhIvectorStreams> h(new KhIvectorStreams());
  h->x = x; h->n = n_utts; h->dstride = dstride; h->ivectors = ivectors; h->ivector_stride = ivector_stride;
  h->off.assign(utt_row_offsets_host, utt_row_offsets_host + n_utts + 1);
  std::vector<int32_t> row_utt(rows);
  for (int u = 0; u < n_utts; u++)
    for (int t = utt_row_offsets_host[u]; t < utt_row_offsets_host[u + 1]; t++) row_utt[t] = u;
  h->d_off = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * (n_utts + 1)));
  h->d_pi = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * static_cast<size_t>(rows) * G));
  h->d_pu = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * G));
  h->d_F = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * dstride));
  h->d_quad = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * x->qdim));
  h->d_lin = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * S));
  h->d_xv = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * S));
  h->d_scal = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * 2));
  h->d_cnt = static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * I));
  int32_t *d_row_utt = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * rows));
  float *d_norm = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * bstride));
  float *d_spl = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * sstride));
  float *d_Fn = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * dstride));
  float *d_ll = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(rows) * istride));
  double *d_sin = state_in_host ? static_cast<double *>(PoolMalloc(sizeof(double) * static_cast<size_t>(n_utts) * state_dim)) : nullptr;
  auto cleanup = [&]() { PoolFree(d_row_utt); PoolFree(d_norm); PoolFree(d_spl); PoolFree(d_Fn); PoolFree(d_ll); PoolFree(d_sin); };
  if (!h->d_off || !h->d_pi || !h->d_pu || !h->d_F || !h->d_quad || !h->d_lin || !h->d_xv || !h->d_scal || !h->d_cnt || !d_row_utt ||
      !d_norm || !d_spl || !d_Fn || !d_ll || (state_in_host && !d_sin)) {
    cleanup();
    SetError("kh_ivector_streams_create: out of device memory");
    return nullptr;
  }
  // the streams' statistics: OnlineIvectorEstimationStats as constructed (ivector-extractor.cc:685-694) or the
  // adaptation state's (SetAdaptationState :151-160; the state carries the quadratic term as diag + counts)
  std::vector<double> lin(static_cast<size_t>(n_utts) * S, 0.0), xv(static_cast<size_t>(n_utts) * S, 0.0), scal(static_cast<size_t>(n_utts) * 2),
      cnt(static_cast<size_t>(n_utts) * I, 0.0), diag(n_utts, 1.0);
  for (int u = 0; u < n_utts; u++) {
    const double *sin = state_in_host ? state_in_host + static_cast<size_t>(u) * state_dim : nullptr;
    for (int s = 0; s < S; s++) lin[static_cast<size_t>(u) * S + s] = sin ? sin[lin_off + s] : (s == 0 ? c.prior_offset : 0.0);
    xv[static_cast<size_t>(u) * S] = c.prior_offset;          // current_ivector_ :358-359
    scal[2 * u] = sin ? sin[lin_off - 2] : 0.0;
    scal[2 * u + 1] = diag[u] = sin ? sin[lin_off - 1] : 1.0;
    if (sin)
      for (int g = 0; g < I; g++) cnt[static_cast<size_t>(u) * I + g] = sin[gamma_off + g];
  }
  int rc = KH_OK;
  double *d_diag = static_cast<double *>(PoolMalloc(sizeof(double) * n_utts));
  do {
    if (!d_diag) { rc = KH_ENOMEM; break; }
#define KH_UP(dst, src, bytes) if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { rc = KH_EDEVICE; break; }
    KH_UP(h->d_off, utt_row_offsets_host, sizeof(int32_t) * (n_utts + 1));
    KH_UP(d_row_utt, row_utt.data(), sizeof(int32_t) * rows);
    KH_UP(h->d_lin, lin.data(), sizeof(double) * lin.size());
    KH_UP(h->d_xv, xv.data(), sizeof(double) * xv.size());
    KH_UP(h->d_scal, scal.data(), sizeof(double) * scal.size());
    KH_UP(h->d_cnt, cnt.data(), sizeof(double) * cnt.size());
    KH_UP(d_diag, diag.data(), sizeof(double) * n_utts);
    if (d_sin) KH_UP(d_sin, state_in_host, sizeof(double) * static_cast<size_t>(n_utts) * state_dim);
#undef KH_UP
    hipLaunchKernelGGL(IvGemmF64Kernel, dim3(DivUp(x->qdim, kGemmN), DivUp(n_utts, kGemmM)), dim3(256), 0, st, h->d_cnt, x->U, h->d_quad,
                       n_utts, x->qdim, I);
    hipLaunchKernelGGL(IvStreamDiagKernel, dim3(std::max(1, std::min(1024, DivUp(n_utts * S, 256)))), dim3(256), 0, st, n_utts, S, x->qdim,
                       d_diag, h->d_quad);
    // the per-frame inputs, as kh_ivector_extract_adapt computes them (posteriors left unscaled)
    const KhMatrixDim dspl{rows, x->sdim, sstride}, dlda{D, x->sdim, x->sdim}, dF{rows, D, dstride};
    const int sgrid = std::min(rows, NumCUs() * 16);
    hipLaunchKernelGGL(IvSpliceKernel, dim3(sgrid), dim3(256), 0, st, feats, feat_stride, d_row_utt, h->d_off, rows, B, c.splice_left,
                       c.splice_right, d_spl, sstride);
    if ((rc = kh_affine(d_spl, dspl, x->lda, dlda, x->lda_off, h->d_F, dF))) break;
    hipLaunchKernelGGL(IvCmvnKernel, dim3(n_utts), dim3(64), 0, st, feats, feat_stride, h->d_off, B, x->gstats, c.cmn_window,
                       c.speaker_frames, c.global_frames, c.normalize_mean, c.normalize_variance, d_norm, bstride, d_sin,
                       static_cast<double *>(nullptr), state_dim);
    hipLaunchKernelGGL(IvSpliceKernel, dim3(sgrid), dim3(256), 0, st, d_norm, bstride, d_row_utt, h->d_off, rows, B, c.splice_left,
                       c.splice_right, d_spl, sstride);
    if ((rc = kh_affine(d_spl, dspl, x->lda, dlda, x->lda_off, d_Fn, dF))) break;
    if ((rc = kh_diag_gmm_loglikes(d_Fn, dF, x->ubm_g, x->ubm_mi, x->ubm_iv, I, d_ll, istride))) break;
    hipLaunchKernelGGL(IvPosteriorKernel, dim3(DivUp(rows, 4)), dim3(256), 0, st, d_ll, istride, rows, I, G, c.min_post, 1.0f, h->d_pi, h->d_pu);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { SetError("kh_ivector_streams_create: kernels failed"); rc = KH_EDEVICE; }
  } while (0);
  (void)hipStreamSynchronize(st);
  PoolFree(d_diag);
  cleanup();
  if (rc) return nullptr;
  h->delta_weights.resize(n_utts);
  h->num_frames_stats.assign(n_utts, 0);
  h->most_recent_frame_with_weight.assign(n_utts, -1);
  h->delta_weights_provided.assign(n_utts, 0);
  h->updated_with_no_delta_weights.assign(n_utts, 0);
  h->current_frame_weight_debug.resize(n_utts);
  return h.release();
}

void kh_ivector_streams_destroy(KhIvectorStreams *h) { delete h; }

int kh_ivector_streams_update_frame_weights(KhIvectorStreams *h, int stream, int n, const int32_t *frames, const float *delta_weights,
                                            int num_frames_ready) {
  KH_CHECK_ARG(h && stream >= 0 && stream < h->n && n >= 0 && (n == 0 || (frames && delta_weights)));
  const int T = h->off[stream + 1] - h->off[stream];
  KH_CHECK_ARG(num_frames_ready >= 0 && num_frames_ready <= T);
  for (int i = 0; i < n; i++) {
    if (frames[i] < 0 || frames[i] >= num_frames_ready) {   // KALDI_ASSERT(frame >= 0 && frame < num_frames_ready) :166
      SetError("UpdateFrameWeights: frame %d outside [0, NumFramesReady() = %d)", frames[i], num_frames_ready);
      return KH_EINVAL;
    }
    h->delta_weights[stream].push(KhIvectorStreams::Delta(frames[i], delta_weights[i]));
    if (frames[i] > h->most_recent_frame_with_weight[stream]) h->most_recent_frame_with_weight[stream] = frames[i];
  }
  h->delta_weights_provided[stream] = 1;
  return KH_OK;
}

int kh_ivector_streams_get_frames(KhIvectorStreams *h, int n, const int32_t *streams, const int32_t *until_frame) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h && n > 0 && streams && until_frame);
  const KhIvectorConfig &c = h->x->cfg;
  std::vector<int32_t> prog_off(1, 0), item_frame, ids;
  std::vector<float> item_w;
  for (int i = 0; i < n; i++) {
    const int u = streams[i], frame = until_frame[i];
    KH_CHECK_ARG(u >= 0 && u < h->n);
    for (int j = 0; j < i; j++) KH_CHECK_ARG(streams[j] != u);
    const int T = h->off[u + 1] - h->off[u];
    KH_CHECK_ARG(frame >= 0 && frame < T);
    if (h->delta_weights_provided[u]) {
      // UpdateStatsUntilFrameWeighted :215-254
      if (h->updated_with_no_delta_weights[u] || frame > h->most_recent_frame_with_weight[u]) {
        SetError("UpdateStatsUntilFrameWeighted: stream %d, frame %d: the frame weights must reach every frame that is asked for "
                 "(most recent frame with a weight: %d) and must have been supplied from the start", u, frame,
                 h->most_recent_frame_with_weight[u]);
        return KH_ESTATE;
      }
    } else {
      h->updated_with_no_delta_weights[u] = 1;   // UpdateStatsUntilFrame :191-213
    }
    std::vector<float> &dbg = h->current_frame_weight_debug[u];
    for (; h->num_frames_stats[u] <= frame; h->num_frames_stats[u]++) {
      const int t = h->num_frames_stats[u];
      if (h->delta_weights_provided[u]) {
        auto &q = h->delta_weights[u];
        while (!q.empty() && q.top().first <= t) {
          const KhIvectorStreams::Delta p = q.top();
          q.pop();
          item_frame.push_back(p.first);
          item_w.push_back(p.second);
          if (static_cast<int>(dbg.size()) <= p.first) dbg.resize(p.first + 1, 0.0f);
          dbg[p.first] += p.second;
          if (!(dbg[p.first] >= -0.01f && dbg[p.first] <= 1.01f)) {   // KALDI_ASSERT :237-238
            SetError("UpdateStatsUntilFrameWeighted: stream %d: the weight of frame %d became %g", u, p.first, dbg[p.first]);
            return KH_ESTATE;
          }
        }
      } else {
        item_frame.push_back(t);
        item_w.push_back(1.0f);
      }
      if (t % c.ivector_period == 0) {
        item_frame.push_back(-(t / c.ivector_period) - 1);
        item_w.push_back(0.0f);
      }
    }
    if (static_cast<int>(item_frame.size()) > prog_off.back()) {
      ids.push_back(u);
      prog_off.push_back(static_cast<int32_t>(item_frame.size()));
    }
  }
  if (ids.empty()) return KH_OK;
  hipStream_t st = Stream();
  const int na = static_cast<int>(ids.size());
  int32_t *d_ids = Upload(ids.data(), ids.size()), *d_poff = Upload(prog_off.data(), prog_off.size()),
          *d_if = Upload(item_frame.data(), item_frame.size());
  float *d_iw = Upload(item_w.data(), item_w.size());
  if (!d_ids || !d_poff || !d_if || !d_iw) { PoolFree(d_ids); PoolFree(d_poff); PoolFree(d_if); PoolFree(d_iw); return KH_ENOMEM; }
  const int D = c.feat_dim, S = c.ivector_dim, G = c.num_gselect;
  const size_t lds = sizeof(double) * (static_cast<size_t>(h->x->qdim) + 5 * S + D + static_cast<size_t>(G) * S);
  hipLaunchKernelGGL(IvStreamKernel, dim3(na), dim3(kIvThreads), lds, st, d_ids, d_poff, d_if, d_iw, h->d_off, h->d_F, h->dstride, h->d_pi,
                     h->d_pu, G, D, S, h->x->qdim, c.num_gauss, h->x->U, h->x->SiM, static_cast<double>(c.prior_offset),
                     static_cast<double>(c.max_count), c.posterior_scale, c.ivector_period, c.num_cg_iters, h->d_quad, h->d_lin, h->d_xv,
                     h->d_scal, h->d_cnt, h->ivectors, h->ivector_stride, Exact(h->x));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  PoolFree(d_ids); PoolFree(d_poff); PoolFree(d_if); PoolFree(d_iw);
  if (e != hipSuccess) { SetError("kh_ivector_streams_get_frames: %s", hipGetErrorString(e)); return KH_EDEVICE; }
  return KH_OK;
}

int kh_ivector_streams_get_stats(const KhIvectorStreams *h, double *states_out) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h && states_out);
  const KhIvectorConfig &c = h->x->cfg;
  const int S = c.ivector_dim, I = c.num_gauss, state_dim = kh_ivector_state_dim(h->x), lin_off = 2 * (c.base_dim + 1) + 2;
  std::vector<double> lin(static_cast<size_t>(h->n) * S), scal(static_cast<size_t>(h->n) * 2), cnt(static_cast<size_t>(h->n) * I);
  if (hipMemcpy(lin.data(), h->d_lin, sizeof(double) * lin.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(scal.data(), h->d_scal, sizeof(double) * scal.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(cnt.data(), h->d_cnt, sizeof(double) * cnt.size(), hipMemcpyDeviceToHost) != hipSuccess)
    return KH_EDEVICE;
  for (int u = 0; u < h->n; u++) {
    double *so = states_out + static_cast<size_t>(u) * state_dim;
    so[lin_off - 2] = scal[2 * u];
    so[lin_off - 1] = scal[2 * u + 1];
    for (int s = 0; s < S; s++) so[lin_off + s] = lin[static_cast<size_t>(u) * S + s];
    for (int g = 0; g < I; g++) so[lin_off + S + g] = cnt[static_cast<size_t>(u) * I + g];
  }
  return KH_OK;
}

}  // extern "C"
