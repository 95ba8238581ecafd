// kh_nnet.hip — nnet2 forward pass over a batch of utterances.
//
// Replaces, for the forward path only: nnet2::Nnet (nnet2/nnet-nnet.{h,cc}),
// NnetComputer / NnetComputation (nnet2/nnet-compute.cc:63-108,159-166),
// Component::Propagate of Splice/FixedAffine/Affine*/Pnorm/Normalize/Softmax/
// SumGroup/FixedScale/FixedBias (nnet2/nnet-component.cc) and the
// DecodableAmNnet epilogue (nnet2/decodable-am-nnet.h:60-69).
//
// MI355X design: all utterances of a batch are stacked by rows so every affine
// layer is ONE large MFMA GEMM (M = total frames, the reference runs one GEMM
// per utterance per layer); context padding by edge-frame duplication
// (nnet-compute.cc:75-89) and frame splicing are index arithmetic inside the
// gather kernels — nothing is uploaded per call except the utterance offsets
// (the reference uploads a row-index vector per CopyRows call,
// cu-matrix.cc:1976); activations ping-pong between two pooled buffers.
#include <algorithm>
#include <set>
#include <vector>

#include "kh_common.h"

using namespace kh;

struct KhNnet {
  struct Comp {
    int type = 0, in = 0, out = 0;
    float *W = nullptr;  // device [out x w_stride]
    int w_stride = 0;
    float *b = nullptr;  // device [out] (bias / scales)
    std::vector<int> context;
    int32_t *context_dev = nullptr;
    int const_dim = 0;
    float p = 2.f;
    int32_t *ranges = nullptr;  // device [2*out]
  };
  std::vector<Comp> comps;
  float *log_priors = nullptr;
  int n_priors = 0;
  // kh_nnet_compute_async: the descriptors and activation buffers of a call that returned before its work had finished -
  // released at the start of the next call on this handle (behind a wait for the stream) or with the handle
  std::vector<void *> held;
  hipStream_t held_stream = nullptr;   // the stream that call's work was queued on (kh_set_stream may have changed since)
};

namespace {

inline int Pad4(int n) { return (n + 3) & ~3; }

std::vector<int> ContextOf(const KhNnet::Comp &c) {
  if (c.type == KH_SPLICE) return c.context;
  return std::vector<int>(1, 0);
}

struct UttDesc {
  int in_off, in_rows, out_off, out_rows;
};

// Pads every utterance by edge-frame duplication: padded row p of utterance u is
// feats row clamp(p - left, 0, T-1)   (nnet-compute.cc:75-89).
__global__ void __launch_bounds__(256)
PadKernel(float *__restrict__ dst, int dst_stride, const float *__restrict__ src,
          int src_stride, int cols, const UttDesc *__restrict__ utts, int left) {
  const UttDesc u = utts[blockIdx.y];
  for (int p = blockIdx.x; p < u.out_rows; p += gridDim.x) {
    int s = p - left;
    s = s < 0 ? 0 : (s >= u.in_rows ? u.in_rows - 1 : s);
    const float *sr = src + static_cast<size_t>(u.in_off + s) * src_stride;
    float *dr = dst + static_cast<size_t>(u.out_off + p) * dst_stride;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dr[c] = sr[c];
  }
}

// SpliceComponent::Propagate (nnet-component.cc:2628-2708) for contiguous chunk
// infos: out[r, k*sd + c] = in[r + delta_k, c]; the trailing const_dim columns
// come from in row r (const_indexes, :2682-2684).
__global__ void __launch_bounds__(256)
SpliceBatchKernel(float *__restrict__ out, int out_stride,
                  const float *__restrict__ in, int in_stride, int in_dim,
                  int const_dim, const int32_t *__restrict__ context, int n_ctx,
                  int delta0, const UttDesc *__restrict__ utts) {
  const UttDesc u = utts[blockIdx.y];
  const int sd = in_dim - const_dim;
  const int out_dim = sd * n_ctx + const_dim;
  for (int r = blockIdx.x; r < u.out_rows; r += gridDim.x) {
    float *orow = out + static_cast<size_t>(u.out_off + r) * out_stride;
    const float *ibase = in + static_cast<size_t>(u.in_off) * in_stride;
    for (int c = threadIdx.x; c < out_dim; c += blockDim.x) {
      float v;
      if (c < sd * n_ctx) {
        const int k = c / sd, cc = c - k * sd;
        const int ir = r + delta0 + context[k];
        v = ibase[static_cast<size_t>(ir) * in_stride + cc];
      } else {
        v = ibase[static_cast<size_t>(r) * in_stride + sd + (c - sd * n_ctx)];
      }
      orow[c] = v;
    }
  }
}

// Non-contiguous chunk infos (sparse splicing in deeper layers): explicit
// per-row source indexes, as the reference builds them (:2649-2680).
__global__ void __launch_bounds__(256)
SpliceIndexedKernel(float *__restrict__ out, int out_stride, int out_rows,
                    const float *__restrict__ in, int in_stride, int in_dim,
                    int const_dim, int n_ctx,
                    const int32_t *__restrict__ indexes /* [n_ctx+1][out_rows] */) {
  const int sd = in_dim - const_dim;
  const int out_dim = sd * n_ctx + const_dim;
  for (int r = blockIdx.x; r < out_rows; r += gridDim.x) {
    float *orow = out + static_cast<size_t>(r) * out_stride;
    for (int c = threadIdx.x; c < out_dim; c += blockDim.x) {
      float v;
      if (c < sd * n_ctx) {
        const int k = c / sd, cc = c - k * sd;
        const int ir = indexes[static_cast<size_t>(k) * out_rows + r];
        v = ir < 0 ? 0.f : in[static_cast<size_t>(ir) * in_stride + cc];
      } else {
        const int ir = indexes[static_cast<size_t>(n_ctx) * out_rows + r];
        v = in[static_cast<size_t>(ir) * in_stride + sd + (c - sd * n_ctx)];
      }
      orow[c] = v;
    }
  }
}

struct Chunk {  // ChunkInfo nnet-component.h:72-146, one chunk
  int first = 0, last = 0;
  std::vector<int> offsets;
  int Size() const { return offsets.empty() ? last - first + 1 : (int)offsets.size(); }
  int GetOffset(int i) const { return offsets.empty() ? first + i : offsets[i]; }
  int GetIndex(int off) const {
    if (offsets.empty()) return off - first;
    return (int)(std::lower_bound(offsets.begin(), offsets.end(), off) - offsets.begin());
  }
};

// Nnet::ComputeChunkInfo nnet-nnet.cc:65-112 for one chunk of num_rows.
bool ComputeChunkInfo(const KhNnet &n, int num_rows, int L, int R,
                      std::vector<Chunk> *info) {
  const int nc = (int)n.comps.size();
  const int out_rows = num_rows - L - R;
  if (out_rows <= 0) return false;
  info->assign(nc + 1, Chunk());
  std::vector<int> cur(out_rows);
  for (int i = 0; i < out_rows; i++) cur[i] = i + L;
  (*info)[nc].first = cur.front();
  (*info)[nc].last = cur.back();
  for (int i = nc - 1; i >= 0; i--) {
    std::vector<int> ctx = ContextOf(n.comps[i]);
    if (ctx.size() > 1 || ctx[0] != 0) {
      std::set<int> s;
      for (int c : ctx)
        for (int k : cur) s.insert(c + k);
      cur.assign(s.begin(), s.end());
    }
    (*info)[i].first = cur.front();
    (*info)[i].last = cur.back();
    if ((int)cur.size() != cur.back() - cur.front() + 1) (*info)[i].offsets = cur;
  }
  return true;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) PoolFree(p); }
  bool Alloc(size_t bytes) {
    if (p) PoolFree(p);
    p = PoolMalloc(bytes ? bytes : 4);
    return p != nullptr;
  }
};

}  // namespace

extern "C" {

KhNnet *kh_nnet_create(void) { return new KhNnet(); }

void kh_nnet_destroy(KhNnet *n) {
  if (!n) return;
  for (auto &c : n->comps) {
    PoolFree(c.W);
    PoolFree(c.b);
    PoolFree(c.ranges);
    PoolFree(c.context_dev);
  }
  PoolFree(n->log_priors);
  if (!n->held.empty()) {
    (void)hipStreamSynchronize(n->held_stream);
    for (void *p : n->held) PoolFree(p);
  }
  delete n;
}

int kh_nnet_add_component(KhNnet *n, const KhComponentDesc *d) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n && d && d->input_dim > 0 && d->output_dim > 0);
  if (!n->comps.empty()) KH_CHECK_ARG(n->comps.back().out == d->input_dim);
  KhNnet::Comp c;
  c.type = d->type;
  c.in = d->input_dim;
  c.out = d->output_dim;
  hipStream_t st = Stream();
  switch (d->type) {
    case KH_SPLICE: {
      KH_CHECK_ARG(d->context && d->n_context > 0 && d->const_dim >= 0 &&
                   d->const_dim < d->input_dim);
      c.context.assign(d->context, d->context + d->n_context);
      for (size_t i = 1; i < c.context.size(); i++)
        KH_CHECK_ARG(c.context[i] > c.context[i - 1]);  // strictly increasing
      KH_CHECK_ARG(c.context.front() <= 0 && c.context.back() >= 0);
      c.const_dim = d->const_dim;
      KH_CHECK_ARG(c.out == (c.in - c.const_dim) * d->n_context + c.const_dim);
      c.context_dev = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * d->n_context));
      if (!c.context_dev) return KH_ENOMEM;
      KH_HIP(hipMemcpyAsync(c.context_dev, d->context, sizeof(int32_t) * d->n_context,
                            hipMemcpyHostToDevice, st));
      break;
    }
    case KH_FIXED_AFFINE:
    case KH_AFFINE: {
      KH_CHECK_ARG(d->linear && d->bias);
      c.w_stride = Pad4(c.in);
      c.W = static_cast<float *>(PoolMalloc(sizeof(float) * c.out * (size_t)c.w_stride));
      c.b = static_cast<float *>(PoolMalloc(sizeof(float) * c.out));
      if (!c.W || !c.b) return KH_ENOMEM;
      KH_HIP(hipMemcpy2DAsync(c.W, sizeof(float) * c.w_stride, d->linear,
                              sizeof(float) * c.in, sizeof(float) * c.in, c.out,
                              hipMemcpyHostToDevice, st));
      KH_HIP(hipMemcpyAsync(c.b, d->bias, sizeof(float) * c.out,
                            hipMemcpyHostToDevice, st));
      break;
    }
    case KH_PNORM:
      KH_CHECK_ARG(c.in % c.out == 0 && d->p >= 0.f);
      c.p = d->p;
      break;
    case KH_NORMALIZE:
    case KH_SOFTMAX:
      KH_CHECK_ARG(c.in == c.out);
      break;
    case KH_SUM_GROUP: {
      KH_CHECK_ARG(d->sizes && d->n_sizes == c.out);
      std::vector<int32_t> ranges(2 * c.out);
      int cur = 0;
      for (int i = 0; i < c.out; i++) {  // SumGroupComponent::Init :2440-2456
        KH_CHECK_ARG(d->sizes[i] > 0);
        ranges[2 * i] = cur;
        cur += d->sizes[i];
        ranges[2 * i + 1] = cur;
      }
      KH_CHECK_ARG(cur == c.in);
      c.ranges = static_cast<int32_t *>(PoolMalloc(sizeof(int32_t) * 2 * c.out));
      if (!c.ranges) return KH_ENOMEM;
      KH_HIP(hipMemcpyAsync(c.ranges, ranges.data(), sizeof(int32_t) * 2 * c.out,
                            hipMemcpyHostToDevice, st));
      KH_HIP(hipStreamSynchronize(st));  // ranges is a local
      break;
    }
    case KH_FIXED_SCALE:
    case KH_FIXED_BIAS:
      KH_CHECK_ARG(c.in == c.out && d->bias);
      c.b = static_cast<float *>(PoolMalloc(sizeof(float) * c.out));
      if (!c.b) return KH_ENOMEM;
      KH_HIP(hipMemcpyAsync(c.b, d->bias, sizeof(float) * c.out,
                            hipMemcpyHostToDevice, st));
      break;
    default:
      SetError("kh_nnet_add_component: unknown component type %d", d->type);
      return KH_EINVAL;
  }
  KH_HIP(hipStreamSynchronize(st));
  n->comps.push_back(c);
  return KH_OK;
}

int kh_nnet_set_priors(KhNnet *n, const float *priors, int np) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n && priors && np > 0);
  std::vector<float> lp(np);
  for (int i = 0; i < np; i++) lp[i] = logf(priors[i]);  // priors.ApplyLog() :63
  PoolFree(n->log_priors);
  n->log_priors = static_cast<float *>(PoolMalloc(sizeof(float) * np));
  if (!n->log_priors) return KH_ENOMEM;
  n->n_priors = np;
  KH_HIP(hipMemcpyAsync(n->log_priors, lp.data(), sizeof(float) * np,
                        hipMemcpyHostToDevice, Stream()));
  KH_HIP(hipStreamSynchronize(Stream()));
  return KH_OK;
}

int kh_nnet_num_components(const KhNnet *n) { return n ? (int)n->comps.size() : 0; }
int kh_nnet_input_dim(const KhNnet *n) { return n && !n->comps.empty() ? n->comps.front().in : 0; }
int kh_nnet_output_dim(const KhNnet *n) { return n && !n->comps.empty() ? n->comps.back().out : 0; }
int kh_nnet_left_context(const KhNnet *n) {
  int a = 0;
  for (auto &c : n->comps) a += ContextOf(c).front();
  return -a;
}
int kh_nnet_right_context(const KhNnet *n) {
  int a = 0;
  for (auto &c : n->comps) a += ContextOf(c).back();
  return a;
}

}  // extern "C"

namespace {
// wait: kh_nnet_compute (the buffers of the call go back to the pool when it returns); !wait: kh_nnet_compute_async
int NnetComputeImpl(KhNnet *n, const float *feats, int feat_stride,
                    const int32_t *utt_off, int n_utts, int pad_input,
                    int epilogue, float prob_scale, float *out, int out_stride,
                    int32_t *out_row_offsets_host, bool wait) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(n && feats && utt_off && out && n_utts > 0 && !n->comps.empty());
  if (!n->held.empty()) {   // what an asynchronous call left in use
    KH_HIP(hipStreamSynchronize(n->held_stream));
    for (void *p : n->held) PoolFree(p);
    n->held.clear();
  }
  const int nc = (int)n->comps.size();
  const int in_dim = n->comps.front().in, out_dim = n->comps.back().out;
  KH_CHECK_ARG(feat_stride >= in_dim && out_stride >= out_dim);
  if (epilogue) KH_CHECK_ARG(n->log_priors && n->n_priors == out_dim);
  const int L = kh_nnet_left_context(n), R = kh_nnet_right_context(n);
  const int left = pad_input ? L : 0, right = pad_input ? R : 0;
  hipStream_t st = Stream();

  // Per-component, per-utterance row extents.  For contiguous chunk infos the
  // number of rows entering component i is T + left + right - (span consumed by
  // components before i); firsts depend on contexts only, not on T.
  std::vector<int> first(nc + 1);  // chunk_info[i].first
  {
    int f = L;
    first[nc] = f;
    for (int i = nc - 1; i >= 0; i--) {
      f += ContextOf(n->comps[i]).front();
      first[i] = f;
    }
  }
  bool contiguous = true;
  for (int i = 0; i < nc && contiguous; i++) {
    std::vector<int> ctx = ContextOf(n->comps[i]);
    for (size_t k = 1; k < ctx.size(); k++)
      if (ctx[k] != ctx[k - 1] + 1) {
        // A sparse context is still contiguous at its INPUT if the rows it
        // feeds are dense enough; decide exactly with ComputeChunkInfo below.
        contiguous = false;
      }
  }
  std::vector<std::vector<Chunk>> infos;  // only for the non-contiguous path
  if (!contiguous) {
    // exact check on the first utterance length class; fall back per utterance
    infos.resize(n_utts);
    contiguous = true;
    for (int u = 0; u < n_utts; u++) {
      const int T = utt_off[u + 1] - utt_off[u];
      KH_CHECK_ARG(T > 0);
      if (!ComputeChunkInfo(*n, left + T + right, L, R, &infos[u])) {
        SetError("utterance %d too short (%d frames) for context %d+%d", u, T, L, R);
        return KH_EINVAL;
      }
      for (auto &ci : infos[u])
        if (!ci.offsets.empty()) contiguous = false;
    }
    if (contiguous) infos.clear();
  }

  // rows[i][u] = rows of utterance u entering component i (i == nc: output)
  std::vector<std::vector<int>> rows(nc + 1, std::vector<int>(n_utts));
  std::vector<std::vector<int>> offs(nc + 1, std::vector<int>(n_utts + 1, 0));
  for (int u = 0; u < n_utts; u++) {
    const int T = utt_off[u + 1] - utt_off[u];
    KH_CHECK_ARG(T > 0);
    const int num_rows = left + T + right;
    if (num_rows - L - R <= 0) {
      SetError("utterance %d too short (%d frames) for context %d+%d", u, T, L, R);
      return KH_EINVAL;
    }
    if (infos.empty()) {
      int r = num_rows;
      for (int i = 0; i <= nc; i++) {
        rows[i][u] = r;
        if (i < nc) {
          std::vector<int> ctx = ContextOf(n->comps[i]);
          r -= ctx.back() - ctx.front();
        }
      }
    } else {
      for (int i = 0; i <= nc; i++) rows[i][u] = infos[u][i].Size();
    }
  }
  for (int i = 0; i <= nc; i++)
    for (int u = 0; u < n_utts; u++) offs[i][u + 1] = offs[i][u] + rows[i][u];
  if (out_row_offsets_host)
    for (int u = 0; u <= n_utts; u++) out_row_offsets_host[u] = offs[nc][u];

  // utterance descriptors for pad + each splice component
  std::vector<UttDesc> descs;
  auto push_descs = [&](const std::vector<int> &ioff, const std::vector<int> &irows,
                        const std::vector<int> &ooff, const std::vector<int> &orows) {
    size_t base = descs.size();
    for (int u = 0; u < n_utts; u++)
      descs.push_back(UttDesc{ioff[u], irows[u], ooff[u], orows[u]});
    return base;
  };
  std::vector<int> feat_off(utt_off, utt_off + n_utts + 1), feat_rows(n_utts);
  for (int u = 0; u < n_utts; u++) feat_rows[u] = utt_off[u + 1] - utt_off[u];
  const size_t pad_desc = push_descs(feat_off, feat_rows, offs[0], rows[0]);
  std::vector<size_t> splice_desc(nc, 0);
  for (int i = 0; i < nc; i++)
    if (n->comps[i].type == KH_SPLICE)
      splice_desc[i] = push_descs(offs[i], rows[i], offs[i + 1], rows[i + 1]);
  DevBuf d_descs;
  if (!d_descs.Alloc(sizeof(UttDesc) * descs.size())) return KH_ENOMEM;
  KH_HIP(hipMemcpyAsync(d_descs.p, descs.data(), sizeof(UttDesc) * descs.size(),
                        hipMemcpyHostToDevice, st));
  const UttDesc *dd = static_cast<const UttDesc *>(d_descs.p);

  // activation buffers
  size_t max_elems = 0;
  for (int i = 0; i <= nc; i++) {
    const int dim = i == 0 ? in_dim : n->comps[i - 1].out;
    max_elems = std::max(max_elems, (size_t)offs[i][n_utts] * Pad4(dim));
  }
  DevBuf buf[2];
  if (!buf[0].Alloc(sizeof(float) * max_elems) || !buf[1].Alloc(sizeof(float) * max_elems))
    return KH_ENOMEM;

  int max_rows_u = 0;
  for (int u = 0; u < n_utts; u++) max_rows_u = std::max(max_rows_u, rows[0][u]);
  auto row_grid = [&](int max_rows) {
    int gx = std::min(max_rows, std::max(1, NumCUs() * 16 / n_utts));
    return dim3(gx, n_utts);
  };

  // stage 0: padded input
  const float *cur = feats;
  int cur_stride = feat_stride;
  int which = 0;
  if (left > 0 || right > 0) {
    float *dst = static_cast<float *>(buf[which].p);
    hipLaunchKernelGGL(PadKernel, row_grid(max_rows_u), dim3(in_dim >= 256 ? 256 : 64),
                       0, st, dst, Pad4(in_dim), feats, feat_stride, in_dim,
                       dd + pad_desc, left);
    KH_LAUNCH_CHECK();
    cur = dst;
    cur_stride = Pad4(in_dim);
    which ^= 1;
  } else {
    // rows[0] == feats rows and offsets coincide only if utt_off[0] == 0 and the
    // utterances are packed; otherwise compact them with the same kernel.
    bool packed = utt_off[0] == 0;
    if (!packed) {
      float *dst = static_cast<float *>(buf[which].p);
      hipLaunchKernelGGL(PadKernel, row_grid(max_rows_u), dim3(in_dim >= 256 ? 256 : 64),
                         0, st, dst, Pad4(in_dim), feats, feat_stride, in_dim,
                         dd + pad_desc, 0);
      KH_LAUNCH_CHECK();
      cur = dst;
      cur_stride = Pad4(in_dim);
      which ^= 1;
    }
  }

  DevBuf d_index;  // non-contiguous splice indexes
  // wait: the descriptors / buffers are released to the pool on return, so the work must have finished; else the handle
  // keeps them until its next call
  auto finish = [&]() -> int {
    if (wait) {
      KH_HIP(hipStreamSynchronize(st));
      return KH_OK;
    }
    for (DevBuf *b : {&d_descs, &buf[0], &buf[1], &d_index})
      if (b->p) { n->held.push_back(b->p); b->p = nullptr; }
    n->held_stream = st;
    return KH_OK;
  };
  for (int i = 0; i < nc; i++) {
    const KhNnet::Comp &c = n->comps[i];
    const int in_rows = offs[i][n_utts], o_rows = offs[i + 1][n_utts];
    const bool last = (i == nc - 1);
    float *dst = last ? out : static_cast<float *>(buf[which].p);
    const int dst_stride = last ? out_stride : Pad4(c.out);
    KhMatrixDim din{in_rows, c.in, cur_stride}, dout{o_rows, c.out, dst_stride};
    switch (c.type) {
      case KH_SPLICE: {
        int mr = 0;
        for (int u = 0; u < n_utts; u++) mr = std::max(mr, rows[i + 1][u]);
        if (infos.empty()) {
          const int delta0 = first[i + 1] - first[i];
          hipLaunchKernelGGL(SpliceBatchKernel, row_grid(mr),
                             dim3(c.out >= 256 ? 256 : 64), 0, st, dst, dst_stride,
                             cur, cur_stride, c.in, c.const_dim, c.context_dev,
                             (int)c.context.size(), delta0, dd + splice_desc[i]);
        } else {
          const int nctx = (int)c.context.size();
          std::vector<int32_t> idx((size_t)(nctx + 1) * o_rows);
          for (int u = 0; u < n_utts; u++) {
            const Chunk &ii = infos[u][i], &oi = infos[u][i + 1];
            for (int r = 0; r < rows[i + 1][u]; r++) {
              for (int k = 0; k < nctx; k++)
                idx[(size_t)k * o_rows + offs[i + 1][u] + r] =
                    offs[i][u] + ii.GetIndex(oi.GetOffset(r) + c.context[k]);
              idx[(size_t)nctx * o_rows + offs[i + 1][u] + r] = offs[i][u] + r;
            }
          }
          if (!d_index.Alloc(sizeof(int32_t) * idx.size())) return KH_ENOMEM;
          KH_HIP(hipMemcpyAsync(d_index.p, idx.data(), sizeof(int32_t) * idx.size(),
                                hipMemcpyHostToDevice, st));
          KH_HIP(hipStreamSynchronize(st));
          hipLaunchKernelGGL(SpliceIndexedKernel,
                             dim3(std::min(o_rows, NumCUs() * 16)),
                             dim3(c.out >= 256 ? 256 : 64), 0, st, dst, dst_stride,
                             o_rows, cur, cur_stride, c.in, c.const_dim, nctx,
                             static_cast<const int32_t *>(d_index.p));
        }
        KH_LAUNCH_CHECK();
        break;
      }
      case KH_FIXED_AFFINE:
      case KH_AFFINE:
        if (i + 1 < nc && n->comps[i + 1].type == KH_PNORM && n->comps[i + 1].p == 2.0f && n->comps[i + 1].in == c.out &&
            offs[i + 2][n_utts] == o_rows && kh_affine_pnorm_supported(c.out / n->comps[i + 1].out) &&
            !getenv("KH_NNET_NO_FUSED_PNORM")) {
          // hidden layer: affine -> p-norm in one kernel, the wide activations are never written
          const KhNnet::Comp &pn = n->comps[i + 1];
          const bool pn_last = (i + 2 == nc);
          float *pdst = pn_last ? out : dst;
          const int pstride = pn_last ? out_stride : Pad4(pn.out);
          rc = kh_affine_pnorm(cur, din, c.W, KhMatrixDim{c.out, c.in, c.w_stride}, c.b, pdst,
                               KhMatrixDim{o_rows, pn.out, pstride}, c.out / pn.out);
          if (rc) return rc;
          cur = pdst;
          cur_stride = pstride;
          which ^= 1;
          i++;
          continue;
        }
        rc = kh_affine(cur, din, c.W, KhMatrixDim{c.out, c.in, c.w_stride}, c.b, dst, dout);
        break;
      case KH_PNORM:
        rc = kh_group_pnorm(dst, cur, dout, cur_stride, c.in / c.out, c.p);
        break;
      case KH_NORMALIZE:
        rc = kh_normalize(dst, cur, dout, cur_stride);
        break;
      case KH_SOFTMAX:
        if (i + 2 == nc && n->comps[i + 1].type == KH_SUM_GROUP && c.out <= SoftmaxLdsCols() &&
            !getenv("KH_NNET_NO_FUSED_OUTPUT")) {
          // output layer: softmax -> sum-group (-> epilogue) in one pass over the logits
          const KhNnet::Comp &sg = n->comps[i + 1];
          rc = FusedSoftmaxSumGroup(out, KhMatrixDim{offs[nc][n_utts], sg.out, out_stride}, cur, din, sg.ranges,
                                    epilogue ? n->log_priors : nullptr, prob_scale);
          if (rc) return rc;
          return finish();
        }
        rc = kh_softmax_per_row(dst, cur, dout, cur_stride);
        if (!rc) rc = kh_apply_floor(dst, dout, 1.0e-20f);  // :942
        break;
      case KH_SUM_GROUP:
        rc = kh_sum_column_ranges(dst, dout, cur, din, c.ranges);
        break;
      case KH_FIXED_SCALE:
      case KH_FIXED_BIAS: {
        KH_HIP(hipMemcpy2DAsync(dst, sizeof(float) * dst_stride, cur,
                                sizeof(float) * cur_stride, sizeof(float) * c.in,
                                in_rows, hipMemcpyDeviceToDevice, st));
        rc = c.type == KH_FIXED_SCALE ? kh_mul_cols_vec(dst, dout, c.b)
                                      : kh_add_vec_to_rows(1.0f, c.b, 1.0f, dst, dout);
        break;
      }
      default:
        return KH_EINVAL;
    }
    if (rc) return rc;
    cur = dst;
    cur_stride = dst_stride;
    which ^= 1;
  }
  if (epilogue) {
    rc = kh_log_prior_scale(out, KhMatrixDim{offs[nc][n_utts], out_dim, out_stride},
                            n->log_priors, prob_scale);
    if (rc) return rc;
  }
  return finish();
}
}  // namespace

extern "C" {

int kh_nnet_compute(KhNnet *n, const float *feats, int feat_stride,
                    const int32_t *utt_off, int n_utts, int pad_input,
                    int epilogue, float prob_scale, float *out, int out_stride,
                    int32_t *out_row_offsets_host) {
  return NnetComputeImpl(n, feats, feat_stride, utt_off, n_utts, pad_input, epilogue, prob_scale, out, out_stride,
                         out_row_offsets_host, true);
}

int kh_nnet_compute_async(KhNnet *n, const float *feats, int feat_stride,
                          const int32_t *utt_off, int n_utts, int pad_input,
                          int epilogue, float prob_scale, float *out, int out_stride,
                          int32_t *out_row_offsets_host) {
  return NnetComputeImpl(n, feats, feat_stride, utt_off, n_utts, pad_input, epilogue, prob_scale, out, out_stride,
                         out_row_offsets_host, false);
}

}  // extern "C"
