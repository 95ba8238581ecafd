// kh_online_nnet.hip — the serving loop of online2-wav-nnet2-latgen-faster (online2bin/online2-wav-nnet2-latgen-faster.cc
// :213-262) for many concurrent streams, ONE library call per step:
//   feature rows of a chunk arrive for every live stream   (OnlineNnet2FeaturePipeline::AcceptWaveform's output)
//   DecodableNnet2Online::ComputeForFrame for the frames that became ready (nnet2/online-nnet2-decodable.cc:91-143,
//     NumFramesReady :75-89): the context-padded input rows of ALL advancing streams gathered into one matrix, ONE
//     forward pass (NnetComputation pad_input = false), floor / log / -log prior / acoustic scale
//   LatticeFasterOnlineDecoder::AdvanceDecoding for those streams (one launch of the online decode kernel)
// The host side of a step — bookkeeping of 256 streams, the gather indices — is a few microseconds of C++ here; it was
// ~4 ms of Python per step in the round-3 serving leg (6.2 ms per 5-frame step at 256 streams, of which the kernels are ~2).
// Row-for-row the scores are those of kh_nnet_compute on the whole utterance (tests/test_gpu_online_nnet.py).
#include <algorithm>
#include <vector>

#include "kh_common.h"

using namespace kh;

namespace {

// dst[dst_rows[i], :] = src[src_rows[i], :]  (one workgroup row per block.y, column on the lane)
__global__ void __launch_bounds__(256)
ScatterRowsKernel(float *__restrict__ dst, int dst_stride, const float *__restrict__ src, int src_stride, const int32_t *__restrict__ dst_rows,
                  const int32_t *__restrict__ src_rows, int n, int cols) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const float *s = src + static_cast<size_t>(src_rows[i]) * src_stride;
    float *d = dst + static_cast<size_t>(dst_rows[i]) * dst_stride;
    for (int c = threadIdx.x; c < cols; c += 256) d[c] = s[c];
  }
}

template <class T>
struct Grow {   // device scratch that only grows
  T *p = nullptr;
  size_t cap = 0;
  int Need(size_t n) {
    if (n <= cap) return KH_OK;
    if (p) PoolFree(p);
    cap = std::max(n, cap * 2);
    p = static_cast<T *>(PoolMalloc(sizeof(T) * cap));
    if (!p) { cap = 0; return KH_ENOMEM; }
    return KH_OK;
  }
  ~Grow() { if (p) PoolFree(p); }
};

}  // namespace

struct KhOnlineNnet2 {
  KhNnet *nnet = nullptr;
  KhOnlineDecoder *dec = nullptr;
  int num_streams = 0, max_frames = 0, L = 0, R = 0, dim = 0, stride = 0, out_dim = 0, out_stride = 0, max_batch = 0, pad_input = 1;
  float scale = 1.0f;
  float *feats = nullptr;                 // [num_streams * max_frames, stride]: every stream's feature history
  std::vector<int32_t> n, finished, decoded;
  Grow<int32_t> d_idx;                    // index uploads of a step
  Grow<float> d_x, d_out;
  int32_t *h_idx = nullptr;               // pinned staging for d_idx
  size_t h_cap = 0;
  std::vector<int32_t> v_idx, off, out_off, act, nfr;
  std::vector<const float *> ptrs;
  // serving through the decoder's persistent kernel (kh_online_nnet2_serve_start): every stream's scores of the current
  // utterance, row t = frame t; `decoded` then counts the frames SUBMITTED to the decoder, which follows at its own pace
  bool serving = false;
  float *ll = nullptr;                    // [num_streams * max_frames, out_stride]
  const int32_t *serve_map = nullptr;
  std::vector<int32_t> avail;
};

extern "C" {

KhOnlineNnet2 *kh_online_nnet2_create(KhNnet *nnet, KhOnlineDecoder *dec, int num_streams, int max_frames, float acoustic_scale,
                                      int pad_input, int max_nnet_batch_size) {
  if (EnsureDevice() != KH_OK) return nullptr;
  if (!nnet || !dec || num_streams <= 0 || max_frames <= 0 || max_nnet_batch_size <= 0) {   // (:40 KALDI_ASSERT(max_nnet_batch_size > 0))
    SetError("kh_online_nnet2_create: bad arguments");
    return nullptr;
  }
  KhOnlineNnet2 *h = new KhOnlineNnet2();
  h->nnet = nnet;
  h->dec = dec;
  h->num_streams = num_streams;
  h->max_frames = max_frames;
  h->L = kh_nnet_left_context(nnet);
  h->R = kh_nnet_right_context(nnet);
  h->dim = kh_nnet_input_dim(nnet);
  h->stride = (h->dim + 3) / 4 * 4;
  h->out_dim = kh_nnet_output_dim(nnet);
  h->out_stride = (h->out_dim + 3) / 4 * 4;
  h->max_batch = max_nnet_batch_size;
  h->pad_input = pad_input != 0;
  h->scale = acoustic_scale;
  h->feats = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(num_streams) * max_frames * h->stride));
  if (!h->feats) {
    SetError("kh_online_nnet2_create: no device memory for %d streams x %d frames of features", num_streams, max_frames);
    delete h;
    return nullptr;
  }
  h->n.assign(num_streams, 0);
  h->finished.assign(num_streams, 0);
  h->decoded.assign(num_streams, 0);
  return h;
}

void kh_online_nnet2_destroy(KhOnlineNnet2 *h) {
  if (!h) return;
  (void)kh_online_nnet2_serve_stop(h);
  if (h->ll) PoolFree(h->ll);
  PoolFree(h->feats);
  if (h->h_idx) (void)hipHostFree(h->h_idx);
  delete h;
}

// A new utterance on each of the streams: the feature history is dropped and LatticeFasterOnlineDecoder::InitDecoding runs.
int kh_online_nnet2_reset(KhOnlineNnet2 *h, const int32_t *streams, int n) {
  KH_CHECK_ARG(h && streams && n > 0);
  for (int i = 0; i < n; i++) {
    KH_CHECK_ARG(streams[i] >= 0 && streams[i] < h->num_streams);
    h->n[streams[i]] = 0;
    h->finished[streams[i]] = 0;
    h->decoded[streams[i]] = 0;
  }
  if (h->serving) return kh_online_decoder_serve_init(h->dec, streams, n);
  return kh_online_decoder_init_decoding(h->dec, streams, n);
}

// One serving step.  Stream streams[i] receives counts[i] (>= 0) feature rows, rows [src_rows[i], src_rows[i] + counts[i])
// of the DEVICE matrix src; finished[i] != 0 = InputFinished() after them.  Every stream then advances by the frames that
// became ready (at most max_nnet_batch_size).  frames_decoded (may be NULL) receives NumFramesDecoded() of every listed
// stream after the step.  tid2pdf: DEVICE map (or NULL), as kh_online_decoder_advance.
int kh_online_nnet2_step(KhOnlineNnet2 *h, const int32_t *streams, int n, const float *src, int src_stride, const int32_t *src_rows,
                         const int32_t *counts, const int32_t *finished, const int32_t *tid2pdf, int32_t *frames_decoded) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h && streams && n > 0 && n <= h->num_streams && counts && finished);
  hipStream_t st = Stream();
  // ---- (1) bookkeeping of the new rows, the scatter's index pairs
  size_t n_new = 0;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    KH_CHECK_ARG(s >= 0 && s < h->num_streams && counts[i] >= 0);
    if (h->finished[s] && counts[i] > 0) {
      SetError("kh_online_nnet2_step: stream %d: features after InputFinished()", s);
      return KH_ESTATE;
    }
    if (h->n[s] + counts[i] > h->max_frames) {
      SetError("kh_online_nnet2_step: stream %d: more than max_frames = %d feature frames", s, h->max_frames);
      return KH_EINVAL;
    }
    n_new += counts[i];
  }
  KH_CHECK_ARG(n_new == 0 || (src && src_rows && src_stride >= h->dim));
  std::vector<int32_t> &v = h->v_idx;
  v.clear();
  v.reserve(2 * n_new + static_cast<size_t>(n) * (h->L + h->R + 64));
  for (int i = 0; i < n; i++)   // destination rows
    for (int k = 0; k < counts[i]; k++) v.push_back(streams[i] * h->max_frames + h->n[streams[i]] + k);
  for (int i = 0; i < n; i++)   // source rows
    for (int k = 0; k < counts[i]; k++) v.push_back(src_rows[i] + k);
  for (int i = 0; i < n; i++) {
    h->n[streams[i]] += counts[i];
    if (finished[i]) h->finished[streams[i]] = 1;
  }
  // ---- (2) what became ready (NumFramesReady :75-89) and the rows ComputeForFrame gathers (:103-124)
  h->off.assign(1, 0);
  h->act.clear();
  h->nfr.clear();
  const size_t gather_b = v.size();
  for (int i = 0; i < n; i++) {
    const int s = streams[i], have = h->n[s];
    int ready = 0;
    if (have > 0) ready = h->pad_input ? (h->finished[s] ? have : std::max(0, have - h->R)) : std::max(0, have - h->R - h->L);
    const int f = h->decoded[s];
    const int m = std::max(0, std::min(ready - f, h->max_batch));
    if (m == 0) continue;
    const int begin = h->pad_input ? f - h->L : f, rows = m + h->L + h->R;
    for (int t = begin; t < begin + rows; t++) v.push_back(s * h->max_frames + std::min(std::max(t, 0), have - 1));
    h->off.push_back(h->off.back() + rows);
    h->act.push_back(s);
    h->nfr.push_back(m);
  }
  const int n_act = static_cast<int>(h->act.size()), n_rows = h->off.back();
  // (serving) the rows of the network's output go to the streams' score buffers: destination rows, then source rows
  const size_t score_b = v.size();
  size_t n_score = 0;
  if (h->serving) {
    for (int k = 0; k < n_act; k++)
      for (int j = 0; j < h->nfr[k]; j++) v.push_back(h->act[k] * h->max_frames + h->decoded[h->act[k]] + j);
    for (int k = 0, r = 0; k < n_act; k++)
      for (int j = 0; j < h->nfr[k]; j++) v.push_back(r++);
    n_score = (v.size() - score_b) / 2;
  }
  // ---- (3) one upload of all indices, the scatter, the gather, the forward pass
  if (!v.empty()) {
    if (v.size() > h->h_cap) {
      if (h->h_idx) (void)hipHostFree(h->h_idx);
      h->h_cap = std::max(v.size() * 2, static_cast<size_t>(1 << 16));
      if (hipHostMalloc(reinterpret_cast<void **>(&h->h_idx), sizeof(int32_t) * h->h_cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        h->h_idx = nullptr;
        h->h_cap = 0;
        SetError("kh_online_nnet2_step: cannot allocate pinned host memory");
        return KH_ENOMEM;
      }
    }
    if ((rc = h->d_idx.Need(v.size()))) return rc;
    memcpy(h->h_idx, v.data(), sizeof(int32_t) * v.size());   // (the previous step's copy has completed: every step ends synchronised)
    KH_HIP(hipMemcpyAsync(h->d_idx.p, h->h_idx, sizeof(int32_t) * v.size(), hipMemcpyHostToDevice, st));
  }
  if (n_new > 0) {
    hipLaunchKernelGGL(ScatterRowsKernel, dim3(static_cast<unsigned>(std::min<size_t>(n_new, 4096))), dim3(256), 0, st, h->feats, h->stride, src,
                       src_stride, h->d_idx.p, h->d_idx.p + n_new, static_cast<int>(n_new), h->dim);
    KH_LAUNCH_CHECK();
  }
  if (n_act > 0) {
    if ((rc = h->d_x.Need(static_cast<size_t>(n_rows) * h->stride)) || (rc = h->d_out.Need(static_cast<size_t>(n_rows) * h->out_stride))) return rc;
    KhMatrixDim dx{n_rows, h->dim, h->stride};
    if ((rc = kh_copy_rows(h->d_x.p, dx, h->feats, h->stride, h->d_idx.p + gather_b))) return rc;
    h->out_off.assign(n_act + 1, 0);
    if ((rc = kh_nnet_compute(h->nnet, h->d_x.p, h->stride, h->off.data(), n_act, /*pad_input=*/0, /*epilogue=*/1, h->scale, h->d_out.p,
                              h->out_stride, h->out_off.data())))
      return rc;
    // ---- (4) AdvanceDecoding of the streams that got frames
    h->ptrs.resize(n_act);
    for (int k = 0; k < n_act; k++) {
      if (h->out_off[k + 1] - h->out_off[k] != h->nfr[k]) {
        SetError("kh_online_nnet2_step: the network returned %d rows for %d frames", h->out_off[k + 1] - h->out_off[k], h->nfr[k]);
        return KH_ESTATE;
      }
      h->ptrs[k] = h->d_out.p + static_cast<size_t>(h->out_off[k]) * h->out_stride;
    }
    if (h->serving) {
      // the scores into the streams' buffers; once they are there (the stream is idle) the decoder is told how far it may go
      hipLaunchKernelGGL(ScatterRowsKernel, dim3(static_cast<unsigned>(std::min<size_t>(n_score, 4096))), dim3(256), 0, st, h->ll,
                         h->out_stride, h->d_out.p, h->out_stride, h->d_idx.p + score_b, h->d_idx.p + score_b + n_score,
                         static_cast<int>(n_score), h->out_dim);
      KH_LAUNCH_CHECK();
      KH_HIP(hipStreamSynchronize(st));
      h->avail.resize(n_act);
      for (int k = 0; k < n_act; k++) {
        h->decoded[h->act[k]] += h->nfr[k];
        h->avail[k] = h->decoded[h->act[k]];
      }
      if ((rc = kh_online_decoder_serve_publish(h->dec, h->act.data(), n_act, h->avail.data()))) return rc;
    } else {
      if ((rc = kh_online_decoder_advance(h->dec, h->act.data(), n_act, h->ptrs.data(), h->out_stride, h->nfr.data(), tid2pdf))) return rc;
      for (int k = 0; k < n_act; k++) h->decoded[h->act[k]] += h->nfr[k];
    }
  } else {
    KH_HIP(hipStreamSynchronize(st));
  }
  if (h->serving) {   // what the decoder has reached so far (it runs behind the submissions)
    if (frames_decoded && (rc = kh_online_decoder_serve_poll(h->dec, streams, n, frames_decoded, nullptr))) return rc;
  } else if (frames_decoded) {
    for (int i = 0; i < n; i++) frames_decoded[i] = h->decoded[streams[i]];
  }
  return KH_OK;
}

// Serving through the decoder's persistent kernel: from here on kh_online_nnet2_step returns when the chunk's scores are
// in the streams' buffers and published; the decoder follows at its own pace (kh_online_nnet2_serve_poll / _wait), a
// finished utterance is finalized by kh_online_nnet2_serve_finalize (asynchronous) and read through the online decoder's
// getters after kh_online_nnet2_serve_wait.  tid2pdf as kh_online_nnet2_step.
int kh_online_nnet2_serve_start(KhOnlineNnet2 *h, const int32_t *tid2pdf) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(h);
  if (h->serving) return KH_OK;
  if (!h->ll) {
    h->ll = static_cast<float *>(PoolMalloc(sizeof(float) * static_cast<size_t>(h->num_streams) * h->max_frames * h->out_stride));
    if (!h->ll) {
      SetError("kh_online_nnet2_serve_start: no device memory for %d streams x %d frames of scores", h->num_streams, h->max_frames);
      return KH_ENOMEM;
    }
  }
  h->serve_map = tid2pdf;
  if ((rc = kh_online_decoder_serve_start(h->dec, h->ll, h->out_stride, h->max_frames, tid2pdf))) return rc;
  h->serving = true;
  return KH_OK;
}

int kh_online_nnet2_serve_stop(KhOnlineNnet2 *h) {
  KH_CHECK_ARG(h);
  if (!h->serving) return KH_OK;
  h->serving = false;
  return kh_online_decoder_serve_stop(h->dec);
}

// FinalizeDecoding of the listed streams once everything submitted to them is decoded (asynchronous).
int kh_online_nnet2_serve_finalize(KhOnlineNnet2 *h, const int32_t *streams, int n) {
  KH_CHECK_ARG(h && h->serving);
  return kh_online_decoder_serve_finalize(h->dec, streams, n);
}

// NumFramesDecoded() of the listed streams so far / whether InitDecoding or FinalizeDecoding is still in flight.
int kh_online_nnet2_serve_poll(KhOnlineNnet2 *h, const int32_t *streams, int n, int32_t *decoded, int32_t *in_flight) {
  KH_CHECK_ARG(h && h->serving);
  return kh_online_decoder_serve_poll(h->dec, streams, n, decoded, in_flight);
}

int kh_online_nnet2_serve_wait(KhOnlineNnet2 *h, const int32_t *streams, int n, int timeout_ms) {
  KH_CHECK_ARG(h && h->serving);
  return kh_online_decoder_serve_wait(h->dec, streams, n, timeout_ms);
}

// NumFramesReady() (:75-89) of a stream, and whether `frame` is its last one (IsLastFrame :67-73; -1 = unknown yet).
int kh_online_nnet2_num_frames_ready(const KhOnlineNnet2 *h, int stream, int32_t *ready) {
  KH_CHECK_ARG(h && ready && stream >= 0 && stream < h->num_streams);
  const int have = h->n[stream];
  *ready = have == 0 ? 0
                     : (h->pad_input ? (h->finished[stream] ? have : std::max(0, have - h->R)) : std::max(0, have - h->R - h->L));
  return KH_OK;
}

}  // extern "C"
