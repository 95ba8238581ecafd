"""Python host-side mirror of the reference's operator interface for the hot path.

Same names / argument meaning / error behaviour as the reference classes
(`CuMatrixBase` methods of cudamatrix/cu-matrix.h, `cu::Splice`, `DiagGmm`,
`nnet2::Nnet` + `NnetComputation` + `DecodableAmNnet`, `LatticeFasterDecoder`,
`LatticeForwardBackward`), implemented by calling the C-ABI of libkaldi_hip.so.
torch is used only as the owner of device memory (tensors -> data_ptr()) and of
the HIP stream; no computation is done by torch and there is no fallback: every
function raises KhError (the KALDI_ERR equivalent) if the library call fails.
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

from . import capi
from .capi import KhMatrixDim, KhComponentDesc, KhDecoderConfig, KhDecodeStats, check, KhError

kTrans, kNoTrans = 1, 0

TYPE_BY_NAME = {
    "splice": 1, "fixed_affine": 2, "affine": 3, "pnorm": 4, "normalize": 5,
    "softmax": 6, "sum_group": 7, "fixed_scale": 8, "fixed_bias": 9,
}


def lib():
    return capi.load()


def use_torch_stream():
    """Enqueue library work on torch's current HIP stream (ordering with tensor
    creation / copies done by torch)."""
    check(lib().kh_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def synchronize():
    """Wait for the library's stream (CU_SAFE_CALL's cudaThreadSynchronize, cu-common.h:37-44)."""
    check(lib().kh_synchronize())


def pool_release():
    """Return the blocks cached by the library's device-memory pool to the driver (kh_pool_release)."""
    check(lib().kh_pool_release())


def select_gpu(ordinal):
    """CuDevice::SelectGpuId with explicit ordinal (cu-device.cc:93-192)."""
    check(lib().kh_select_gpu(int(ordinal)))
    torch.cuda.set_device(int(ordinal))
    use_torch_stream()


def _fn(name, t, *others):
    """The entry point for the tensors' element type: kh_<name> (float) or its <double> twin kh_<name>_d.  Every
    floating-point operand of the call must have that one type (a float operand handed to the double kernel - or the
    other way round - would be reinterpreted, not converted)."""
    for o in others:
        assert not o.dtype.is_floating_point or o.dtype == t.dtype, "operands of different element types: %s and %s" % (t.dtype, o.dtype)
    if t.dtype == torch.float64:
        return getattr(lib(), name + "_d")
    assert t.dtype == torch.float32
    return getattr(lib(), name)


def _dim(t, dtype=torch.float32):
    """MatrixDim of a 2-D device tensor of element type `dtype` - float32 unless the caller dispatches on the type (_fn):
    the entry points without a <double> twin (affine, the network, the decoder, the GMM, iVectors, lattices) take float only."""
    assert t.dim() == 2 and t.dtype == dtype and t.is_cuda, "a 2-D %s device tensor is expected, got %s" % (dtype, t.dtype)
    assert t.stride(1) == 1 or t.shape[1] <= 1
    return KhMatrixDim(t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _p(t):
    return C.c_void_p(t.data_ptr())


def _dev_i32(a, device):
    if isinstance(a, torch.Tensor):
        assert a.dtype == torch.int32 and a.is_cuda
        return a
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=device)


# ---------------------------------------------------------------- CuMatrix ops
def add_mat_mat(Cm, alpha, A, transA, B, transB, beta):
    """CuMatrixBase::AddMatMat (cu-matrix.cc:947-982): C = alpha op(A) op(B) + beta C."""
    check(_fn("kh_add_mat_mat", Cm, A, B)(alpha, _p(A), _dim(A, Cm.dtype), int(transA), _p(B), _dim(B, Cm.dtype), int(transB), beta, _p(Cm), _dim(Cm, Cm.dtype)))
    return Cm


def affine(out, A, W, bias):
    check(lib().kh_affine(_p(A), _dim(A), _p(W), _dim(W), _p(bias), _p(out), _dim(out)))


def affine_pnorm(out, A, W, bias):
    """AffineComponent + PnormComponent (p = 2) in one kernel: out[r][c] = 2-norm of group c of row r of A W^T + bias
    (nnet-component.cc:1219-1224, :386-391); group size = W.shape[0] // out.shape[1]."""
    check(lib().kh_affine_pnorm(_p(A), _dim(A), _p(W), _dim(W), _p(bias), _p(out), _dim(out), W.shape[0] // out.shape[1]))
    return out


def apply_softmax_per_row(dst, src):
    """CuMatrixBase::ApplySoftMaxPerRow (cu-matrix.cc:1251-1271)."""
    assert dst.shape == src.shape  # KALDI_ASSERT(SameDim(*this, src))
    check(_fn("kh_softmax_per_row", dst, src)(_p(dst), _p(src), _dim(dst, dst.dtype), _dim(src, dst.dtype).stride))
    return dst


def apply_log_softmax_per_row(dst, src):
    assert dst.shape == src.shape
    check(_fn("kh_log_softmax_per_row", dst, src)(_p(dst), _p(src), _dim(dst, dst.dtype), _dim(src, dst.dtype).stride))
    return dst


def copy_rows(dst, src, indices):
    """CuMatrixBase::CopyRows (cu-matrix.cc:1965-1990)."""
    idx = _dev_i32(indices, dst.device)
    if dst.shape[1] != src.shape[1] or dst.shape[0] != idx.numel():
        raise KhError("CopyRows: dimension mismatch")
    check(_fn("kh_copy_rows", dst, src, idx)(_p(dst), _dim(dst, dst.dtype), _p(src), _dim(src, dst.dtype).stride, _p(idx)))
    return dst


def splice(src, frame_offsets, tgt):
    """cu::Splice (cudamatrix/cu-math.cc:130-165)."""
    off = _dev_i32(frame_offsets, src.device)
    check(_fn("kh_splice", tgt, src, off)(_p(tgt), _dim(tgt, tgt.dtype), _p(src), _dim(src, tgt.dtype), _p(off), off.numel()))
    return tgt


def group_pnorm(dst, src, power):
    """CuMatrixBase::GroupPnorm (cu-matrix.cc:1147-1164)."""
    if src.shape[1] % dst.shape[1] != 0 or src.shape[0] != dst.shape[0]:
        raise KhError("GroupPnorm: dimension mismatch")
    check(_fn("kh_group_pnorm", dst, src)(_p(dst), _p(src), _dim(dst, dst.dtype), _dim(src, dst.dtype).stride, src.shape[1] // dst.shape[1], power))
    return dst


def normalize(dst, src):
    """NormalizeComponent::Propagate (nnet2/nnet-component.cc:576-588)."""
    assert dst.shape == src.shape
    check(lib().kh_normalize(_p(dst), _p(src), _dim(dst), _dim(src).stride))
    return dst


def add_diag_mat2(v, alpha, M, beta):
    """CuVectorBase::AddDiagMat2 kNoTrans (cu-vector.cc:517-580)."""
    assert v.numel() == M.shape[0]
    check(_fn("kh_add_diag_mat2", M, v)(alpha, _p(M), _dim(M, M.dtype), beta, _p(v)))
    return v


def mul_rows_vec(M, s):
    assert s.numel() == M.shape[0]
    check(_fn("kh_mul_rows_vec", M, s)(_p(M), _dim(M, M.dtype), _p(s)))
    return M


def mul_cols_vec(M, s):
    assert s.numel() == M.shape[1]
    check(_fn("kh_mul_cols_vec", M, s)(_p(M), _dim(M, M.dtype), _p(s)))
    return M


def copy_rows_from_vec(M, v):
    assert v.numel() == M.shape[1]
    check(_fn("kh_copy_rows_from_vec", M, v)(_p(M), _dim(M, M.dtype), _p(v)))
    return M


def add_vec_to_rows(M, alpha, v, beta=1.0):
    assert v.numel() == M.shape[1]
    check(_fn("kh_add_vec_to_rows", M, v)(alpha, _p(v), beta, _p(M), _dim(M, M.dtype)))
    return M


def apply_floor(M, f):
    check(_fn("kh_apply_floor", M)(_p(M), _dim(M, M.dtype), f))
    return M


def apply_log(M):
    check(_fn("kh_apply_log", M)(_p(M), _dim(M, M.dtype)))
    return M


def apply_exp(M):
    check(_fn("kh_apply_exp", M)(_p(M), _dim(M, M.dtype)))
    return M


def apply_pow(M, p):
    check(_fn("kh_apply_pow", M)(_p(M), _dim(M, M.dtype), p))
    return M


def scale(M, a):
    check(_fn("kh_scale", M)(_p(M), _dim(M, M.dtype), a))
    return M


def sum_column_ranges(dst, src, ranges):
    """CuMatrixBase::SumColumnRanges (cu-matrix.cc:1994-2028)."""
    r = _dev_i32(ranges, dst.device)
    assert r.numel() == 2 * dst.shape[1]
    check(_fn("kh_sum_column_ranges", dst, src, r)(_p(dst), _dim(dst, dst.dtype), _p(src), _dim(src, dst.dtype), _p(r)))
    return dst


def lookup(M, pairs):
    """CuMatrixBase::Lookup (cu-matrix.cc:2327)."""
    pr = _dev_i32(pairs, M.device)
    n = pr.numel() // 2
    out = torch.empty(n, dtype=M.dtype, device=M.device)
    check(_fn("kh_matrix_lookup", M, pr, out)(_p(M), _dim(M, M.dtype), _p(pr), n, _p(out)))
    return out


# ---------------------------------------------------------------- nnet2
class Nnet:
    """nnet2::Nnet (forward only) + NnetComputation + DecodableAmNnet epilogue."""

    def __init__(self, components, priors=None):
        self._h = C.c_void_p(lib().kh_nnet_create())
        self._keep = []
        for comp in components:
            d = KhComponentDesc()
            d.type = TYPE_BY_NAME[comp["type"]]
            d.input_dim, d.output_dim = int(comp["input_dim"]), int(comp["output_dim"])
            keep = []
            if "linear" in comp:
                w = np.ascontiguousarray(comp["linear"], np.float32)
                keep.append(w)
                d.linear = w.ctypes.data_as(capi.c_float_p)
            if "bias" in comp:
                b = np.ascontiguousarray(comp["bias"], np.float32)
                keep.append(b)
                d.bias = b.ctypes.data_as(capi.c_float_p)
            if "context" in comp:
                ctx = np.ascontiguousarray(comp["context"], np.int32)
                keep.append(ctx)
                d.context = ctx.ctypes.data_as(capi.c_int32_p)
                d.n_context = len(ctx)
                d.const_dim = int(comp.get("const_dim", 0))
            d.p = float(comp.get("p", 2.0))
            if "sizes" in comp:
                sz = np.ascontiguousarray(comp["sizes"], np.int32)
                keep.append(sz)
                d.sizes = sz.ctypes.data_as(capi.c_int32_p)
                d.n_sizes = len(sz)
            check(lib().kh_nnet_add_component(self._h, C.byref(d)))
        if priors is not None:
            pr = np.ascontiguousarray(priors, np.float32)
            check(lib().kh_nnet_set_priors(self._h, pr.ctypes.data_as(capi.c_float_p), len(pr)))

    def __del__(self):
        try:
            if self._h:
                lib().kh_nnet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def input_dim(self):
        return lib().kh_nnet_input_dim(self._h)

    def output_dim(self):
        return lib().kh_nnet_output_dim(self._h)

    def left_context(self):
        return lib().kh_nnet_left_context(self._h)

    def right_context(self):
        return lib().kh_nnet_right_context(self._h)

    def compute(self, feats, utt_row_offsets=None, pad_input=True, epilogue=False, prob_scale=1.0, out=None, wait=True):
        """NnetComputation (nnet-compute.cc:159-166) for a batch stacked by rows;
        epilogue=True adds DecodableAmNnet's floor/log/-logprior/scale.
        wait=False: kh_nnet_compute_async - the call returns when the work is queued on the library's stream."""
        T = feats.shape[0]
        if utt_row_offsets is None:
            utt_row_offsets = [0, T]
        off = np.ascontiguousarray(utt_row_offsets, np.int32)
        n_utts = len(off) - 1
        L, R = self.left_context(), self.right_context()
        rows = int(off[-1] - off[0]) if pad_input else int(off[-1] - off[0]) - n_utts * (L + R)
        if rows <= 0:
            raise KhError("Nnet.compute: no output rows")
        if out is None:
            od = self.output_dim()
            stride = (od + 3) // 4 * 4
            out = torch.empty((rows, stride), dtype=torch.float32, device=feats.device)[:, :od]
        out_off = np.zeros(n_utts + 1, np.int32)
        check((lib().kh_nnet_compute if wait else lib().kh_nnet_compute_async)(self._h, _p(feats), _dim(feats).stride,
                                    off.ctypes.data_as(capi.c_int32_p), n_utts, int(pad_input),
                                    int(epilogue), float(prob_scale), _p(out), _dim(out).stride,
                                    out_off.ctypes.data_as(capi.c_int32_p)))
        return out, out_off


class DecodableNnet2Online:
    """nnet2/online-nnet2-decodable.{h,cc} for num_streams concurrent utterances: the
    feature rows of a stream arrive in chunks (the OnlineFeatureInterface side:
    accept_features / input_finished), NumFramesReady() follows :75-89, and compute()
    is ComputeForFrame (:91-143) for a list of streams at once — one gather of the
    context-padded input rows (first / last frame duplicated past the edges when
    pad_input), one NnetComputation(pad_input=false) over all of them, the
    floor / log / -log prior / acoustic scale epilogue."""

    def __init__(self, nnet, num_streams, max_frames, acoustic_scale=0.1, pad_input=True, max_nnet_batch_size=256,
                 device="cuda"):
        assert max_nnet_batch_size > 0  # :40
        self.nnet, self.scale, self.pad_input, self.max_batch = nnet, float(acoustic_scale), bool(pad_input), int(max_nnet_batch_size)
        self.num_streams, self.max_frames = int(num_streams), int(max_frames)
        self.L, self.R = nnet.left_context(), nnet.right_context()
        dim = nnet.input_dim()
        stride = (dim + 3) // 4 * 4
        self._feats = torch.empty((self.num_streams * self.max_frames, stride), dtype=torch.float32, device=device)[:, :dim]
        self._n = [0] * self.num_streams
        self._finished = [False] * self.num_streams

    def reset(self, streams):
        for s in streams:
            self._n[s], self._finished[s] = 0, False

    def accept_features(self, stream, feats, input_finished=False):
        """Append rows (host array or device tensor) to the stream's features."""
        assert not self._finished[stream], "input already finished"
        x = feats if torch.is_tensor(feats) else torch.from_numpy(np.ascontiguousarray(feats, np.float32))
        k = x.shape[0]
        if self._n[stream] + k > self.max_frames:
            raise KhError("DecodableNnet2Online: more than max_frames=%d feature frames" % self.max_frames)
        if k:
            b = stream * self.max_frames + self._n[stream]
            self._feats[b:b + k].copy_(x)
        self._n[stream] += k
        self._finished[stream] = bool(input_finished)

    def accept_features_many(self, streams, src, src_rows, counts, finished):
        """accept_features for many streams with ONE device copy: streams[i] takes rows [src_rows[i], src_rows[i] + counts[i])
        of the device matrix `src` (what a batched feature front end hands over per chunk); finished[i]: InputFinished()."""
        src_idx, dst_idx = [], []
        for s, r, k, fin in zip(streams, src_rows, counts, finished):
            assert not self._finished[s], "input already finished"
            if self._n[s] + k > self.max_frames:
                raise KhError("DecodableNnet2Online: more than max_frames=%d feature frames" % self.max_frames)
            if k:
                src_idx.append(np.arange(r, r + k))
                dst_idx.append(s * self.max_frames + self._n[s] + np.arange(k))
            self._n[s] += int(k)
            self._finished[s] = bool(fin)
        if src_idx:
            si = torch.from_numpy(np.concatenate(src_idx)).to(self._feats.device)
            di = torch.from_numpy(np.concatenate(dst_idx)).to(self._feats.device)
            self._feats.index_copy_(0, di, src.index_select(0, si))

    def num_frames_ready(self, stream):
        """NumFramesReady() :75-89."""
        ready = self._n[stream]
        if ready == 0:
            return 0
        if self.pad_input:
            return ready if self._finished[stream] else max(0, ready - self.R)
        return max(0, ready - self.R - self.L)

    def is_last_frame(self, stream, frame):
        """IsLastFrame() :67-73."""
        last = self._n[stream] - 1 if self._finished[stream] else None
        if last is None:
            return False
        return frame == last if self.pad_input else frame + self.L + self.R == last

    def compute(self, streams, frames):
        """Scaled log-likelihoods of frames [frames[i], frames[i] + n_i) of streams[i],
        n_i = min(NumFramesReady - frames[i], max_nnet_batch_size) (ComputeForFrame).
        Returns one device matrix per stream (0 rows if nothing is ready)."""
        idx, off, n_out = [], [0], []
        for s, f in zip(streams, frames):
            ready = self._n[s]
            n = max(0, min(self.num_frames_ready(s) - f, self.max_batch))
            n_out.append(n)
            if n == 0:
                continue
            begin = f - self.L if self.pad_input else f          # :103-107
            t = np.arange(begin, begin + n + self.L + self.R)
            t = np.clip(t, 0, ready - 1)                            # :118-124 (only ever clips when pad_input)
            idx.append(s * self.max_frames + t)
            off.append(off[-1] + len(t))
        od = self.nnet.output_dim()
        if not idx:
            return [torch.empty((0, od), dtype=torch.float32, device=self._feats.device) for _ in streams]
        idx = np.concatenate(idx).astype(np.int32)
        stride = self._feats.stride(0)
        x = torch.empty((len(idx), stride), dtype=torch.float32, device=self._feats.device)[:, :self._feats.shape[1]]
        copy_rows(x, self._feats, idx)
        out, out_off = self.nnet.compute(x, off, pad_input=False, epilogue=True, prob_scale=self.scale)
        res, k = [], 0
        for n in n_out:
            if n == 0:
                res.append(out[0:0])
            else:
                res.append(out[int(out_off[k]):int(out_off[k]) + n])
                k += 1
        return res


class OnlineNnet2Pipeline:
    """The serving loop of online2-wav-nnet2-latgen-faster (:213-262) for num_streams concurrent utterances with ONE library
    call per step (kh_online_nnet2_*): step() hands every live stream its chunk of feature rows, DecodableNnet2Online's
    ComputeForFrame runs for all advancing streams in one forward pass, AdvanceDecoding consumes the frames.  `decoder` is a
    LatticeFasterOnlineDecoder (results are read from it: get_best_path, get_raw_lattice, finalize_decoding)."""

    def __init__(self, nnet, decoder, max_frames, acoustic_scale=0.1, pad_input=True, max_nnet_batch_size=256):
        self.nnet, self.decoder = nnet, decoder
        h = lib().kh_online_nnet2_create(nnet._h, decoder._h, decoder.num_streams, int(max_frames), float(acoustic_scale),
                                         int(bool(pad_input)), int(max_nnet_batch_size))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)
        fst = decoder.fst
        self._t2p = _p(fst.tid2pdf) if fst.tid2pdf is not None else None
        fst.check_pdf_map(nnet.output_dim())
        # the decoder's arc records for this (map, columns) pair once, not at every chunk
        check(lib().kh_online_decoder_set_pdf_map(decoder._h, self._t2p, (nnet.output_dim() + 3) // 4 * 4))

    def __del__(self):
        try:
            if self._h:
                lib().kh_online_nnet2_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self, streams):
        st = np.ascontiguousarray(streams, np.int32)
        check(lib().kh_online_nnet2_reset(self._h, st.ctypes.data_as(capi.c_int32_p), len(st)))

    def step(self, streams, src, src_rows, counts, finished):
        """streams[i] receives rows [src_rows[i], src_rows[i] + counts[i]) of the device matrix src; finished[i]:
        InputFinished().  Returns NumFramesDecoded() of the listed streams after the step."""
        st = np.ascontiguousarray(streams, np.int32)
        sr = np.ascontiguousarray(src_rows, np.int32)
        ct = np.ascontiguousarray(counts, np.int32)
        fi = np.ascontiguousarray(finished, np.int32)
        out = np.empty(len(st), np.int32)
        ip = capi.c_int32_p
        check(lib().kh_online_nnet2_step(self._h, st.ctypes.data_as(ip), len(st), _p(src), _dim(src).stride, sr.ctypes.data_as(ip),
                                         ct.ctypes.data_as(ip), fi.ctypes.data_as(ip), self._t2p, out.ctypes.data_as(ip)))
        return out

    # ---- serving through the decoder's persistent kernel (kh_online_nnet2_serve_*): step() then returns once the chunk's
    # scores are published, the decoder follows at its own pace (a stream that prunes does not hold up the others)
    def serve_start(self):
        check(lib().kh_online_nnet2_serve_start(self._h, self._t2p))

    def serve_stop(self):
        check(lib().kh_online_nnet2_serve_stop(self._h))

    def serve_finalize(self, streams):
        """FinalizeDecoding of the streams once everything submitted to them is decoded (asynchronous)."""
        st = np.ascontiguousarray(streams, np.int32)
        check(lib().kh_online_nnet2_serve_finalize(self._h, st.ctypes.data_as(capi.c_int32_p), len(st)))

    def serve_poll(self, streams):
        """(NumFramesDecoded() so far, request still in flight) of the streams."""
        st = np.ascontiguousarray(streams, np.int32)
        dec, fl = np.empty(len(st), np.int32), np.empty(len(st), np.int32)
        ip = capi.c_int32_p
        check(lib().kh_online_nnet2_serve_poll(self._h, st.ctypes.data_as(ip), len(st), dec.ctypes.data_as(ip), fl.ctypes.data_as(ip)))
        return dec, fl.astype(bool)

    def serve_wait(self, streams, timeout_ms=0):
        st = np.ascontiguousarray(streams, np.int32)
        check(lib().kh_online_nnet2_serve_wait(self._h, st.ctypes.data_as(capi.c_int32_p), len(st), int(timeout_ms)))


# ---------------------------------------------------------------- feature front-end
class Mfcc:
    """feat/feature-mfcc.h Mfcc with MfccOptions: the constructor builds the reference's tables
    (window function, MelBanks, DCT rows, lifter) on the host, compute() runs the frames on the
    GPU.  Defaults here are the recipes' (--use-energy=false, no dither); the reference's
    struct defaults are use_energy = true, dither = 1.0 (feature-mfcc.h:54, feature-functions.h:91).
    dither > 0 draws from a counter-based generator seeded with dither_seed."""

    def __init__(self, samp_freq=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type="povey", num_bins=23, low_freq=20.0, high_freq=0.0, num_ceps=13,
                 cepstral_lifter=22.0, snip_edges=True, use_energy=False, raw_energy=True, energy_floor=0.0,
                 htk_compat=False, dither=0.0, dither_seed=0):
        f32 = np.float32
        self.opts = capi.KhMfccOptions(int(snip_edges), int(use_energy), int(raw_energy), int(htk_compat),
                                       float(energy_floor), float(dither), int(dither_seed))
        self.samp_freq = float(samp_freq)
        self.frame_shift = int(f32(samp_freq) * f32(0.001) * f32(frame_shift_ms))      # WindowShift() feature-functions.h:119
        self.frame_length = int(f32(samp_freq) * f32(0.001) * f32(frame_length_ms))    # WindowSize()
        self.padded = 1 << int(np.ceil(np.log2(self.frame_length)))                    # PaddedWindowSize()
        self.preemph, self.remove_dc = float(preemph_coeff), bool(remove_dc_offset)
        self.num_bins, self.num_ceps = int(num_bins), int(num_ceps)
        n = self.frame_length
        i = np.arange(n, dtype=np.float32).astype(np.float64)
        a = 2.0 * np.pi * i / (n - 1)
        if window_type == "hanning":                                                    # FeatureWindowFunction :74-92
            w = 0.5 - 0.5 * np.cos(a)
        elif window_type == "hamming":
            w = 0.54 - 0.46 * np.cos(a)
        elif window_type == "povey":
            w = np.power(0.5 - 0.5 * np.cos(a), 0.85)
        elif window_type == "rectangular":
            w = np.ones(n)
        else:
            raise KhError("Invalid window type " + window_type)
        self.window = w.astype(np.float32)
        # MelBanks (vtln_warp 1.0) mel-computations.cc:33-152, float arithmetic as there
        mel = lambda fr: f32(1127.0) * np.log(f32(1.0) + np.asarray(fr, np.float32) / f32(700.0), dtype=np.float32)
        nyquist = f32(0.5) * f32(samp_freq)
        hf = f32(high_freq) if high_freq > 0.0 else nyquist + f32(high_freq)
        lf = f32(low_freq)
        if lf < 0.0 or lf >= nyquist or hf <= 0.0 or hf > nyquist or hf <= lf:
            raise KhError("Bad values in options: low-freq %g and high-freq %g vs. nyquist %g" % (lf, hf, nyquist))
        num_fft_bins = self.padded // 2
        fft_bin_width = f32(samp_freq) / f32(self.padded)
        mel_low, mel_high = mel(lf), mel(hf)
        delta = (mel_high - mel_low) / f32(num_bins + 1)
        mels = mel(fft_bin_width * np.arange(num_fft_bins, dtype=np.float32))
        first, off, weights = [], [0], []
        for b in range(num_bins):
            left, center, right = mel_low + f32(b) * delta, mel_low + f32(b + 1) * delta, mel_low + f32(b + 2) * delta
            idx = np.nonzero((mels > left) & (mels < right))[0]
            if len(idx) == 0:
                raise KhError("You may have set --num-mel-bins too large.")
            m = mels[idx[0]:idx[-1] + 1]
            wts = np.where(m <= center, (m - left) / (center - left), (right - m) / (right - center)).astype(np.float32)
            wts[~((m > left) & (m < right))] = 0.0
            first.append(int(idx[0]))
            weights.append(wts)
            off.append(off[-1] + len(wts))
        self.mel_first = np.asarray(first, np.int32)
        self.mel_off = np.asarray(off, np.int32)
        self.mel_weights = np.concatenate(weights).astype(np.float32)
        # ComputeDctMatrix matrix-functions.cc:592-608 (first num_ceps rows)
        N = num_bins
        dct = np.empty((num_ceps, N), np.float32)
        dct[0] = f32(np.sqrt(1.0 / float(f32(N))))
        norm = f32(np.sqrt(2.0 / float(f32(N))))
        k = np.arange(1, num_ceps)[:, None]
        nn = np.arange(N)[None, :]
        dct[1:] = (float(norm) * np.cos(np.pi / N * (nn + 0.5) * k)).astype(np.float32)
        self.dct = np.ascontiguousarray(dct)
        self.lifter = None
        if cepstral_lifter != 0.0:                                                      # mel-computations.cc:248-254
            q = float(f32(cepstral_lifter))
            self.lifter = (1.0 + 0.5 * q * np.sin(np.pi * np.arange(num_ceps) / q)).astype(np.float32)

    def num_frames(self, n_samples):
        """NumFrames feature-functions.cc:29-48."""
        if not self.opts.snip_edges:
            return int(np.float32(n_samples) * np.float32(1.0) / np.float32(self.frame_shift) + np.float32(0.5))
        return 0 if n_samples < self.frame_length else 1 + (n_samples - self.frame_length) // self.frame_shift

    def compute(self, wave):
        """Mfcc::Compute(wave, 1.0, &output): wave = 1-D float32 device tensor."""
        rows = self.num_frames(wave.numel())
        stride = (self.num_ceps + 3) // 4 * 4
        out = torch.empty((max(rows, 1), stride), dtype=torch.float32, device=wave.device)[:rows, :self.num_ceps]
        nf = C.c_int32()
        fp, ip = capi.c_float_p, capi.c_int32_p
        check(lib().kh_mfcc_compute_opts(
            C.c_void_p(wave.data_ptr()), wave.numel(), self.frame_shift, self.frame_length, self.padded, self.preemph,
            int(self.remove_dc), self.window.ctypes.data_as(fp), self.num_bins, self.mel_first.ctypes.data_as(ip),
            self.mel_off.ctypes.data_as(ip), self.mel_weights.ctypes.data_as(fp), self.num_ceps,
            self.dct.ctypes.data_as(fp), self.lifter.ctypes.data_as(fp) if self.lifter is not None else None,
            C.byref(self.opts), C.c_void_p(out.data_ptr()) if rows else None, stride, C.byref(nf)))
        assert nf.value == rows
        return out


def delta_scales(order, window):
    """DeltaFeatures::DeltaFeatures feat/feature-functions.cc:210-242."""
    scales = [np.ones(1, np.float32)]
    for _ in range(order):
        prev = scales[-1]
        prev_offset = (len(prev) - 1) // 2
        cur = np.zeros(len(prev) + 2 * window, np.float32)
        normalizer = np.float32(0.0)
        for j in range(-window, window + 1):
            normalizer = np.float32(normalizer + np.float32(j * j))
            for k in range(-prev_offset, prev_offset + 1):
                cur[j + k + prev_offset + window] = np.float32(cur[j + k + prev_offset + window] + np.float32(j) * prev[k + prev_offset])
        scales.append((cur * np.float32(1.0 / float(normalizer))).astype(np.float32))
    return scales


def compute_deltas(feats, order=2, window=2):
    """ComputeDeltas (feat/feature-functions.cc:361-372): feats [T x D] device -> [T x D(order+1)]."""
    if not (0 <= order < 1000 and 0 < window < 1000):
        raise KhError("DeltaFeaturesOptions: bad order / window")
    sc = delta_scales(order, window)
    flat = np.ascontiguousarray(np.concatenate(sc), np.float32)
    lens = np.ascontiguousarray([len(x) for x in sc], np.int32)
    cols = feats.shape[1] * (order + 1)
    stride = (cols + 3) // 4 * 4
    out = torch.empty((feats.shape[0], stride), dtype=torch.float32, device=feats.device)[:, :cols]
    check(lib().kh_compute_deltas(_p(feats), _dim(feats), int(order), flat.ctypes.data_as(capi.c_float_p),
                                  lens.ctypes.data_as(capi.c_int32_p), _p(out), stride))
    return out


def acc_cmvn_stats(feats, stats=None):
    """AccCmvnStats (transform/cmvn.cc:49-62): stats [2 x (dim + 1)] float64 on the host."""
    st = np.zeros((2, feats.shape[1] + 1)) if stats is None else np.ascontiguousarray(stats, np.float64)
    check(lib().kh_acc_cmvn_stats(_p(feats), _dim(feats), st.ctypes.data_as(capi.c_double_p)))
    return st


def apply_cmvn(stats, var_norm, feats):
    """ApplyCmvn (transform/cmvn.cc:64-113), in place on the device matrix."""
    dim = stats.shape[1] - 1
    if stats.shape[0] not in (1, 2) or feats.shape[1] != dim:
        raise KhError("Dim mismatch: cmvn %dx%d, feats %dx%d" % (stats.shape + tuple(feats.shape)))
    if stats.shape[0] == 1 and var_norm:
        raise KhError("You requested variance normalization but no variance stats are supplied.")
    count = float(stats[0, dim])
    if count < 1.0:
        raise KhError("Insufficient stats for cepstral mean and variance normalization: count = %g" % count)
    mean = stats[0, :dim] / count
    if not var_norm:
        scale, offset = np.ones(dim), -mean
    else:
        var = np.maximum(stats[1, :dim] / count - mean * mean, 1.0e-20)
        scale = 1.0 / np.sqrt(var)
        offset = -(mean * scale)
    dev = feats.device
    if var_norm:
        mul_cols_vec(feats, torch.from_numpy(scale.astype(np.float32)).to(dev))
    add_vec_to_rows(feats, 1.0, torch.from_numpy(offset.astype(np.float32)).to(dev))
    return feats


# ---------------------------------------------------------------- DiagGmm
def gmm_compute_gconsts(weights, means_invvars, inv_vars):
    """DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-152); host arrays."""
    w = np.ascontiguousarray(weights, np.float32)
    mi = np.ascontiguousarray(means_invvars, np.float32)
    iv = np.ascontiguousarray(inv_vars, np.float32)
    g = np.empty(len(w), np.float32)
    fp = capi.c_float_p
    rc = lib().kh_gmm_compute_gconsts(w.ctypes.data_as(fp), mi.ctypes.data_as(fp), iv.ctypes.data_as(fp),
                                      mi.shape[0], mi.shape[1], g.ctypes.data_as(fp))
    if rc < 0:
        check(rc)
    return g, rc


class AmDiagGmm:
    """All pdfs' DiagGmm parameters concatenated, resident on the device
    (AmDiagGmm gmm/am-diag-gmm.h; scoring as DecodableAmDiagGmmUnmapped)."""

    def __init__(self, gconsts, means_invvars, inv_vars, pdf_offsets, device="cuda"):
        self.gconsts = torch.as_tensor(np.ascontiguousarray(gconsts, np.float32), device=device)
        self.means_invvars = torch.as_tensor(np.ascontiguousarray(means_invvars, np.float32), device=device)
        self.inv_vars = torch.as_tensor(np.ascontiguousarray(inv_vars, np.float32), device=device)
        self.pdf_offsets = torch.as_tensor(np.ascontiguousarray(pdf_offsets, np.int32), device=device)
        self.num_mix = self.means_invvars.shape[0]
        self.dim = self.means_invvars.shape[1]
        self.num_pdfs = self.pdf_offsets.numel() - 1

    def log_likelihoods(self, data, out=None):
        """DiagGmm::LogLikelihoods(Matrix) (diag-gmm.cc:546-562) over ALL Gaussians: T x M."""
        if data.shape[0] == 0:
            raise KhError("KALDI_ASSERT: data.NumRows() != 0")
        if data.shape[1] != self.dim:
            raise KhError("DiagGmm::ComponentLogLikelihood, dimension mismatch %d vs. %d" % (data.shape[1], self.dim))
        if out is None:
            out = torch.empty((data.shape[0], self.num_mix), dtype=torch.float32, device=data.device)
        check(lib().kh_diag_gmm_loglikes(_p(data), _dim(data), _p(self.gconsts), _p(self.means_invvars),
                                         _p(self.inv_vars), self.num_mix, _p(out), _dim(out).stride))
        return out

    def pdf_log_likelihoods(self, data, log_sum_exp_prune=-1.0, out=None):
        """frame x pdf matrix (gmm-compute-likes.cc:70-77)."""
        if data.shape[1] != self.dim:
            raise KhError("Dim mismatch: data dim = %d vs. model dim = %d" % (data.shape[1], self.dim))
        if out is None:
            out = torch.empty((data.shape[0], self.num_pdfs), dtype=torch.float32, device=data.device)
        check(lib().kh_am_gmm_loglikes(_p(data), _dim(data), _p(self.gconsts), _p(self.means_invvars),
                                       _p(self.inv_vars), _p(self.pdf_offsets), self.num_pdfs, self.num_mix,
                                       float(log_sum_exp_prune), _p(out), _dim(out).stride))
        return out


# ---------------------------------------------------------------- decoder
def decoder_config(beam=16.0, max_active=2147483647, min_active=200, lattice_beam=10.0,
                   prune_interval=25, beam_delta=0.5, hash_ratio=2.0, prune_scale=0.1):
    """LatticeFasterDecoderConfig with the reference's defaults
    (decoder/lattice-faster-decoder.h:58-66)."""
    return dict(beam=beam, max_active=max_active, min_active=min_active, lattice_beam=lattice_beam,
                prune_interval=prune_interval, beam_delta=beam_delta, hash_ratio=hash_ratio,
                prune_scale=prune_scale)


class Fst:
    """fst::Fst<StdArc> (HCLG) resident on the device as CSR."""

    def __init__(self, graph):
        off = np.ascontiguousarray(graph["arc_offsets"], np.int64)
        il = np.ascontiguousarray(graph["ilabel"], np.int32)
        ol = np.ascontiguousarray(graph["olabel"], np.int32)
        w = np.ascontiguousarray(graph["weight"], np.float32)
        ns = np.ascontiguousarray(graph["nextstate"], np.int32)
        fin = np.ascontiguousarray(graph["final"], np.float32)
        h = lib().kh_fst_create(int(graph["num_states"]), int(graph["start"]),
                                off.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(capi.c_int32_p),
                                ol.ctypes.data_as(capi.c_int32_p), w.ctypes.data_as(capi.c_float_p),
                                ns.ctypes.data_as(capi.c_int32_p), fin.ctypes.data_as(capi.c_float_p))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)
        self.tid2pdf = None
        self._tid2pdf_host = None
        self._checked_cols = set()
        if graph.get("tid2pdf") is not None:
            self._tid2pdf_host = np.ascontiguousarray(graph["tid2pdf"], np.int32)
            self.tid2pdf = torch.as_tensor(self._tid2pdf_host, device="cuda")

    def check_pdf_map(self, num_cols):
        """Every transition-id of the graph maps to a column of the log-likelihood matrix
        (checked once per matrix width; KhError otherwise)."""
        if num_cols in self._checked_cols:
            return
        h = self._tid2pdf_host
        check(lib().kh_fst_check_pdf_map(self._h, h.ctypes.data_as(capi.c_int32_p) if h is not None else None,
                                         len(h) if h is not None else 0, int(num_cols)))
        self._checked_cols.add(num_cols)

    def num_arcs(self):
        return lib().kh_fst_num_arcs(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().kh_fst_destroy(self._h)
                self._h = None
        except Exception:
            pass


class LatticeFasterDecoder:
    """decoder/lattice-faster-decoder.h:96-413 for a batch of utterances.

    decode(loglikes, utt_row_offsets) == Decode(&decodable) per utterance with a
    DecodableMatrixScaledMapped-style decodable (decoder/decodable-matrix.h:33-84):
    loglikes[t, tid2pdf[ilabel]] already scaled."""

    def __init__(self, fst, config=None, max_batch=256, max_frames=4096, exact_reference_order=True):
        self.fst = fst
        cfg = decoder_config() if config is None else config
        self.cfg = KhDecoderConfig(**cfg)
        h = lib().kh_decoder_create(fst._h, C.byref(self.cfg), int(max_batch), int(max_frames))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)
        self.n_utts = 0
        if not exact_reference_order:      # (the library's default is the reference's order)
            self.set_reference_order(False)

    def set_reference_order(self, enable):
        """True: the reference's own iteration order (running next_cutoff in HashList order, first-minimum ties, LIFO
        closure insertions: include/kaldi_hip.h kh_decoder_set_reference_order) - tokens and links are the ones
        LatticeFasterDecoder itself creates; the default.  False: the order-independent ("canonical") rule, a cheaper
        kernel whose lattices are the reference's only where no token lies between the final and the running cutoff."""
        check(lib().kh_decoder_set_reference_order(self._h, int(bool(enable))))

    def search_counters(self, utt=0):
        c = (C.c_int64 * 2)()
        check(lib().kh_decoder_get_search_counters(self._h, int(utt), c))
        return dict(candidates_materialised=int(c[0]), reference_order=bool(c[1]))

    def __del__(self):
        try:
            if self._h:
                lib().kh_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def decode(self, loglikes, utt_row_offsets=None):
        if utt_row_offsets is None:
            utt_row_offsets = [0, loglikes.shape[0]]
        off = np.ascontiguousarray(utt_row_offsets, np.int32)
        self.n_utts = len(off) - 1
        self._T = np.diff(off)
        t2p = _p(self.fst.tid2pdf) if self.fst.tid2pdf is not None else None
        self.fst.check_pdf_map(loglikes.shape[1])
        self._ll = loglikes  # keep alive
        check(lib().kh_decoder_decode(self._h, _p(loglikes), _dim(loglikes).stride,
                                      off.ctypes.data_as(capi.c_int32_p), self.n_utts, t2p))
        exc, self._after_exc = getattr(self, "_after_exc", None), None
        if exc is not None:
            raise exc

    def stats(self, utt=0):
        st = KhDecodeStats()
        check(lib().kh_decoder_get_stats(self._h, int(utt), C.byref(st)))
        return {k: getattr(st, k) for k, _ in KhDecodeStats._fields_}

    def counters(self, utt=0):
        st = KhDecodeStats()
        check(lib().kh_decoder_get_counters(self._h, int(utt), C.byref(st)))
        return {k: getattr(st, k) for k, _ in KhDecodeStats._fields_}

    def set_determinize(self, enable, beam=None, delta=2.0 ** -10, max_mem=50000000, tid_phone=None, phone_determinize=None,
                        word_determinize=True, minimize=False):
        """LatticeFasterDecoderConfig::determinize_lattice + det_opts (lattice-faster-decoder.h:75-91): the host
        threads that build an utterance's raw lattice while the kernel decodes the rest of the batch also run
        DeterminizeLatticePhonePrunedWrapper on it (decoder-wrappers.cc:264-274); beam defaults to the lattice beam,
        phone_determinize to the reference's default (true) when the transition model's tid_phone map is given."""
        if phone_determinize is None:
            phone_determinize = tid_phone is not None
        tp = np.ascontiguousarray(tid_phone, np.int32) if tid_phone is not None else None
        check(lib().kh_decoder_set_determinize(self._h, int(bool(enable)), float(beam if beam is not None else self.cfg.lattice_beam),
                                               float(delta), int(max_mem), tp.ctypes.data_as(capi.c_int32_p) if tp is not None else None,
                                               len(tp) if tp is not None else 0, int(bool(phone_determinize)), int(bool(word_determinize)),
                                               int(bool(minimize))))

    def get_compact_lattice(self, utt=0):
        """The utterance's determinized CompactLattice (set_determinize(True) first), determinize_lattice_pruned's layout."""
        h = lib().kh_decoder_get_compact_lattice(self._h, int(utt))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        return _read_compact_lattice(C.c_void_p(h))

    def compact_lattice_totals(self):
        """States / arcs / transition-ids of the batch's CompactLattices and how many stopped at the memory limit."""
        out = (C.c_int64 * 4)()
        check(lib().kh_decoder_compact_lattice_totals(self._h, out))
        return dict(states=int(out[0]), arcs=int(out[1]), string_labels=int(out[2]), incomplete=int(out[3]))

    def schedule_counters(self, utt=0):
        """How the pruning schedule treated the utterance (include/kaldi_hip.h kh_decoder_get_schedule_counters)."""
        c = np.zeros(4, np.int32)
        check(lib().kh_decoder_get_schedule_counters(self._h, int(utt), c.ctypes.data_as(capi.c_int32_p)))
        return dict(garbage_collections=int(c[0]), dense_final_visits=int(c[1]), general_final_visits=int(c[2]),
                    handoffs_through_memory=int(c[3]))

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().kh_decoder_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_host_tail_ms(self):
        ms = C.c_float()
        check(lib().kh_decoder_last_host_tail_ms(self._h, C.byref(ms)))
        return ms.value

    def set_after_launch(self, fn):
        """fn() is called inside decode() when its LAST decode kernel has finished (nothing on the device reads this batch's
        scores any more) and before decode() waits for its host threads, on the calling thread: the caller's turn while
        the host finishes the batch - e.g. start the next batch's forward pass, which then runs under this batch's
        determinization.  None: off.  An exception raised by fn is re-raised by decode()."""
        self._after_exc = None
        if fn is None:
            self._after_cb = None
            check(lib().kh_decoder_set_after_launch(self._h, None, None))
            return

        def tramp(_arg):
            try:
                fn()
            except BaseException as e:   # (must not propagate through the C frames)
                self._after_exc = e
        self._after_cb = C.CFUNCTYPE(None, C.c_void_p)(tramp)
        check(lib().kh_decoder_set_after_launch(self._h, C.cast(self._after_cb, C.c_void_p), None))

    def reached_final(self, utt=0):
        return bool(self.stats(utt)["reached_final"])

    def get_raw_lattice(self, utt=0):
        """GetRawLattice (lattice-faster-decoder.cc:109-191), canonical form."""
        st = self.stats(utt)
        n, m = st["num_tokens"], st["num_links"]
        L = dict(state_frame=np.empty(n, np.int32), state_hclg=np.empty(n, np.int32),
                 state_final=np.empty(n, np.float32), arc_src=np.empty(m, np.int32),
                 arc_dst=np.empty(m, np.int32), arc_il=np.empty(m, np.int32), arc_ol=np.empty(m, np.int32),
                 arc_g=np.empty(m, np.float32), arc_a=np.empty(m, np.float32))
        ip, fp = capi.c_int32_p, capi.c_float_p
        check(lib().kh_decoder_get_raw_lattice(
            self._h, int(utt), L["state_frame"].ctypes.data_as(ip), L["state_hclg"].ctypes.data_as(ip),
            L["state_final"].ctypes.data_as(fp), L["arc_src"].ctypes.data_as(ip), L["arc_dst"].ctypes.data_as(ip),
            L["arc_il"].ctypes.data_as(ip), L["arc_ol"].ctypes.data_as(ip), L["arc_g"].ctypes.data_as(fp),
            L["arc_a"].ctypes.data_as(fp)))
        return L

    def prepare(self, num_threads=0):
        """GetRawLattice + GetBestPath of every utterance of the batch on host
        threads (decoder-wrappers.cc:215-262 per utterance); 0 = all cores."""
        check(lib().kh_decoder_prepare(self._h, int(num_threads)))

    def get_best_path(self, utt=0):
        """GetBestPath + GetLinearSymbolSequence (decoder-wrappers.cc:232-246)."""
        cap = int(self._T[utt]) + 16
        capw = 4 * cap + 64
        ali, words = np.empty(cap, np.int32), np.empty(capw, np.int32)
        na, nw = C.c_int32(), C.c_int32()
        g, a = C.c_float(), C.c_float()
        check(lib().kh_decoder_get_best_path(self._h, int(utt), ali.ctypes.data_as(capi.c_int32_p), cap, C.byref(na),
                                             words.ctypes.data_as(capi.c_int32_p), capw, C.byref(nw),
                                             C.byref(g), C.byref(a)))
        return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(),
                    graph_cost=g.value, acoustic_cost=a.value)

    def get_best_paths(self, first=0, n=None):
        """get_best_path of utterances [first, first + n) in one call: dict of alignment / words (concatenated,
        with ali_off / words_off = n + 1 offsets), graph_cost / acoustic_cost arrays [n]."""
        n = len(self._T) - first if n is None else int(n)
        cap = int(np.sum(self._T[first:first + n])) + 16 * n + 16
        capw = 4 * cap + 64
        ali, words = np.empty(cap, np.int32), np.empty(capw, np.int32)
        ao, wo = np.empty(n + 1, np.int64), np.empty(n + 1, np.int64)
        g, a = np.empty(n, np.float32), np.empty(n, np.float32)
        i64 = C.POINTER(C.c_int64)
        check(lib().kh_decoder_get_best_paths(self._h, int(first), n, ali.ctypes.data_as(capi.c_int32_p), cap, ao.ctypes.data_as(i64),
                                              words.ctypes.data_as(capi.c_int32_p), capw, wo.ctypes.data_as(i64),
                                              g.ctypes.data_as(capi.c_float_p), a.ctypes.data_as(capi.c_float_p)))
        return dict(alignment=ali[:ao[n]], ali_off=ao, words=words[:wo[n]], words_off=wo, graph_cost=g, acoustic_cost=a)

    def stats_batch(self, first=0, n=None, counters=True, stats=True):
        """counters(u) / stats(u) of utterances [first, first + n) in one call: two structured arrays (or None)."""
        n = len(self._T) - first if n is None else int(n)
        ca = (KhDecodeStats * n)() if counters else None
        sa = (KhDecodeStats * n)() if stats else None
        check(lib().kh_decoder_get_stats_batch(self._h, int(first), n, ca, sa))
        as_np = lambda arr: None if arr is None else np.ctypeslib.as_array(arr)
        return as_np(ca), as_np(sa)


class LatticeFasterOnlineDecoder:
    """decoder/lattice-faster-online-decoder.h:44-200 for num_streams concurrent
    utterances: InitDecoding / AdvanceDecoding (a chunk of frames at a time) /
    FinalizeDecoding, GetRawLattice and GetBestPath at any point.  The scaled
    log-likelihood chunks are what DecodableNnet2Online would serve for the next
    frames (rows [t, t + n) of the utterance's matrix)."""

    def __init__(self, fst, config=None, num_streams=1, max_frames=4096, exact_reference_order=True):
        self.fst = fst
        cfg = decoder_config() if config is None else config
        self.cfg = KhDecoderConfig(**cfg)
        self.num_streams = int(num_streams)
        h = lib().kh_online_decoder_create(fst._h, C.byref(self.cfg), self.num_streams, int(max_frames))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)
        if not exact_reference_order:      # (the library's default is the reference's order)
            self.set_reference_order(False)

    def __del__(self):
        try:
            if self._h:
                lib().kh_online_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _streams(streams):
        a = np.ascontiguousarray(streams, np.int32).reshape(-1)
        return a, a.ctypes.data_as(capi.c_int32_p), len(a)

    def init_decoding(self, streams):
        a, ptr, n = self._streams(streams)
        check(lib().kh_online_decoder_init_decoding(self._h, ptr, n))

    def advance_decoding(self, streams, chunks):
        """chunks[i]: 2-D float32 device tensor, the next rows of stream streams[i]
        (all with the same row stride)."""
        a, ptr, n = self._streams(streams)
        assert len(chunks) == n
        strides = {int(_dim(c).stride) for c in chunks if c.shape[0] > 0}
        assert len(strides) <= 1, "chunks of one call must share their row stride"
        stride = strides.pop() if strides else 1
        ptrs = (C.c_void_p * n)(*[C.c_void_p(c.data_ptr()) if c.shape[0] > 0 else None for c in chunks])
        nf = np.ascontiguousarray([c.shape[0] for c in chunks], np.int32)
        t2p = _p(self.fst.tid2pdf) if self.fst.tid2pdf is not None else None
        for c in chunks:
            if c.shape[0] > 0:
                self.fst.check_pdf_map(c.shape[1])
        self._keep = chunks
        check(lib().kh_online_decoder_advance(self._h, ptr, n, ptrs, stride, nf.ctypes.data_as(capi.c_int32_p), t2p))

    def set_lazy_prune(self, enable=True):
        """The offline kernel's lazy pruning schedule (kh_online_decoder_set_lazy_prune): same final lattices and best paths,
        no pruning while the streams advance; every stream must be idle."""
        check(lib().kh_online_decoder_set_lazy_prune(self._h, int(bool(enable))))

    def set_reference_order(self, enable=True):
        """The reference's own iteration order for the streams (kh_online_decoder_set_reference_order;
        lattice-faster-online-decoder.cc:864-951): the lattices LatticeFasterOnlineDecoder itself would build; the default.
        Every stream must be idle and the persistent serving kernel stopped while switching; a serving kernel started
        afterwards decodes in the same order (ServeKernel<kExact>)."""
        check(lib().kh_online_decoder_set_reference_order(self._h, int(bool(enable))))

    def num_frames_decoded(self, stream=0):
        n = C.c_int32()
        check(lib().kh_online_decoder_num_frames_decoded(self._h, int(stream), C.byref(n)))
        return n.value

    def finalize_decoding(self, streams):
        a, ptr, n = self._streams(streams)
        check(lib().kh_online_decoder_finalize(self._h, ptr, n))

    def stats(self, stream=0, use_final_probs=True):
        st = KhDecodeStats()
        check(lib().kh_online_decoder_get_stats(self._h, int(stream), int(bool(use_final_probs)), C.byref(st)))
        return {k: getattr(st, k) for k, _ in KhDecodeStats._fields_}

    def get_raw_lattice(self, stream=0, use_final_probs=True):
        st = self.stats(stream, use_final_probs)
        n, m = st["num_tokens"], st["num_links"]
        L = dict(state_frame=np.empty(n, np.int32), state_hclg=np.empty(n, np.int32),
                 state_final=np.empty(n, np.float32), arc_src=np.empty(m, np.int32),
                 arc_dst=np.empty(m, np.int32), arc_il=np.empty(m, np.int32), arc_ol=np.empty(m, np.int32),
                 arc_g=np.empty(m, np.float32), arc_a=np.empty(m, np.float32))
        i32 = lambda k: L[k].ctypes.data_as(capi.c_int32_p)
        f32 = lambda k: L[k].ctypes.data_as(capi.c_float_p)
        check(lib().kh_online_decoder_get_raw_lattice(
            self._h, int(stream), int(bool(use_final_probs)), i32("state_frame"), i32("state_hclg"),
            f32("state_final"), i32("arc_src"), i32("arc_dst"), i32("arc_il"), i32("arc_ol"), f32("arc_g"), f32("arc_a")))
        return L

    def get_best_path(self, stream=0, use_final_probs=True):
        cap = self.num_frames_decoded(stream) + 16
        capw = 4 * cap + 64
        ali, words = np.empty(cap, np.int32), np.empty(capw, np.int32)
        na, nw = C.c_int32(), C.c_int32()
        g, a = C.c_float(), C.c_float()
        check(lib().kh_online_decoder_get_best_path(
            self._h, int(stream), int(bool(use_final_probs)), ali.ctypes.data_as(capi.c_int32_p), cap, C.byref(na),
            words.ctypes.data_as(capi.c_int32_p), capw, C.byref(nw), C.byref(g), C.byref(a)))
        return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(),
                    graph_cost=g.value, acoustic_cost=a.value)


# ---------------------------------------------------------------- lattice forward-backward
def lattice_to_csr(L):
    """Raw lattice (get_raw_lattice: states in canonical (frame, HCLG state) order, start first)
    -> top-sorted CSR, the form LatticeForwardBackward needs (the reference top-sorts lattices
    it reads: TopSortLatticeIfNeeded, lat/lattice-functions.cc, before
    nnet-compute-discriminative.cc:178).  Epsilon arcs may point backwards in the canonical
    order, so states are ordered by (frame, epsilon depth, state)."""
    n = len(L["state_frame"])
    depth = np.zeros(n, np.int64)
    eps = L["arc_il"] == 0
    es, ed = L["arc_src"][eps], L["arc_dst"][eps]
    for _ in range(n + 1):
        nd = depth.copy()
        np.maximum.at(nd, ed, depth[es] + 1)
        if np.array_equal(nd, depth):
            break
        depth = nd
    order = np.lexsort((L["state_hclg"], depth, L["state_frame"]))
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    src, dst = rank[L["arc_src"]], rank[L["arc_dst"]]
    perm = np.lexsort((np.arange(len(src)), src))
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(src, minlength=n))
    return dict(n_states=n, arc_offsets=off, arc_ilabel=L["arc_il"][perm].astype(np.int32),
                arc_nextstate=dst[perm].astype(np.int32), arc_graph=L["arc_g"][perm].astype(np.float32),
                arc_acoustic=L["arc_a"][perm].astype(np.float32),
                state_final=L["state_final"][order].astype(np.float32), perm=perm, order=order)


def lattice_forward_backward(lats):
    """LatticeForwardBackward (lat/lattice-functions.cc:272-354) for a batch of
    top-sorted lattices.  Each lattice is a dict with n_states, arc_offsets (int64,
    n_states+1), arc_ilabel, arc_nextstate, arc_graph, arc_acoustic, state_final.
    Returns per lattice: dict(arc_post, tot_like, acoustic_like_sum, state_times, post)
    where post is the Posterior: per frame a sorted list of (transition-id, weight)
    merged as MergePairVectorSumming does (:351-352)."""
    n = len(lats)
    soff = np.zeros(n + 1, np.int32)
    for i, L in enumerate(lats):
        soff[i + 1] = soff[i] + L["n_states"]
    aoff = [np.zeros(1, np.int64)]
    base = 0
    for L in lats:
        o = np.asarray(L["arc_offsets"], np.int64)
        aoff.append(o[1:] + base)
        base += int(o[-1])
    aoff = np.ascontiguousarray(np.concatenate(aoff))
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(L[k], dt) for L in lats]))
    il, ns = cat("arc_ilabel", np.int32), cat("arc_nextstate", np.int32)
    g, a, fin = cat("arc_graph", np.float32), cat("arc_acoustic", np.float32), cat("state_final", np.float32)
    post = np.empty(len(il), np.float32)
    tot, ac = np.empty(n, np.float64), np.empty(n, np.float64)
    times = np.empty(int(soff[-1]), np.int32)
    ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p
    check(lib().kh_lattice_forward_backward(
        n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip),
        ns.ctypes.data_as(ip), g.ctypes.data_as(fp), a.ctypes.data_as(fp), fin.ctypes.data_as(fp),
        post.ctypes.data_as(fp), tot.ctypes.data_as(dp), ac.ctypes.data_as(dp), times.ctypes.data_as(ip)))
    out = []
    for i, L in enumerate(lats):
        a0, a1 = int(aoff[soff[i]]), int(aoff[soff[i + 1]])
        t = times[soff[i]:soff[i + 1]]
        ap = post[a0:a1]
        max_time = int(t.max()) if len(t) else 0
        posterior = _arc_posterior_to_post(L, t, il[a0:a1], ap, max_time)
        out.append(dict(arc_post=ap.copy(), tot_like=float(tot[i]), acoustic_like_sum=float(ac[i]),
                        state_times=t.copy(), post=posterior))
    return out


def _cat_lattices(lats):
    n = len(lats)
    soff = np.zeros(n + 1, np.int32)
    for i, L in enumerate(lats):
        soff[i + 1] = soff[i] + L["n_states"]
    aoff = [np.zeros(1, np.int64)]
    base = 0
    for L in lats:
        o = np.asarray(L["arc_offsets"], np.int64)
        aoff.append(o[1:] + base)
        base += int(o[-1])
    aoff = np.ascontiguousarray(np.concatenate(aoff))
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(L[k], dt) for L in lats]))
    return (n, soff, aoff, cat("arc_ilabel", np.int32), cat("arc_nextstate", np.int32), cat("arc_graph", np.float32),
            cat("arc_acoustic", np.float32), cat("state_final", np.float32))


def _arc_posterior_to_post(L, times, tids, arc_post, max_time):
    """(*post)[state_times[s]].push_back((tid, p)) for tid != 0, then MergePairVectorSumming
    (util/stl-utils.h: sort by tid, sum equal keys in float in their original order, drop
    zeros).  Vectorised: stable sort by (frame, tid), sequential float32 sums (np.add.at)."""
    src = np.repeat(np.arange(L["n_states"]), np.diff(np.asarray(L["arc_offsets"], np.int64)))
    m = tids != 0
    fr, ti, p = times[src[m]].astype(np.int64), tids[m].astype(np.int64), arc_post[m].astype(np.float32)
    key = fr * (int(ti.max()) + 1 if len(ti) else 1) + ti
    order = np.argsort(key, kind="stable")
    key, fr, ti, p = key[order], fr[order], ti[order], p[order]
    first = np.ones(len(key), bool)
    first[1:] = key[1:] != key[:-1]
    grp = np.cumsum(first) - 1
    acc = np.zeros(int(grp[-1]) + 1 if len(grp) else 0, np.float32)
    np.add.at(acc, grp, p)                       # unbuffered: adds in array order
    gfr, gti = fr[first], ti[first]
    keep = acc != 0.0
    gfr, gti, acc = gfr[keep], gti[keep], acc[keep]
    bounds = np.searchsorted(gfr, np.arange(max_time + 1))
    tl, wl = gti.tolist(), acc.tolist()
    return [list(zip(tl[bounds[t]:bounds[t + 1]], wl[bounds[t]:bounds[t + 1]])) for t in range(max_time)]


def lattice_alphas_betas(lats, viterbi=False):
    """ComputeLatticeAlphasAndBetas (lat/lattice-functions.cc:412-463) for a batch."""
    n, soff, aoff, il, ns, g, a, fin = _cat_lattices(lats)
    alpha, beta, tot = np.empty(int(soff[-1])), np.empty(int(soff[-1])), np.empty(n)
    ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p
    check(lib().kh_lattice_alphas_betas(
        n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
        g.ctypes.data_as(fp), a.ctypes.data_as(fp), fin.ctypes.data_as(fp), int(bool(viterbi)),
        alpha.ctypes.data_as(dp), beta.ctypes.data_as(dp), tot.ctypes.data_as(dp)))
    return [dict(alpha=alpha[soff[i]:soff[i + 1]].copy(), beta=beta[soff[i]:soff[i + 1]].copy(), tot=float(tot[i]))
            for i in range(n)]


def lattice_forward_backward_mpe(lats, tid2phone, tid2pdf, silence_phones, num_alis, criterion="smbr",
                                 one_silence_class=False):
    """LatticeForwardBackwardMpeVariants (lat/lattice-functions.cc:740-919) for a batch;
    returns per lattice dict(arc_post, post, tot_forward_score)."""
    if criterion not in ("smbr", "mpfe"):
        raise KhError('criterion must be "mpfe" or "smbr" (lattice-functions.cc:753)')
    n, soff, aoff, il, ns, g, a, fin = _cat_lattices(lats)
    t2ph, t2pdf = np.ascontiguousarray(tid2phone, np.int32), np.ascontiguousarray(tid2pdf, np.int32)
    sil = np.ascontiguousarray(sorted(silence_phones), np.int32)
    ali_off = np.concatenate([[0], np.cumsum([len(x) for x in num_alis])]).astype(np.int32)
    ali = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in num_alis]) if len(num_alis) else [], np.int32)
    post = np.empty(len(il), np.float32)
    score = np.empty(n)
    all_times = np.empty(int(soff[-1]), np.int32)
    ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p
    check(lib().kh_lattice_forward_backward_mpe(
        n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
        g.ctypes.data_as(fp), a.ctypes.data_as(fp), fin.ctypes.data_as(fp), t2ph.ctypes.data_as(ip),
        t2pdf.ctypes.data_as(ip), len(t2ph) - 1, sil.ctypes.data_as(ip), len(sil), ali.ctypes.data_as(ip),
        ali_off.ctypes.data_as(ip), int(criterion == "mpfe"), int(bool(one_silence_class)), post.ctypes.data_as(fp),
        score.ctypes.data_as(dp), all_times.ctypes.data_as(ip)))
    out = []
    for i, L in enumerate(lats):
        a0, a1 = int(aoff[soff[i]]), int(aoff[soff[i + 1]])
        times = all_times[soff[i]:soff[i + 1]]
        out.append(dict(arc_post=post[a0:a1].copy(), tot_forward_score=float(score[i]),
                        post=_arc_posterior_to_post(L, times, il[a0:a1], post[a0:a1], len(num_alis[i]))))
    return out


def lattice_state_times(L):
    """LatticeStateTimes (lat/lattice-functions.cc:36-67) of one top-sorted lattice."""
    n = L["n_states"]
    off = np.asarray(L["arc_offsets"], np.int64)
    il, ns = np.asarray(L["arc_ilabel"]), np.asarray(L["arc_nextstate"])
    times = np.full(n, -1, np.int32)
    times[0] = 0
    for s in range(n):
        for a in range(off[s], off[s + 1]):
            want = times[s] + (1 if il[a] != 0 else 0)
            if times[ns[a]] == -1:
                times[ns[a]] = want
            elif times[ns[a]] != want:
                raise KhError("LatticeStateTimes: inconsistent lattice")
    return times


class LatticeBatch:
    """A batch of top-sorted lattices kept on the device (kh_lattice_batch_*): uploaded and prepared once; forward-backward,
    rescoring with a new score matrix and the forward-backward after it run from there (what
    NnetDiscriminativeUpdater::LatticeComputations does to one lattice, nnet-compute-discriminative.cc:178-321)."""

    def __init__(self, lats):
        self.n, self.soff, self.aoff, il, ns, g, a, fin = _cat_lattices(lats) if not isinstance(lats, tuple) else lats
        self._il = il
        ip, fp = capi.c_int32_p, capi.c_float_p
        h = lib().kh_lattice_batch_create(self.n, self.soff.ctypes.data_as(ip), self.aoff.ctypes.data_as(capi.c_int64_p),
                                          il.ctypes.data_as(ip), ns.ctypes.data_as(ip), g.ctypes.data_as(fp), a.ctypes.data_as(fp),
                                          fin.ctypes.data_as(fp))
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)
        self.total_arcs, self.total_states = len(il), int(self.soff[-1])

    def __del__(self):
        try:
            if self._h:
                lib().kh_lattice_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def forward_backward(self, want_times=False):
        """LatticeForwardBackward of every lattice: dict(arc_post [total_arcs], tot_like [n], acoustic_like_sum [n], state_times)."""
        post = np.empty(self.total_arcs, np.float32)
        tot, ac = np.empty(self.n, np.float64), np.empty(self.n, np.float64)
        times = np.empty(self.total_states, np.int32) if want_times else None
        check(lib().kh_lattice_batch_forward_backward(self._h, post.ctypes.data_as(capi.c_float_p), tot.ctypes.data_as(capi.c_double_p),
                                                      ac.ctypes.data_as(capi.c_double_p),
                                                      times.ctypes.data_as(capi.c_int32_p) if want_times else None))
        return dict(arc_post=post, tot_like=tot, acoustic_like_sum=ac, state_times=times)

    def forward_backward_device(self, out=None):
        """The same with the arc posteriors left on the device: dict(arc_post = torch tensor [total_arcs] on the GPU, tot_like,
        acoustic_like_sum)."""
        import torch
        if out is None:
            out = torch.empty(self.total_arcs, dtype=torch.float32, device="cuda")
        tot, ac = np.empty(self.n, np.float64), np.empty(self.n, np.float64)
        check(lib().kh_lattice_batch_forward_backward_dev(self._h, _p(out), tot.ctypes.data_as(capi.c_double_p),
                                                          ac.ctypes.data_as(capi.c_double_p)))
        return dict(arc_post=out, tot_like=tot, acoustic_like_sum=ac)

    def rescore(self, loglikes, utt_row_offsets, tid2pdf=None, fetch=False):
        """RescoreLattice with a device score matrix; the acoustic costs change on the device (fetch: return them)."""
        off = np.ascontiguousarray(utt_row_offsets, np.int32)
        out = np.empty(self.total_arcs, np.float32) if fetch else None
        check(lib().kh_lattice_batch_rescore(self._h, _p(loglikes), _dim(loglikes).stride, off.ctypes.data_as(capi.c_int32_p),
                                             _p(tid2pdf) if tid2pdf is not None else None,
                                             out.ctypes.data_as(capi.c_float_p) if fetch else None))
        return out


def lattice_last_timings():
    """Milliseconds the last lattice call of this thread spent in upload / device preparation / sweeps / download."""
    ms = (C.c_float * 4)()
    check(lib().kh_lattice_last_timings(ms))
    return dict(upload_ms=ms[0], prep_ms=ms[1], sweeps_ms=ms[2], download_ms=ms[3])


def rescore_lattice(lats, loglikes, utt_row_offsets, tid2pdf=None):
    """RescoreLattice (lat/lattice-functions.cc:1307-1358) for a batch: loglikes = device
    matrix (rows of lattice i at utt_row_offsets[i]...).  Returns the new arc_acoustic arrays."""
    n, soff, aoff, il, ns, g, a, fin = _cat_lattices(lats)
    a = a.copy()
    off = np.ascontiguousarray(utt_row_offsets, np.int32)
    ip, fp = capi.c_int32_p, capi.c_float_p
    check(lib().kh_rescore_lattice(n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip),
                                   ns.ctypes.data_as(ip), a.ctypes.data_as(fp), _p(loglikes), _dim(loglikes).stride,
                                   off.ctypes.data_as(ip), _p(tid2pdf) if tid2pdf is not None else None))
    return [a[int(aoff[soff[i]]):int(aoff[soff[i + 1]])].copy() for i in range(n)]


def comp_objf_and_deriv(deriv, sv_labels, output):
    """CuMatrix::CompObjfAndDeriv (cu-matrix.cc:1198-1248): deriv.CompObjfAndDeriv(sv_labels,
    output, &tot_objf, &tot_weight); sv_labels = [(row, column, weight), ...]."""
    r = np.ascontiguousarray([x[0] for x in sv_labels], np.int32)
    c = np.ascontiguousarray([x[1] for x in sv_labels], np.int32)
    w = np.ascontiguousarray([x[2] for x in sv_labels], np.float32)
    objf, wt = C.c_float(), C.c_float()
    check(lib().kh_comp_objf_and_deriv(len(r), r.ctypes.data_as(capi.c_int32_p), c.ctypes.data_as(capi.c_int32_p),
                                       w.ctypes.data_as(capi.c_float_p), _p(output), _dim(output), _p(deriv), _dim(deriv),
                                       C.byref(objf), C.byref(wt)))
    return objf.value, wt.value


# ---- Posterior algebra of hmm/posterior.cc used by LatticeForwardBackwardMmi (host lists,
# as in the reference).  A Posterior is a list over frames of [(id, weight), ...].
def alignment_to_posterior(ali):
    """AlignmentToPosterior hmm/posterior.cc:276-285."""
    return [[(int(t), 1.0)] for t in ali]


def scale_posterior(scale, post):
    """ScalePosterior hmm/posterior.cc:204-218 (scale == 0 clears the frames)."""
    if scale == 0.0:
        return [[] for _ in post]
    return [[(i, float(np.float32(np.float32(w) * np.float32(scale)))) for i, w in fr] for fr in post]


def convert_posterior_to_pdfs(tid2pdf, post):
    """ConvertPosteriorToPdfs hmm/posterior.cc:308-332 (entries of a frame in ascending pdf
    order here; the reference's order is that of an unordered_map)."""
    out = []
    for fr in post:
        acc = {}
        for tid, w in fr:
            pdf = int(tid2pdf[tid])
            acc[pdf] = np.float32(acc[pdf] + np.float32(w)) if pdf in acc else np.float32(w)
        out.append([(k, float(v)) for k, v in sorted(acc.items()) if v != 0.0])
    return out


def merge_posteriors(post1, post2, merge, drop_frames):
    """MergePosteriors hmm/posterior.cc:244-274; returns (post, num_disjoint)."""
    assert len(post1) == len(post2)
    out, num_disjoint = [], 0
    for a, b in zip(post1, post2):
        fr = list(a) + list(b)
        if merge:  # MergePairVectorSumming: sort on the id, sum equal ids, drop zeros
            fr.sort(key=lambda x: x[0])
            merged = []
            for i, w in fr:
                if merged and merged[-1][0] == i:
                    merged[-1] = (i, float(np.float32(np.float32(merged[-1][1]) + np.float32(w))))
                else:
                    merged.append((i, float(np.float32(w))))
            fr = [(i, w) for i, w in merged if w != 0.0]
        else:
            fr.sort()
        if not ({i for i, _ in a} & {i for i, _ in b}):  # PosteriorEntriesAreDisjoint :220-241
            num_disjoint += 1
            if drop_frames:
                fr = []
        out.append(fr)
    return out, num_disjoint


def lattice_forward_backward_mmi(lats, tid2pdf, num_alis, drop_frames, convert_to_pdf_ids, cancel):
    """LatticeForwardBackwardMmi (lat/lattice-functions.cc:1361-1396) for a batch: the
    denominator forward-backward on the GPU, the posterior algebra on the host as in the
    reference.  Returns per lattice dict(post, tot_like)."""
    fb = lattice_forward_backward(lats)
    out = []
    for r, ali in zip(fb, num_alis):
        den_post = scale_posterior(-1.0, r["post"])
        num_post = alignment_to_posterior(ali)
        if convert_to_pdf_ids:
            num_post = convert_posterior_to_pdfs(tid2pdf, num_post)
            den_post = convert_posterior_to_pdfs(tid2pdf, den_post)
        post, _ = merge_posteriors(num_post, den_post, cancel, drop_frames)
        out.append(dict(post=post, tot_like=r["tot_like"]))
    return out


# ---------------------------------------------------------------- config 5: discriminative pass
def _lookup_values(M, rows, cols):
    """CuMatrix::Lookup (cu-matrix.cc:2327) for index arrays."""
    pairs = np.empty((len(rows), 2), np.int32)
    pairs[:, 0] = rows
    pairs[:, 1] = cols
    out = lookup(M, pairs)
    synchronize()
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _merge_keyed(rows, cols, weights, n_rows=None):
    """Sum float32 weights per (row, col), drop exact zeros; keys sorted (MergePairVectorSumming
    per frame, util/stl-utils.h:303-322, for all frames at once: kh_merge_pair_vector_summing)."""
    n = len(rows)
    if n == 0:
        return rows.astype(np.int64), cols.astype(np.int64), weights.astype(np.float32)
    r = np.ascontiguousarray(rows, np.int32)
    c = np.ascontiguousarray(cols, np.int32)
    w = np.ascontiguousarray(weights, np.float32)
    if n_rows is None:
        n_rows = int(r.max()) + 1
    orow, ocol, ow = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
    n_out = C.c_int64()
    ip, fp = capi.c_int32_p, capi.c_float_p
    check(lib().kh_merge_pair_vector_summing(n, r.ctypes.data_as(ip), c.ctypes.data_as(ip), w.ctypes.data_as(fp), int(n_rows),
                                             orow.ctypes.data_as(ip), ocol.ctypes.data_as(ip), ow.ctypes.data_as(fp),
                                             C.byref(n_out)))
    m = n_out.value
    return orow[:m].astype(np.int64), ocol[:m].astype(np.int64), ow[:m]


class DiscriminativeCall:
    """A discriminative_lattice_computations(..., begin=True) in flight: its device work is queued on a stream of its own
    (kh_discriminative_lattice_computations_begin); end() waits for it and returns what the plain call returns."""

    def __init__(self, handle, out, deriv, Ts, weights):
        self._h, self.output, self.deriv, self._Ts, self._w = handle, out, deriv, Ts, weights

    def end(self):
        if self._h is None:
            raise KhError("DiscriminativeCall.end() called twice")
        st = np.zeros(5)
        h, self._h = self._h, None
        check(lib().kh_discriminative_lattice_computations_end(h, st.ctypes.data_as(capi.c_double_p)))
        stats = dict(tot_t=float(self._Ts.sum()), tot_t_weighted=float((self._Ts * self._w).sum()), tot_num_count=float(st[0]),
                     tot_num_objf=float(st[1]), tot_den_objf=float(st[2]))
        return dict(stats=stats, deriv=self.deriv, output=self.output, objf=float(st[3]), weight=float(st[4]))

    def __del__(self):
        try:
            if self._h is not None:
                lib().kh_discriminative_lattice_computations_end(self._h, np.zeros(5).ctypes.data_as(capi.c_double_p))
                self._h = None
        except Exception:
            pass


def discriminative_lattice_computations(nnet, priors, tid2pdf, egs, criterion="smbr", acoustic_scale=0.1,
                                        drop_frames=False, one_silence_class=False, tid2phone=None,
                                        silence_phones=(), den_lats=None, begin=False):
    """NnetDiscriminativeUpdater::Propagate + LatticeComputations
    (nnet2/nnet-compute-discriminative.cc:150-321) for a BATCH of examples, as one pipeline
    on the device: network forward -> CuMatrix::Lookup of the posteriors the numerator
    alignment and the denominator lattice arcs need (:196-226) -> scaled pseudo log-likelihoods
    log(post / prior) x acoustic_scale with the 1e-20 floor (:231-247) written into the lattice
    (:259-277) -> MMI / sMBR / MPFE forward-backward (GetDiscriminativePosteriors :324-343) ->
    the Posterior algebra of hmm/posterior.cc + ScalePosterior(weight) -> CompObjfAndDeriv (:279-316);
    everything behind the forward pass is ONE library call, kh_discriminative_lattice_computations
    (the host hands over the lattices and the alignments, nothing comes back but five numbers).
    (--boost is not supported.)

    egs: list of dict(feats = device [T + left + right, D] with exactly the network's context,
    num_ali = int32 [T], den_lat = top-sorted CSR lattice (lattice_to_csr), weight = float).
    den_lats: optionally the examples' lattices already concatenated (cat_lattices(...)), e.g. by the
    data loader's worker - the egs' den_lat entries are then not read.
    Returns dict(stats = NnetDiscriminativeStats fields, deriv = device matrix [sum T, num_pdfs]:
    the derivative at the network output, output = the posteriors).
    begin=True: returns a DiscriminativeCall as soon as the work is queued (the lattice steps on a stream of the call's
    own); launch the NEXT batch's call - its forward pass runs beside this batch's sweeps - then .end() this one."""
    if criterion not in ("mmi", "smbr", "mpfe"):
        raise KhError('criterion must be "mmi", "mpfe" or "smbr"')
    n = len(egs)
    timing = os.environ.get("KH_LATTICE_TIMING") is not None   # host-side split of the call on stderr (tools/time_cfg5.py)
    t_in = time.perf_counter()
    t2p = np.ascontiguousarray(tid2pdf, np.int32)
    pri = np.ascontiguousarray(priors, np.float32)
    Ts = np.array([len(e["num_ali"]) for e in egs], np.int64)
    row_off = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int32)
    # ---- Propagate (:150-175): the examples carry exactly the context the network needs
    feat_rows = np.array([e["feats"].shape[0] for e in egs], np.int64)
    foff = np.concatenate([[0], np.cumsum(feat_rows)]).astype(np.int32)
    feats = torch.cat([e["feats"] for e in egs], 0) if n > 1 else egs[0]["feats"]
    out, out_off = nnet.compute(feats, foff, pad_input=False, wait=False)   # (the lattice call below ends with a wait for the stream)
    t_fwd = time.perf_counter()
    if not np.array_equal(np.diff(np.asarray(out_off)), Ts):
        raise KhError("KALDI_ASSERT(posteriors.NumRows() == num_frames) nnet-compute-discriminative.cc:194")
    if out.shape[1] != len(pri):
        raise KhError("KALDI_ASSERT(num_pdfs == priors.Dim()) :196")
    if criterion != "mmi" and tid2phone is None:
        raise KhError("sMBR / MPFE need the transition-id -> phone map")
    ali = np.ascontiguousarray(np.concatenate([np.asarray(e["num_ali"], np.int32) for e in egs]), np.int32)
    weights = np.array([float(e.get("weight", 1.0)) for e in egs], np.float32)
    t2ph = np.ascontiguousarray(tid2phone, np.int32) if tid2phone is not None else None
    sil = np.ascontiguousarray(sorted(silence_phones), np.int32)
    deriv = torch.empty_like(out)
    st = np.zeros(5)
    ip, fp = capi.c_int32_p, capi.c_float_p
    tail = (ali.ctypes.data_as(ip), row_off.ctypes.data_as(ip),
            weights.ctypes.data_as(fp), t2p.ctypes.data_as(ip), t2ph.ctypes.data_as(ip) if t2ph is not None else None, len(t2p) - 1,
            sil.ctypes.data_as(ip), len(sil), {"mmi": 0, "smbr": 1, "mpfe": 2}[criterion], float(acoustic_scale), int(bool(drop_frames)),
            int(bool(one_silence_class)), pri.ctypes.data_as(fp), _p(out), _dim(out), _p(deriv), _dim(deriv),
            st.ctypes.data_as(capi.c_double_p))
    if den_lats is None:
        # the lattices as the examples hold them: the library assembles the batch (host threads, pinned memory) and prepares
        # it on the device beside the forward pass launched above, which nobody has waited for yet
        keep, cols = [], []
        for key, dt in (("arc_offsets", np.int64), ("arc_ilabel", np.int32), ("arc_nextstate", np.int32), ("arc_graph", np.float32),
                        ("arc_acoustic", np.float32), ("state_final", np.float32)):
            arrs = [np.ascontiguousarray(e["den_lat"][key], dt) for e in egs]
            keep.append(arrs)
            cols.append(np.fromiter((x.__array_interface__["data"][0] for x in arrs), np.uint64, n))
        nst = np.fromiter((e["den_lat"]["n_states"] for e in egs), np.int32, n)
        lens = [np.fromiter((x.size if x.ndim == 1 else -1 for x in arrs), np.int64, n) for arrs in keep]
        na = np.fromiter((int(x[-1]) if x.size else -1 for x in keep[0]), np.int64, n)
        good = (lens[0] == nst + 1) & (lens[5] == nst)
        for k in range(1, 5):
            good &= lens[k] == na
        if not good.all():
            raise KhError("example %d: the arrays of den_lat do not have the lengths n_states / arc_offsets give" % int(np.argmin(good)))
        vp = C.c_void_p
        if timing:
            sys.stderr.write("[api timing] forward launched after %.2f ms, lattice arrays listed after %.2f ms\n"
                             % ((t_fwd - t_in) * 1e3, (time.perf_counter() - t_in) * 1e3))
        if begin:
            h = vp()
            check(lib().kh_discriminative_lattice_computations_begin(
                n, nst.ctypes.data_as(ip), *[c.ctypes.data_as(vp) for c in cols], *tail[:-1], C.byref(h)))
            return DiscriminativeCall(h, out, deriv, Ts, weights)   # (the host arrays have been copied or uploaded)
        check(lib().kh_discriminative_lattice_computations_parts(
            n, nst.ctypes.data_as(ip), *[c.ctypes.data_as(vp) for c in cols], *tail))
        del keep
    else:
        if begin:
            raise KhError("begin=True takes the lattices by example (den_lats=None)")
        nl, soff, aoff, il, ns, g, a, fin = den_lats
        if nl != n:
            raise KhError("one denominator lattice per example")
        check(lib().kh_discriminative_lattice_computations(
            nl, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
            g.ctypes.data_as(fp), a.ctypes.data_as(fp), fin.ctypes.data_as(fp), *tail))
    stats = dict(tot_t=float(Ts.sum()), tot_t_weighted=float((Ts * weights).sum()), tot_num_count=float(st[0]),
                 tot_num_objf=float(st[1]), tot_den_objf=float(st[2]))
    return dict(stats=stats, deriv=deriv, output=out, objf=float(st[3]), weight=float(st[4]))


def cat_lattices(lats):
    """The batch form of top-sorted CSR lattices the library's lattice calls take (what a data loader can prepare ahead)."""
    return _cat_lattices(lats)


def lattice_forward_backward_mpe_raw(lats, tid2phone, tid2pdf, silence_phones, num_alis, criterion, one_silence_class):
    """kh_lattice_forward_backward_mpe for a batch, arrays only (no per-frame Python lists)."""
    n, soff, aoff, il, ns, g, a, fin = _cat_lattices(lats)
    t2ph, t2pdf = np.ascontiguousarray(tid2phone, np.int32), np.ascontiguousarray(tid2pdf, np.int32)
    sil = np.ascontiguousarray(sorted(silence_phones), np.int32)
    ali_off = np.concatenate([[0], np.cumsum([len(x) for x in num_alis])]).astype(np.int32)
    ali = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in num_alis]), np.int32)
    post = np.empty(len(il), np.float32)
    score = np.empty(n)
    ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p
    check(lib().kh_lattice_forward_backward_mpe(
        n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
        g.ctypes.data_as(fp), a.ctypes.data_as(fp), fin.ctypes.data_as(fp), t2ph.ctypes.data_as(ip),
        t2pdf.ctypes.data_as(ip), len(t2ph) - 1, sil.ctypes.data_as(ip), len(sil), ali.ctypes.data_as(ip),
        ali_off.ctypes.data_as(ip), int(criterion == "mpfe"), int(bool(one_silence_class)), post.ctypes.data_as(fp),
        score.ctypes.data_as(dp), None))
    return dict(arc_post=post, tot_forward_score=score)


# ---------------------------------------------------------------- lattice determinization
def tid_phone_map(tm):
    """What DeterminizeLatticeInsertPhones asks the TransitionModel per transition-id (determinize-lattice-pruned.cc:1335-1338):
    tid_phone[tid] = TransitionIdToPhone(tid) when TransitionIdToHmmState(tid) == 0 and not IsSelfLoop(tid), else 0.
    `tm`: the dict kaldi_io.read_transition_model returns (tid2phone, tid2hmm_state, tid_is_self_loop)."""
    ph = np.asarray(tm["tid2phone"], np.int32)
    keep = (np.asarray(tm["tid2hmm_state"]) == 0) & ~np.asarray(tm["tid_is_self_loop"], bool)
    out = np.where(keep, ph, 0).astype(np.int32)
    out[0] = 0
    return out


def determinize_lattice_pruned(L, beam, delta=2.0 ** -10, max_mem=50000000, tid_phone=None, phone_determinize=None,
                               word_determinize=True, minimize=False):
    """DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1497-1519; call
    site decoder/decoder-wrappers.cc:264-274) on a raw lattice (get_raw_lattice layout).
    Returns the CompactLattice as a dict: n_states (state 0 = start), arcs sorted by source
    (arc_src, arc_dst, arc_label = word, arc_g, arc_a, arc_string = list of int32 arrays of
    transition-ids), final_g / final_a (+inf for non-final states), final_string, complete
    (False: "Determinization finished earlier than the beam").  Host code, no GPU involved."""
    ip, fp = capi.c_int32_p, capi.c_float_p
    src = np.ascontiguousarray(L["arc_src"], np.int32)
    dst = np.ascontiguousarray(L["arc_dst"], np.int32)
    il = np.ascontiguousarray(L["arc_il"], np.int32)
    ol = np.ascontiguousarray(L["arc_ol"], np.int32)
    g = np.ascontiguousarray(L["arc_g"], np.float32)
    a = np.ascontiguousarray(L["arc_a"], np.float32)
    fin = np.ascontiguousarray(L["state_final"], np.float32)
    if phone_determinize is None:       # the reference's default (true) wherever the transition model's map is given
        phone_determinize = tid_phone is not None
    tp = np.ascontiguousarray(tid_phone, np.int32) if tid_phone is not None else None
    if phone_determinize and tp is None:
        raise KhError("phone_determinize needs tid_phone (api.tid_phone_map of the transition model)")
    h = lib().kh_determinize_lattice_phone_pruned(len(fin), len(src), src.ctypes.data_as(ip), dst.ctypes.data_as(ip),
                                                  il.ctypes.data_as(ip), ol.ctypes.data_as(ip), g.ctypes.data_as(fp),
                                                  a.ctypes.data_as(fp), fin.ctypes.data_as(fp),
                                                  tp.ctypes.data_as(ip) if tp is not None else None, len(tp) if tp is not None else 0,
                                                  float(beam), float(delta), int(max_mem), int(bool(phone_determinize)),
                                                  int(bool(word_determinize)), int(bool(minimize)))
    if not h:
        raise KhError(lib().kh_last_error().decode())
    h = C.c_void_p(h)
    try:
        return _read_compact_lattice(h)
    finally:
        lib().kh_compact_lattice_free(h)


def _read_compact_lattice(h):
    """KhCompactLattice handle -> the dict layout of determinize_lattice_pruned."""
    ip, fp = capi.c_int32_p, capi.c_float_p
    if True:
        n, m, ns_, nf_, comp = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().kh_compact_lattice_sizes(h, C.byref(n), C.byref(m), C.byref(ns_), C.byref(nf_), C.byref(comp)))
        n, m = n.value, m.value
        out = dict(n_states=n, arc_src=np.empty(m, np.int32), arc_dst=np.empty(m, np.int32), arc_label=np.empty(m, np.int32),
                   arc_g=np.empty(m, np.float32), arc_a=np.empty(m, np.float32), final_g=np.empty(n, np.float32),
                   final_a=np.empty(n, np.float32), complete=bool(comp.value))
        aso, fso = np.zeros(m + 1, np.int32), np.zeros(n + 1, np.int32)
        astr, fstr = np.empty(ns_.value, np.int32), np.empty(nf_.value, np.int32)
        check(lib().kh_compact_lattice_get(h, out["arc_src"].ctypes.data_as(ip), out["arc_dst"].ctypes.data_as(ip),
                                           out["arc_label"].ctypes.data_as(ip), out["arc_g"].ctypes.data_as(fp),
                                           out["arc_a"].ctypes.data_as(fp), aso.ctypes.data_as(ip), astr.ctypes.data_as(ip),
                                           out["final_g"].ctypes.data_as(fp), out["final_a"].ctypes.data_as(fp),
                                           fso.ctypes.data_as(ip), fstr.ctypes.data_as(ip)))
        out["arc_string"] = [astr[aso[j]:aso[j + 1]].copy() for j in range(m)]
        out["final_string"] = [fstr[fso[s]:fso[s + 1]].copy() for s in range(n)]
        return out


def host_cpus():
    """CPUs this process may use: the affinity mask, capped by the cgroup's CPU quota (cpu.max; a container that sees 256
    cores and is granted 16 runs 16 threads well and 256 badly - beyond the quota the kernel throttles the whole group)."""
    import os as _os
    n = len(_os.sched_getaffinity(0)) if hasattr(_os, "sched_getaffinity") else (_os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def determinize_lattices(lats, beam, delta=2.0 ** -10, max_mem=50000000, num_threads=0, **kw):
    """determinize_lattice_pruned for a batch on host threads (the library call releases the
    GIL; utterances are independent, as the reference's $nj jobs / TaskSequencer threads)."""
    import concurrent.futures
    nt = num_threads if num_threads > 0 else min(len(lats), host_cpus())
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, nt)) as ex:
        return list(ex.map(lambda L: determinize_lattice_pruned(L, beam, delta, max_mem, **kw), lats))


# ---------------------------------------------------------------- iVector extraction (f3)
class OnlineIvectorExtractor:
    """OnlineIvectorExtractionInfo + OnlineIvectorFeature (online2/online-ivector-feature.h:51-134,
    :226-330) in the deterministic mode (no silence weighting, use_most_recent_ivector = false, a
    fresh adaptation state per utterance), batched over utterances.  `info`: lda_mat
    [feat_dim x (spliced dim (+ 1))], global_cmvn_stats [2 x (base_dim + 1)], splice_left/right,
    cmn_window / speaker_frames / global_frames / normalize_mean / normalize_variance
    (OnlineCmvnOptions), the diagonal UBM (ubm_weights, ubm_means, ubm_vars), the IvectorExtractor
    (M [I x D x S], Sigma_inv [I x D x D], prior_offset) and ivector_period, num_gselect, min_post,
    posterior_scale, max_count, num_cg_iters (online-ivector-feature.h:102-107)."""

    def __init__(self, info):
        lda = np.ascontiguousarray(info["lda_mat"], np.float32)
        gs = np.ascontiguousarray(info["global_cmvn_stats"], np.float64)
        M = np.ascontiguousarray(info["M"], np.float64)
        si = np.ascontiguousarray(info["Sigma_inv"], np.float64)
        inv = (1.0 / np.asarray(info["ubm_vars"], np.float64)).astype(np.float32)
        mi = (np.asarray(info["ubm_means"], np.float64) / np.asarray(info["ubm_vars"], np.float64)).astype(np.float32)
        g, bad = gmm_compute_gconsts(info["ubm_weights"], mi, inv)
        if bad:
            raise KhError("DiagGmm::ComputeGconsts: %d bad gconsts in the UBM" % bad)
        cfg = capi.KhIvectorConfig()
        cfg.base_dim = gs.shape[1] - 1
        cfg.splice_left, cfg.splice_right = int(info["splice_left"]), int(info["splice_right"])
        cfg.feat_dim, cfg.lda_cols = lda.shape
        cfg.num_gauss, _, cfg.ivector_dim = M.shape
        if M.shape[1] != cfg.feat_dim or si.shape != (cfg.num_gauss, cfg.feat_dim, cfg.feat_dim) or mi.shape != (cfg.num_gauss, cfg.feat_dim):
            raise KhError("OnlineIvectorExtractionInfo: model dimensions do not match")
        for k in ("cmn_window", "speaker_frames", "global_frames", "normalize_mean", "normalize_variance",
                  "ivector_period", "num_gselect", "num_cg_iters"):
            setattr(cfg, k, int(info[k]))
        for k in ("min_post", "posterior_scale", "max_count", "prior_offset"):
            setattr(cfg, k, float(info[k]))
        # use_most_recent_ivector + greedy_ivector_extractor (--online=false): one estimate per utterance
        cfg.greedy_most_recent = int(bool(info.get("greedy_most_recent", False)))
        self.cfg = cfg
        fp, dp = capi.c_float_p, capi.c_double_p
        self._h = lib().kh_ivector_extractor_create(C.byref(cfg), lda.ctypes.data_as(fp), gs.ctypes.data_as(dp),
                                                    g.ctypes.data_as(fp), mi.ctypes.data_as(fp), inv.ctypes.data_as(fp),
                                                    M.ctypes.data_as(dp), si.ctypes.data_as(dp))
        if not self._h:
            raise KhError(lib().kh_last_error().decode())

    @property
    def ivector_dim(self):
        return self.cfg.ivector_dim

    def state_dim(self):
        return int(lib().kh_ivector_state_dim(self._h))

    def fresh_state(self, n):
        """OnlineIvectorExtractorAdaptationState of n new speakers (online-ivector-feature.h:138-176) in the
        library's layout: CMVN speaker stats [2 (B + 1)], num_frames, prior share of the quadratic diagonal (1),
        linear term [S] (prior_offset in the first dimension), per-Gaussian counts [I]."""
        st = np.zeros((n, self.state_dim()))
        lo = 2 * (self.cfg.base_dim + 1) + 2
        st[:, lo - 1] = 1.0
        st[:, lo] = self.cfg.prior_offset
        return st

    def limit_frames(self, state, max_remembered_frames=1000.0):
        """OnlineIvectorExtractorAdaptationState::LimitFrames (online-ivector-feature.cc:99-117) +
        OnlineIvectorEstimationStats::Scale (ivector-extractor.cc:570-592), in place on [n x state_dim]."""
        B, S = self.cfg.base_dim, self.cfg.ivector_dim
        nc, lo = 2 * (B + 1), 2 * (B + 1) + 2
        mc, po = float(self.cfg.max_count), float(self.cfg.prior_offset)
        for st in state:
            count = np.float32(st[B])
            if count > max_remembered_frames:
                st[:nc] *= float(np.float32(max_remembered_frames) / count)
            lim = float(np.float32(max_remembered_frames) * np.float32(self.cfg.posterior_scale))
            n = st[lo - 2]
            if n > lim:
                scale = lim / n
                st[lo - 2] = n * scale
                st[lo + S:] *= scale                      # the counts: the data part of the quadratic term
                st[lo:lo + S] *= scale
                add = (1.0 - scale) if mc == 0.0 else max(n * scale, mc) / mc - scale * max(n, mc) / mc
                st[lo - 1] = st[lo - 1] * scale + add
                st[lo] += po * add
        return state

    def extract(self, feats, utt_row_offsets, out=None, state=None, return_state=False):
        """OnlineIvectorFeature::GetFrame for every frame: `feats` = device [sum T x base_dim] base
        features of the utterances row-concatenated -> device [sum T x ivector_dim].  state: [n_utts x
        state_dim()] adaptation states the utterances start from (SetAdaptationState); return_state:
        also the states after them, before LimitFrames (GetAdaptationState)."""
        off = np.ascontiguousarray(utt_row_offsets, np.int32)
        if feats.shape[1] != self.cfg.base_dim or off[-1] != feats.shape[0]:
            raise KhError("OnlineIvectorFeature: feature dimension / row offsets mismatch")
        if out is None:
            out = torch.empty((feats.shape[0], self.cfg.ivector_dim), dtype=torch.float32, device=feats.device)
        n = len(off) - 1
        if state is None and not return_state:
            check(lib().kh_ivector_extract(self._h, _p(feats), _dim(feats).stride, off.ctypes.data_as(capi.c_int32_p), n,
                                           _p(out), _dim(out).stride))
            return out
        sin = None
        if state is not None:
            sin = np.ascontiguousarray(state, np.float64)
            if sin.shape != (n, self.state_dim()):
                raise KhError("OnlineIvectorFeature::SetAdaptationState: state of the wrong shape")
        sout = np.empty((n, self.state_dim())) if return_state else None
        dp = capi.c_double_p
        check(lib().kh_ivector_extract_adapt(self._h, _p(feats), _dim(feats).stride, off.ctypes.data_as(capi.c_int32_p), n,
                                             sin.ctypes.data_as(dp) if sin is not None else None,
                                             sout.ctypes.data_as(dp) if sout is not None else None, _p(out), _dim(out).stride))
        return (out, sout) if return_state else out

    def __del__(self):
        if getattr(self, "_h", None):
            capi.load().kh_ivector_extractor_destroy(self._h)
            self._h = None


class OnlineIvectorStreams:
    """OnlineIvectorFeature objects of n utterances side by side WITH frame weights (online-ivector-feature.h:299-362):
    the feature side of the silence weighting.  feats: device [sum T x base_dim]; state: [n x state_dim] adaptation
    states (None = fresh); out: device [sum T x ivector_dim] (may be a column block of a wider feature matrix) whose rows
    get_frames() fills."""

    def __init__(self, extractor, feats, utt_row_offsets, out, state=None):
        self.extractor = extractor
        self.off = np.ascontiguousarray(utt_row_offsets, np.int32)
        if feats.shape[1] != extractor.cfg.base_dim or self.off[-1] != feats.shape[0] or out.shape != (feats.shape[0], extractor.cfg.ivector_dim):
            raise KhError("OnlineIvectorFeature: feature dimension / row offsets mismatch")
        self.n = len(self.off) - 1
        st = None if state is None else np.ascontiguousarray(state, np.float64)
        if st is not None and st.shape != (self.n, extractor.state_dim()):
            raise KhError("OnlineIvectorFeature: one adaptation state per utterance")
        self._keep = (feats, out)
        h = lib().kh_ivector_streams_create(extractor._h, _p(feats), _dim(feats).stride, self.off.ctypes.data_as(capi.c_int32_p), self.n,
                                            st.ctypes.data_as(capi.c_double_p) if st is not None else None, _p(out), _dim(out).stride)
        if not h:
            raise KhError(lib().kh_last_error().decode())
        self._h = C.c_void_p(h)

    def __del__(self):
        try:
            if self._h:
                lib().kh_ivector_streams_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def update_frame_weights(self, stream, delta_weights, num_frames_ready):
        """UpdateFrameWeights(delta_weights) :155-170; delta_weights = [(frame, delta), ...]."""
        fr = np.ascontiguousarray([d[0] for d in delta_weights], np.int32)
        w = np.ascontiguousarray([d[1] for d in delta_weights], np.float32)
        check(lib().kh_ivector_streams_update_frame_weights(self._h, int(stream), len(fr), fr.ctypes.data_as(capi.c_int32_p),
                                                            w.ctypes.data_as(capi.c_float_p), int(num_frames_ready)))

    def get_frames(self, streams, until_frames):
        """GetFrame(until) for every listed stream: the statistics advance to that frame, the iVector rows up to it are valid."""
        s = np.ascontiguousarray(streams, np.int32)
        u = np.ascontiguousarray(until_frames, np.int32)
        if len(s):
            check(lib().kh_ivector_streams_get_frames(self._h, len(s), s.ctypes.data_as(capi.c_int32_p), u.ctypes.data_as(capi.c_int32_p)))

    def get_stats(self, states):
        """Writes the OnlineIvectorEstimationStats part of GetAdaptationState() into states [n x state_dim] (CMVN part untouched)."""
        st = np.ascontiguousarray(states, np.float64)
        assert st.shape == (self.n, self.extractor.state_dim())
        check(lib().kh_ivector_streams_get_stats(self._h, st.ctypes.data_as(capi.c_double_p)))
        return st


class OnlineNnet2FeaturePipeline:
    """online2/online-nnet2-feature-pipeline.{h,cc} for a batch of whole utterances (what
    online2-wav-nnet2-latgen-faster --online=false feeds the decoder): base features (OnlineMfcc,
    :83) of every waveform, the iVector of every frame (OnlineIvectorFeature, :105-106) and
    OnlineAppendFeature (:107-108): row = [mfcc, ivector]."""

    def __init__(self, mfcc, ivector_extractor=None):
        self.mfcc, self.ivector = mfcc, ivector_extractor
        if ivector_extractor is not None and ivector_extractor.cfg.base_dim != mfcc.num_ceps:
            raise KhError("OnlineNnet2FeaturePipeline: iVector extractor expects %d-dim base features, MFCC gives %d"
                          % (ivector_extractor.cfg.base_dim, mfcc.num_ceps))

    def dim(self):
        return self.mfcc.num_ceps + (self.ivector.ivector_dim if self.ivector is not None else 0)

    def compute(self, waves, speakers=None, max_remembered_frames=1000.0):
        """waves: list of 1-D float32 device tensors.  Returns (features [sum T x Dim()] on the device,
        utterance row offsets); an utterance shorter than one frame contributes no rows.  speakers: one
        label per waveform — the utterances of a speaker are processed in list order with the adaptation
        state carried from one to the next (SetAdaptationState / GetAdaptationState,
        online2-wav-nnet2-latgen-faster.cc:186-200, :283), round by round over the speakers."""
        base = [self.mfcc.compute(w) for w in waves]
        lens = np.array([b.shape[0] for b in base], np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = int(off[-1])
        d0 = self.mfcc.num_ceps
        stride = (self.dim() + 3) // 4 * 4
        dev = waves[0].device if waves else "cuda"
        out = torch.empty((max(rows, 1), stride), dtype=torch.float32, device=dev)[:rows, :self.dim()]
        if rows == 0:
            return out, off
        for u, b in enumerate(base):
            if b.shape[0]:
                out[off[u]:off[u + 1], :d0] = b
        if self.ivector is None:
            return out, off
        if speakers is None:
            speakers = list(range(len(waves)))
        # round r: the r-th utterance (with frames) of every speaker, from the state its predecessor left
        queues = {}
        for u, s in enumerate(speakers):
            if lens[u] > 0:
                queues.setdefault(s, []).append(u)
        state = {s: self.ivector.fresh_state(1)[0] for s in queues}
        chained = any(len(q) > 1 for q in queues.values())
        r = 0
        while True:
            batch = [(s, q[r]) for s, q in queues.items() if r < len(q)]
            if not batch:
                break
            feats = torch.cat([base[u] for _, u in batch], 0).contiguous()
            boff = np.concatenate([[0], np.cumsum([lens[u] for _, u in batch])]).astype(np.int32)
            if chained:
                iv, st = self.ivector.extract(feats, boff, state=np.stack([state[s] for s, _ in batch]), return_state=True)
                self.ivector.limit_frames(st, max_remembered_frames)
                for j, (s, _) in enumerate(batch):
                    state[s] = st[j]
            else:
                iv = self.ivector.extract(feats, boff)
            for j, (_, u) in enumerate(batch):
                out[off[u]:off[u + 1], d0:] = iv[boff[j]:boff[j + 1]]
            r += 1
        return out, off
