"""Adapter: the product's device API (old-kaldi-git_amd.api, i.e. the C-ABI of
libkaldi_hip.so) behind the same numpy-level interface as oracle.binding.OracleLib,
so the same golden/parity cases run against both."""
import numpy as np
import torch


class GpuImpl:
    kind = "gpu"

    def __init__(self, api):
        self.api = api
        self.dev = torch.device("cuda")

    def _d(self, a, pad=0):
        """numpy -> device tensor; pad > 0 embeds it in a wider buffer (stride != cols)."""
        a = np.ascontiguousarray(a, np.float32)
        if pad and a.ndim == 2:
            buf = torch.full((a.shape[0], a.shape[1] + pad), float("nan"), dtype=torch.float32, device=self.dev)
            buf[:, :a.shape[1]] = torch.from_numpy(a)
            return buf[:, :a.shape[1]]
        return torch.from_numpy(a).to(self.dev)

    def _empty(self, rows, cols, pad=3):
        buf = torch.full((rows, cols + pad), float("nan"), dtype=torch.float32, device=self.dev)
        return buf[:, :cols]

    @staticmethod
    def _h(t):
        torch.cuda.synchronize()
        return t.cpu().numpy().copy()

    def add_mat_mat(self, alpha, A, transA, B, transB, beta, Cm):
        c = self._d(Cm, pad=1)
        self.api.add_mat_mat(c, alpha, self._d(A, pad=4), transA, self._d(B), transB, beta)
        return self._h(c)

    def softmax_per_row(self, X):
        x = self._d(X, pad=2)
        return self._h(self.api.apply_softmax_per_row(self._empty(*X.shape), x))

    def log_softmax_per_row(self, X):
        x = self._d(X, pad=2)
        return self._h(self.api.apply_log_softmax_per_row(self._empty(*X.shape), x))

    def copy_rows(self, src, indices, dst=None):
        out = self._empty(len(indices), src.shape[1])
        return self._h(self.api.copy_rows(out, self._d(src, pad=5), np.asarray(indices, np.int32)))

    def splice(self, src, offsets):
        out = self._empty(src.shape[0], src.shape[1] * len(offsets))
        return self._h(self.api.splice(self._d(src, pad=1), np.asarray(offsets, np.int32), out))

    def group_pnorm(self, src, group, power):
        out = self._empty(src.shape[0], src.shape[1] // group)
        return self._h(self.api.group_pnorm(out, self._d(src, pad=2), power))

    def normalize(self, src):
        return self._h(self.api.normalize(self._empty(*src.shape), self._d(src, pad=1)))

    def add_diag_mat2(self, alpha, M, beta, v):
        vv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(self.dev)
        return self._h(self.api.add_diag_mat2(vv, alpha, self._d(M, pad=3), beta))

    def _vec(self, v):
        return torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(self.dev)

    def mul_rows_vec(self, M, s):
        return self._h(self.api.mul_rows_vec(self._d(M, pad=1), self._vec(s)))

    def mul_cols_vec(self, M, s):
        return self._h(self.api.mul_cols_vec(self._d(M, pad=1), self._vec(s)))

    def copy_rows_from_vec(self, rows, v):
        return self._h(self.api.copy_rows_from_vec(self._empty(rows, len(v)), self._vec(v)))

    def add_vec_to_rows(self, alpha, v, beta, M):
        return self._h(self.api.add_vec_to_rows(self._d(M, pad=1), alpha, self._vec(v), beta))

    def apply_floor(self, M, f):
        return self._h(self.api.apply_floor(self._d(M, pad=1), f))

    def apply_log(self, M):
        return self._h(self.api.apply_log(self._d(M, pad=1)))

    def apply_exp(self, M):
        return self._h(self.api.apply_exp(self._d(M, pad=1)))

    def apply_pow(self, M, p):
        return self._h(self.api.apply_pow(self._d(M, pad=1), p))

    def scale(self, M, a):
        return self._h(self.api.scale(self._d(M, pad=1), a))

    def sum_column_ranges(self, src, ranges):
        out = self._empty(src.shape[0], len(ranges) // 2)
        return self._h(self.api.sum_column_ranges(out, self._d(src, pad=2), np.asarray(ranges, np.int32)))

    def matrix_lookup(self, M, pairs):
        return self._h(self.api.lookup(self._d(M, pad=2), np.asarray(pairs, np.int32)))

    def nnet_context(self, net):
        n = self.api.Nnet(net)
        return (n.left_context(), n.right_context())

    def nnet_forward(self, net, feats, pad_input=True, utt_offsets=None):
        n = self.api.Nnet(net)
        out, off = n.compute(self._d(feats, pad=1), utt_offsets, pad_input)
        return self._h(out)

    def decodable_am_nnet(self, net, priors, prob_scale, feats, utt_offsets=None):
        n = self.api.Nnet(net, priors)
        out, off = n.compute(self._d(feats), utt_offsets, True, epilogue=True, prob_scale=prob_scale)
        return self._h(out)

    def gmm_compute_gconsts(self, w, mi, iv):
        return self.api.gmm_compute_gconsts(w, mi, iv)

    def diag_gmm_loglikes_stored(self, data, g, mi, iv):
        am = self.api.AmDiagGmm(g, mi, iv, [0, len(g)])
        return self._h(am.log_likelihoods(self._d(data, pad=1)))

    def am_gmm_loglikes(self, data, g, mi, iv, pdf_offsets, prune=-1.0):
        am = self.api.AmDiagGmm(g, mi, iv, pdf_offsets)
        return self._h(am.pdf_log_likelihoods(self._d(data, pad=1), prune))
