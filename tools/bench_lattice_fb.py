"""Lattice forward-backward rate (BASELINE config 5 shape): raw lattices
of N decoded utterances -> LatticeForwardBackward and the sMBR variant, batch calls;
arcs/s including the host preparation (levels, incoming-arc CSR) and uploads."""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
api = importlib.import_module("old-kaldi-git_amd.api")
W = importlib.import_module("old-kaldi-git_amd.workloads")
from oracle import binding as B   # lattice_csr (host top-sort helper of the tests), checker only

api.select_gpu(0)
rng = np.random.default_rng(5)
N, T, P = 256, 500, 2000
g = W.make_hclg_like(rng, 1_000_000, P)
cfg = api.decoder_config(beam=13.0, max_active=3000, min_active=200, lattice_beam=8.0)
ll = torch.from_numpy(np.stack([W.make_loglikes(rng, T, P) for _ in range(8)])).cuda()
flat = ll[torch.arange(N) % 8].reshape(N * T, P).contiguous()
dec = api.LatticeFasterDecoder(api.Fst(g), cfg, max_batch=N, max_frames=T)
dec.decode(flat, (np.arange(N + 1) * T).astype(np.int32))
dec.prepare()
lats = [B.lattice_csr(dec.get_raw_lattice(u)) for u in range(N)]
arcs = sum(len(L["arc_ilabel"]) for L in lats)
states = sum(L["n_states"] for L in lats)
print("%d lattices, %d states, %d arcs (%.0f arcs per frame)" % (N, states, arcs, arcs / (N * T)))
ntid = int(max(L["arc_ilabel"].max() for L in lats))
t2ph = np.concatenate([[0], rng.integers(1, 40, ntid)]).astype(np.int32)
t2pdf = np.concatenate([[0], rng.integers(0, P, ntid)]).astype(np.int32)
alis = [rng.integers(1, ntid + 1, T).astype(np.int32) for _ in range(N)]
for name, fn in (("LatticeForwardBackward", lambda: api.lattice_forward_backward(lats)),
                 ("LatticeForwardBackwardMpeVariants(smbr)", lambda: api.lattice_forward_backward_mpe(lats, t2ph, t2pdf, [1, 2], alis, "smbr", True))):
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print("%s: %.1f ms per batch, %.1f M arcs/s, %.2f M frames/s (host posterior merge included)" %
          (name, dt * 1e3, arcs / dt / 1e6, N * T / dt / 1e6))

# the C call alone (host preparation + uploads + kernel + download), without the Python lists
import ctypes as C
capi = importlib.import_module("old-kaldi-git_amd.capi")
n, soff, aoff, il, ns, gg, aa, fin = api._cat_lattices(lats)
post = np.empty(len(il), np.float32)
tot, ac = np.empty(n), np.empty(n)
times = np.empty(int(soff[-1]), np.int32)
ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p
def call():
    api.check(api.lib().kh_lattice_forward_backward(
        n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
        gg.ctypes.data_as(fp), aa.ctypes.data_as(fp), fin.ctypes.data_as(fp), post.ctypes.data_as(fp),
        tot.ctypes.data_as(dp), ac.ctypes.data_as(dp), times.ctypes.data_as(ip)))
call()
t0 = time.perf_counter(); call(); dt = time.perf_counter() - t0
print("kh_lattice_forward_backward alone: %.1f ms per batch, %.1f M arcs/s" % (dt * 1e3, arcs / dt / 1e6))
