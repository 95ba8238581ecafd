#!/usr/bin/env python3
"""Does the lattice forward-backward leg run at idle clocks?  The sweeps of one resident 256-lattice batch called back to back
(no host work in between), then again after a GEMM burst that holds the chip busy: sweeps_ms of every call
(kh_lattice_last_timings).  python tools/exp_lattice_clock.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "old-kaldi-git_amd"


def main():
    import torch
    api = importlib.import_module(PKG + ".api")
    W = importlib.import_module(PKG + ".workloads")
    api.select_gpu(0)
    N, T, P = 256, 400, 2000
    rng = np.random.default_rng(5)
    g = W.make_hclg_structured(rng, 1_000_000, P)
    seqs = W.sample_paths(rng, g, [T] * N)
    lls = []
    for q in seqs:
        x = (rng.standard_normal((T, P)) * 0.28 - 0.37).astype(np.float32)
        x[np.arange(T), q] = (0.5 + 0.3 * rng.standard_normal(T)).astype(np.float32)
        lls.append(x)
    flat = torch.from_numpy(np.concatenate(lls)).cuda()
    dec = api.LatticeFasterDecoder(api.Fst(g), api.decoder_config(beam=13.0, max_active=7000, min_active=200, lattice_beam=8.0),
                                   max_batch=N, max_frames=T)
    dec.decode(flat, (np.arange(N + 1) * T).astype(np.int32))
    lats = [api.lattice_to_csr(dec.get_raw_lattice(u)) for u in range(N)]
    del dec, flat
    cat = api._cat_lattices(lats)
    lo = np.asarray(cat[1])
    nst = np.diff(lo)
    print("states per lattice: min %d median %d mean %.0f p90 %d max %d; lattices over 4096 / 5120 states: %d / %d" % (
        nst.min(), np.median(nst), nst.mean(), np.percentile(nst, 90), nst.max(), (nst > 4096).sum(), (nst > 5120).sum()))
    na = len(cat[3])
    B = api.LatticeBatch(cat)
    dev_post = torch.empty(na, dtype=torch.float32, device="cuda")
    a = torch.randn(8192, 8192, device="cuda")
    for label, burst in (("after 0.5 s of host sleep", 0), ("back to back", 0), ("behind a GEMM burst", 30)):
        if label.startswith("after"):
            time.sleep(0.5)
        for _ in range(burst):
            a @ a
        ms = []
        for _ in range(12):
            B.forward_backward_device(dev_post)
            ms.append(api.lattice_last_timings()["sweeps_ms"])
            if burst:
                for _ in range(3):
                    a @ a
        print("%-28s sweeps_ms per call: %s" % (label, " ".join("%.2f" % v for v in ms)))


if __name__ == "__main__":
    main()
