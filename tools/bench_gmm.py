"""DiagGmm scoring rate (cfg 2: RM tri1, 39 dims, 1800 pdfs / 9000
Gaussians): frames/s of the frame x pdf matrix (all Gaussians + per-pdf LogSumExp) and
the share of the fp32 peak the per-Gaussian kernel reaches."""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
api = importlib.import_module("old-kaldi-git_amd.api")
W = importlib.import_module("old-kaldi-git_amd.workloads")
api.select_gpu(0)
rng = np.random.default_rng(1)
am = W.make_am_gmm(rng, 1800, 9000, 39)
means_invvars, inv_vars = W.gmm_inv_params(am)
gconsts, _ = api.gmm_compute_gconsts(am["weights"], means_invvars, inv_vars)
gmm = api.AmDiagGmm(gconsts, means_invvars, inv_vars, am["pdf_offsets"])
T = 200_000
x = torch.from_numpy(rng.standard_normal((T, 39)).astype(np.float32)).cuda()
out = torch.empty((T, 1800), dtype=torch.float32, device="cuda")
allg = torch.empty((T, 9000), dtype=torch.float32, device="cuda")
for name, fn in (("per-Gaussian loglikes (T x 9000)", lambda: gmm.log_likelihoods(x, out=allg)),
                 ("frame x pdf matrix (T x 1800)", lambda: gmm.pdf_log_likelihoods(x, out=out))):
    fn(); api.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    api.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    flops = T * 9000 * (2 * 78 + 20)
    print("%s: %.1f ms, %.2f M frames/s, %.1f TFLOP/s (algorithmic 176 flop per frame x Gaussian)" %
          (name, dt * 1e3, T / dt / 1e6, flops / dt / 1e12))
