"""MFCC (hires, 40 x 40) + CMVN statistics rate for one hour of 16 kHz audio on the device."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
api = importlib.import_module("old-kaldi-git_amd.api")
api.select_gpu(0)
n = 16000 * 3600
w = (1000 * torch.randn(n, device="cuda")).contiguous()
mf = api.Mfcc(num_bins=40, num_ceps=40, low_freq=40.0, high_freq=-200.0)
x = mf.compute(w); api.synchronize()
t0 = time.perf_counter(); x = mf.compute(w); api.synchronize(); dt = time.perf_counter() - t0
print("MFCC hires: %d frames in %.1f ms = %.1f M frames/s (%.0f x real time)" % (x.shape[0], dt * 1e3, x.shape[0] / dt / 1e6, 3600 / dt))
t0 = time.perf_counter(); st = api.acc_cmvn_stats(x); y = api.apply_cmvn(st, False, x); api.synchronize(); dt = time.perf_counter() - t0
print("CMVN stats + apply: %.1f ms" % (dt * 1e3))
