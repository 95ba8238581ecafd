"""Throughput of the online call sequence: N streams advanced in chunks of C frames
(one launch per round of chunks), same workload family as bench.py (--small graph/model
by default so that it runs in seconds).  Prints frames/s for a few chunk sizes next to the
one-shot decode of the same utterances."""
import argparse
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
api = importlib.import_module("old-kaldi-git_amd.api")
W = importlib.import_module("old-kaldi-git_amd.workloads")

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=512)
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--graph-states", type=int, default=1_000_000)
ap.add_argument("--pdfs", type=int, default=2000)
args = ap.parse_args()
api.select_gpu(0)
rng = np.random.default_rng(3)
g = W.make_hclg_like(rng, args.graph_states, args.pdfs)
fst = api.Fst(g)
cfg = api.decoder_config(beam=13.0, max_active=3000, min_active=200, lattice_beam=6.0)
N, T = args.streams, args.frames
ll = torch.from_numpy(np.stack([W.make_loglikes(rng, T, args.pdfs) for _ in range(8)])).cuda()
ll = ll[torch.arange(N) % 8].contiguous()            # [N, T, pdfs]
flat = ll.reshape(N * T, args.pdfs)
off = (np.arange(N + 1) * T).astype(np.int32)
dec = api.LatticeFasterDecoder(fst, cfg, max_batch=N, max_frames=T)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.decode(flat, off)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("one-shot decode: %.0f frames/s (%d streams x %d frames, %.2f s)" % (N * T / dt, N, T, dt))
ref = dec.get_best_path(0)
on = api.LatticeFasterOnlineDecoder(fst, cfg, num_streams=N, max_frames=T)
streams = list(range(N))
for chunk in (10, 30, 100):
    on.init_decoding(streams)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(0, T, chunk):
        on.advance_decoding(streams, [ll[s, t:t + chunk] for s in streams])
    on.finalize_decoding(streams)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bp = on.get_best_path(0)
    assert np.array_equal(bp["alignment"], ref["alignment"])
    print("online, chunks of %3d frames: %.0f frames/s (%.2f s, %d launches)" % (chunk, N * T / dt, dt, (T + chunk - 1) // chunk + 1))
