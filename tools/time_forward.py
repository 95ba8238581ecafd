#!/usr/bin/env python3
"""Wall time of the bench's forward pass alone (1.94 M frames in groups of <= 60 k rows): python tools/time_forward.py"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
import bench

api = importlib.import_module("old-kaldi-git_amd.api")
api.select_gpu(0)
net, priors, g, protos = bench.build_model_and_graph(3456, 100_000, False)
feats, off = bench.build_utterances(3456, 0, 2620, net, g, protos, False)
nnet = api.Nnet(net, priors)
x = torch.from_numpy(feats).cuda()
ll = torch.empty((int(off[-1]), net[-1]["output_dim"]), dtype=torch.float32, device="cuda")
for max_rows in [int(v) for v in (sys.argv[1:] or ["60000"])]:
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.forward_all(nnet, x, off, ll, max_rows)
        api.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("forward pass in groups of <= %d rows: %s ms; checksum %.10f" % (max_rows, " ".join("%.1f" % t for t in ts), float(ll.double().sum().item() / ll.numel())))
