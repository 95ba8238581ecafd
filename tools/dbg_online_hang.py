#!/usr/bin/env python3
"""Diagnostic: the chunk-of-5-frames serving loop of tools/bench_secondary.online2_cfg4 with a progress watchdog
(prints the status of a stream that stops advancing).  python tools/dbg_online_hang.py [streams]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

api = importlib.import_module("old-kaldi-git_amd.api")
bench = importlib.import_module("bench")
api.select_gpu(0)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net, priors, g, protos = bench.build_model_and_graph(3456, 2_000_000, False)
feats, off = bench.build_utterances(3456, 0, 2 * streams, net, g, protos, False)
dcfg, acwt = bench.DECODE_CFG, bench.ACWT
n = min(streams, len(off) - 1)
max_t = int(np.diff(off).max())
nnet = api.Nnet(net, priors)
dec = api.LatticeFasterOnlineDecoder(api.Fst(g), api.decoder_config(**dcfg), num_streams=n, max_frames=max_t)
off = np.asarray(off)
x_all = torch.from_numpy(np.ascontiguousarray(feats[:off[-1]])).cuda()
pipe = api.OnlineNnet2Pipeline(nnet, dec, max_frames=max_t, acoustic_scale=acwt, pad_input=True, max_nnet_batch_size=256)
all_lens = np.diff(off).astype(np.int64)
c = 5
slot_utt = np.arange(n)
given = np.zeros(n, np.int64)
pipe.reset(list(range(n)))
live = np.arange(n)
last = np.zeros(n, np.int64) - 1
stuck = np.zeros(n, np.int64)
t_start = time.time()
steps = 0
while len(live):
    u = slot_utt[live]
    cnt = np.minimum(c, all_lens[u] - given[live])
    fin = given[live] + cnt == all_lens[u]
    done = pipe.step(live, x_all, off[u].astype(np.int64) + given[live], cnt, fin)
    given[live] += cnt
    steps += 1
    same = done == last[live]
    stuck[live] = np.where(same, stuck[live] + 1, 0)
    last[live] = done
    bad = live[stuck[live] > 20]
    if len(bad):
        s = int(bad[0])
        print("STUCK stream", s, "utterance", int(slot_utt[s]), "len", int(all_lens[slot_utt[s]]), "given", int(given[s]), "done", int(last[s]), "after", steps, "steps")
        try:
            print("stats:", dec.stats(s, use_final_probs=False))
        except Exception as e:   # noqa: BLE001
            print("stats failed:", repr(e))
        sys.exit(3)
    ended = live[done >= all_lens[u]]
    if len(ended):
        dec.finalize_decoding(ended)
        live = np.array([s for s in live if s not in set(ended.tolist())], np.int64)
    if time.time() - t_start > 150:
        print("TIMEOUT in the loop after", steps, "steps; live", len(live), "min done/len", (last[live] / all_lens[slot_utt[live]]).min())
        sys.exit(4)
print("OK: all", n, "streams decoded in", steps, "steps,", round(time.time() - t_start, 1), "s")
