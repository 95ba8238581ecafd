import sys, importlib, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
api = importlib.import_module('old-kaldi-git_amd.api'); api.select_gpu(0)
W = importlib.import_module('old-kaldi-git_amd.workloads')
from oracle import binding as B
def case(seed, n, npdf, Ts, cfg):
    rng = np.random.default_rng(seed)
    g = W.make_hclg_like(rng, n, npdf)
    lls = [W.make_loglikes(rng, T, npdf) for T in Ts]
    return g, lls, cfg
def run(g, lls, cfg):
    fst = api.Fst(g)
    dec = api.LatticeFasterDecoder(fst, cfg, max_batch=len(lls), max_frames=max(len(x) for x in lls), exact_reference_order=False)
    off = np.concatenate([[0], np.cumsum([len(x) for x in lls])]).astype(np.int32)
    dec.decode(torch.from_numpy(np.concatenate(lls, 0)).cuda(), off)
    global last_stats
    last_stats = [dec.counters(u) for u in range(len(lls))]
    return [dec.get_raw_lattice(u) for u in range(len(lls))]
c1 = case(1, 50, 10, (1, 2, 26, 60), api.decoder_config())
c2 = case(2, 20000, 200, (75, 130), api.decoder_config(beam=9.0, lattice_beam=6.0))
want = []
for x in c2[1]:
    o = B.DecoderOracle(c2[0], c2[2], 'canonical'); o.decode(x); want.append(o.raw_lattice())
import os
for it in range(int(os.environ.get('ITERS','8'))):
    if it % 2 == 1: run(*c1)
    got = run(*c2)
    print(it, [(len(a['state_frame']), len(b['state_frame'])) for a, b in zip(got, want)], [(s['tokens_created'], s['arcs_expanded'], round(s['final_best_cost'],4), s['max_tokens_frame']) for s in last_stats], flush=True)
