import sys, importlib, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
api = importlib.import_module('old-kaldi-git_amd.api'); api.select_gpu(0)
W = importlib.import_module('old-kaldi-git_amd.workloads')
from oracle import binding as B
seed = int(os.environ.get("SEED", "1"))
rng = np.random.default_rng(1000 + seed)
n_states = int(rng.choice([30, 300, 3000, 20000]))
n_pdf = int(rng.choice([5, 40, 200]))
g = W.make_hclg_like(rng, n_states, n_pdf, eps_frac=float(rng.choice([0.0, 0.05, 0.2, 0.4])), final_frac=float(rng.choice([0.0, 0.05, 0.5])))
n_utt = int(rng.integers(1, 6))
Ts = rng.integers(1, 130, n_utt)
lls = [W.make_loglikes(rng, int(T), n_pdf) for T in Ts]
max_active = int(rng.choice([2, 5, 60, 800, 2147483647]))
min_active = int(rng.choice([m for m in (0, 1, 20, 300) if m < max_active]))
cfg = api.decoder_config(beam=float(rng.choice([2.0, 6.0, 11.0, 15.0])), max_active=max_active, min_active=min_active,
                         lattice_beam=float(rng.choice([0.3, 2.0, 6.0, 10.0])),
                         prune_interval=int(rng.choice([1, 2, 7, 25, 30])), beam_delta=float(rng.choice([0.1, 0.5])),
                         prune_scale=float(rng.choice([0.05, 0.1, 0.5])))
print("states", n_states, "pdf", n_pdf, "Ts", Ts, cfg)
fst = api.Fst(g)
dec = api.LatticeFasterDecoder(fst, cfg, max_batch=len(lls), max_frames=int(max(Ts)), exact_reference_order=False)
off = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int32)
dec.decode(torch.from_numpy(np.concatenate(lls, 0)).cuda(), off)
for u, x in enumerate(lls):
    oc = B.DecoderOracle(g, cfg, "canonical"); oc.decode(x)
    want, got = oc.raw_lattice(), dec.get_raw_lattice(u)
    so, sg = oc.stats(), dec.stats(u)
    print(u, "T", len(x), "states", len(got["state_frame"]), len(want["state_frame"]), "arcs", len(got["arc_src"]), len(want["arc_src"]),
          "created", sg["tokens_created"], so["tokens_created"], "maxtok", sg["max_tokens_frame"], so["max_tokens_frame"])
    for k in got:
        if got[k].shape == want[k].shape and not np.array_equal(got[k].view(np.int32), want[k].view(np.int32)):
            bad = np.nonzero(got[k].view(np.int32) != want[k].view(np.int32))[0]
            print("  key", k, "differs at", bad[:8], got[k][bad[:8]], want[k][bad[:8]])
    if len(got["state_frame"]) != len(want["state_frame"]):
        gf = np.bincount(got["state_frame"], minlength=len(x) + 1); wf = np.bincount(want["state_frame"], minlength=len(x) + 1)
        bad = np.nonzero(gf != wf)[0]
        print("  first differing frames", bad[:10], gf[bad[:10]], wf[bad[:10]])
    bo, bg = oc.best_path(), dec.get_best_path(u)
    if not np.array_equal(bo["alignment"], bg["alignment"]):
        d = np.nonzero(bo["alignment"] != bg["alignment"])[0]
        print("  best path differs at frames", d[:10], bo["alignment"][d[:10]], bg["alignment"][d[:10]], "costs", bo["graph_cost"], bo["acoustic_cost"], bg["graph_cost"], bg["acoustic_cost"])
    for k in ("final_relative_cost", "reached_final", "num_frames"):
        if np.float32(so[k]).tobytes() != np.float32(sg[k]).tobytes():
            print("  stat", k, so[k], sg[k])
    if np.float32(bo["graph_cost"]).tobytes() != np.float32(bg["graph_cost"]).tobytes() or np.float32(bo["acoustic_cost"]).tobytes() != np.float32(bg["acoustic_cost"]).tobytes():
        print("  best path cost", bo["graph_cost"], bg["graph_cost"], bo["acoustic_cost"], bg["acoustic_cost"])
    orf = B.DecoderOracle(g, cfg, "reference"); orf.decode(x); br = orf.best_path()
    print("  reference-order best path same words:", np.array_equal(br["words"], bg["words"]), "same ali:", np.array_equal(br["alignment"], bg["alignment"]), br["graph_cost"], br["acoustic_cost"], bg["graph_cost"], bg["acoustic_cost"])
