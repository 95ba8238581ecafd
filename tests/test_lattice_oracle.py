"""CPU: pins the lattice forward-backward restatement (oracle/lattice_oracle.cc)
by brute-force enumeration of every path of small lattices (posterior of an arc =
sum of path probabilities through it / total) and by the reference's own
self-checks (forward total == backward total, lat/lattice-functions.cc:346;
posteriors of every frame sum to one)."""
import importlib

import numpy as np
import pytest

from oracle import binding as B

workloads = importlib.import_module("old-kaldi-git_amd.workloads")


def random_lattice(rng, n_frames, width, eps_frac=0.2):
    """Top-sorted time-synchronous lattice: states grouped by frame; emitting arcs go to
    the next frame, epsilon arcs forward within the frame."""
    states = [[0]]
    n = 1
    for t in range(1, n_frames + 1):
        k = int(rng.integers(1, width + 1))
        states.append(list(range(n, n + k)))
        n += k
    arcs = [[] for _ in range(n)]
    for t in range(n_frames):
        for s in states[t]:
            for d in states[t + 1]:
                if rng.random() < 0.7 or s == states[t][0]:   # every state stays reachable
                    arcs[s].append((int(rng.integers(1, 50)), d, float(rng.random() * 3), float(rng.random() * 3)))
        for i, s in enumerate(states[t]):
            for d in states[t][i + 1:]:
                if rng.random() < eps_frac:
                    arcs[s].append((0, d, float(rng.random() * 2), 0.0))
    for s in states[n_frames][:-1]:
        pass
    off = np.zeros(n + 1, np.int64)
    il, ns, g, a = [], [], [], []
    for s in range(n):
        arcs[s].sort(key=lambda x: x[1])
        for (i, d, gg, aa) in arcs[s]:
            il.append(i); ns.append(d); g.append(gg); a.append(aa)
        off[s + 1] = len(il)
    final = np.full(n, np.inf, np.float32)
    for s in states[n_frames]:
        final[s] = rng.random() * 2
    return dict(n_states=n, arc_offsets=off, arc_ilabel=np.array(il, np.int32), arc_nextstate=np.array(ns, np.int32),
                arc_graph=np.array(g, np.float32), arc_acoustic=np.array(a, np.float32), state_final=final)


def brute_force(L):
    n = L["n_states"]
    off = L["arc_offsets"]
    cost = (L["arc_graph"] + L["arc_acoustic"]).astype(np.float64)   # float32 sum, as ConvertToCost
    post = np.zeros(len(cost))
    tot = 0.0
    stack = [(0, 0.0, [])]
    while stack:
        s, c, path = stack.pop()
        if np.isfinite(L["state_final"][s]):
            p = np.exp(-(c + float(L["state_final"][s])))
            tot += p
            for a in path:
                post[a] += p
        for a in range(off[s], off[s + 1]):
            stack.append((int(L["arc_nextstate"][a]), c + cost[a], path + [a]))
    return post / tot, np.log(tot)


@pytest.mark.parametrize("seed", range(5))
def test_forward_backward_matches_path_enumeration(seed):
    rng = np.random.default_rng(seed)
    L = random_lattice(rng, n_frames=5, width=3)
    out = B.lattice_forward_backward(L)
    post, logtot = brute_force(L)
    assert abs(out["tot_like"] - logtot) < 1e-9
    assert abs(out["tot_forward"] - out["tot_like"]) < 1e-9        # the reference's own check :346
    assert np.abs(out["arc_post"] - post).max() < 1e-6
    # LatticeStateTimes
    t = out["state_times"]
    assert t[0] == 0 and t.max() == 5
    # occupation of every frame sums to one (emitting arcs only)
    src = np.repeat(np.arange(L["n_states"]), np.diff(L["arc_offsets"]))
    em = L["arc_ilabel"] != 0
    for fr in range(5):
        assert abs(out["arc_post"][em & (t[src] == fr)].sum() - 1.0) < 1e-5
    ac = -(out["arc_post"].astype(np.float64) * L["arc_acoustic"]).sum()
    assert abs(out["acoustic_like_sum"] - ac) < 1e-4


def test_forward_backward_on_decoder_lattice():
    rng = np.random.default_rng(7)
    g = workloads.make_hclg_like(rng, 2000, 40)
    ll = workloads.make_loglikes(rng, 50, 40)
    dec = B.DecoderOracle(g, B.decoder_config(beam=12.0, max_active=500, lattice_beam=6.0), "canonical")
    assert dec.decode(ll)
    csr = B.lattice_csr(dec.raw_lattice())
    out = B.lattice_forward_backward(csr)
    assert abs(out["tot_forward"] - out["tot_like"]) < 1e-8
    bp = dec.best_path()
    # total likelihood >= best path likelihood
    assert out["tot_like"] >= -(bp["graph_cost"] + bp["acoustic_cost"]) - 1e-4
    assert out["state_times"].max() == 50
