"""Seeded synthetic workloads for the BASELINE.json configs.

No corpora or trained models are available offline, so every model, feature
matrix and decoding graph is generated from a fixed seed with the shapes the
reference's recipes use (citations in each builder).  Pure numpy; no oracle, no
GPU dependency — shared by tests/, bench.py and __graft_entry__.smoke().
"""
import numpy as np


# --------------------------------------------------------------------------
# nnet2 p-norm networks (steps/nnet2/train_pnorm_simple2.sh:248-262 layout:
# Splice, FixedAffine(LDA-like), [Affine, Pnorm, Normalize] x L, Affine,
# Softmax, [SumGroup after mix-up]).
# --------------------------------------------------------------------------
def make_pnorm_net(rng, feat_dim, splice, const_dim, pnorm_in, pnorm_out, n_hidden,
                   n_mix, n_pdf, p=2.0, final_scale=1.0):
    """Returns (components, priors).  Components are dicts understood by both
    oracle.binding.pack_components and the product's Nnet2Forward."""
    ctx = list(range(-splice, splice + 1))
    sdim = (feat_dim - const_dim) * len(ctx) + const_dim
    net = [dict(type="splice", input_dim=feat_dim, output_dim=sdim, context=ctx, const_dim=const_dim)]

    def W(o, i, scale=1.0):
        return (rng.standard_normal((o, i)) * (scale / np.sqrt(i))).astype(np.float32)

    def b(o):
        return (rng.standard_normal(o) * 0.1).astype(np.float32)

    net.append(dict(type="fixed_affine", input_dim=sdim, output_dim=sdim, linear=W(sdim, sdim), bias=b(sdim)))
    cur = sdim
    for _ in range(n_hidden):
        net.append(dict(type="affine", input_dim=cur, output_dim=pnorm_in, linear=W(pnorm_in, cur, 2.0), bias=b(pnorm_in)))
        net.append(dict(type="pnorm", input_dim=pnorm_in, output_dim=pnorm_out, p=p))
        net.append(dict(type="normalize", input_dim=pnorm_out, output_dim=pnorm_out))
        cur = pnorm_out
    net.append(dict(type="affine", input_dim=cur, output_dim=n_mix, linear=W(n_mix, cur, final_scale), bias=b(n_mix)))
    net.append(dict(type="softmax", input_dim=n_mix, output_dim=n_mix))
    if n_mix != n_pdf:
        # mix-up: every pdf gets >= 1 softmax output (nnet2/mixup-nnet.cc gives
        # more to frequent pdfs; sizes here: 1 + multinomial remainder).
        sizes = 1 + rng.multinomial(n_mix - n_pdf, np.full(n_pdf, 1.0 / n_pdf))
        net.append(dict(type="sum_group", input_dim=n_mix, output_dim=n_pdf, sizes=sizes.astype(np.int32)))
    priors = rng.dirichlet(np.full(n_pdf, 5.0)).astype(np.float32)
    priors = np.maximum(priors, 1e-6).astype(np.float32)
    priors /= priors.sum()
    return net, priors.astype(np.float32)


def librispeech_nnet_a(rng, final_scale=2.0):
    """cfg 4: 40-dim hires MFCC + 100-dim iVector (const), splice +-7 -> 700,
    p-norm 3500->350 x4, mix-up 12000, ~5800 pdfs
    (egs/librispeech/s5/local/online/run_nnet2.sh; SURVEY.md §8 header)."""
    return make_pnorm_net(rng, feat_dim=140, splice=7, const_dim=100, pnorm_in=3500, pnorm_out=350,
                          n_hidden=4, n_mix=12000, n_pdf=5800, final_scale=final_scale)


def wsj_nnet5d(rng, final_scale=4.0):
    """cfg 3: input 40, splice +-4 -> 360, 4 x (2000 -> 400), 8000 mix -> 3400 pdfs
    (egs/wsj/s5/local/nnet2/run_5d.sh:49-58)."""
    return make_pnorm_net(rng, feat_dim=40, splice=4, const_dim=0, pnorm_in=2000, pnorm_out=400,
                          n_hidden=4, n_mix=8000, n_pdf=3400, final_scale=final_scale)


def tiny_net(rng, n_pdf=40):
    return make_pnorm_net(rng, feat_dim=13, splice=2, const_dim=3, pnorm_in=60, pnorm_out=12,
                          n_hidden=2, n_mix=2 * n_pdf, n_pdf=n_pdf, final_scale=3.0)


# --------------------------------------------------------------------------
# Diagonal GMM acoustic models (cfg 1 yesno mono, cfg 2 rm tri1;
# egs/rm/s5/run.sh:64-65: 1800 leaves / 9000 Gaussians).
# --------------------------------------------------------------------------
def make_am_gmm(rng, num_pdfs, tot_gauss, dim):
    per = np.full(num_pdfs, tot_gauss // num_pdfs, np.int64)
    per[: tot_gauss - per.sum()] += 1
    # jitter the mixture counts (+-2) keeping the total
    for _ in range(num_pdfs):
        i, j = rng.integers(0, num_pdfs, 2)
        if per[i] > 1:
            per[i] -= 1
            per[j] += 1
    offsets = np.zeros(num_pdfs + 1, np.int32)
    offsets[1:] = np.cumsum(per)
    M = int(offsets[-1])
    means = rng.standard_normal((M, dim)).astype(np.float32)
    vars_ = np.exp(rng.standard_normal((M, dim)) * 0.5).astype(np.float32)
    weights = np.empty(M, np.float32)
    for j in range(num_pdfs):
        n = offsets[j + 1] - offsets[j]
        weights[offsets[j]:offsets[j + 1]] = rng.dirichlet(np.full(n, 2.0))
    return dict(weights=weights, means=means, vars=vars_, pdf_offsets=offsets, dim=dim)


def gmm_inv_params(am):
    """inv_vars and means_invvars as DiagGmm::SetInvVarsAndMeans stores them
    (gmm/diag-gmm.cc: inv_vars = 1/var; means_invvars = mean * inv_var, float32).
    gconsts come from the library's ComputeGconsts (kh_gmm_compute_gconsts)."""
    inv_vars = (np.float32(1.0) / am["vars"]).astype(np.float32)
    means_invvars = (am["means"] * inv_vars).astype(np.float32)
    return means_invvars, inv_vars


# --------------------------------------------------------------------------
# HCLG-like decoding graphs: random sparse graph,
# out-degree ~ geometric, a fraction of epsilon arcs that are acyclic
# (epsilon arcs only go to higher state ids), ilabels are 1-based
# transition-ids mapped to pdfs.
# --------------------------------------------------------------------------
def make_graph(rng, num_states, mean_degree=2.5, eps_frac=0.15, num_tids=None, num_pdfs=100,
               num_words=1000, final_frac=0.02, weight_max=10.0, locality=None, start_degree=12):
    """Random HCLG-like graph in CSR form.  State 0 is the start state and gets
    `start_degree` emitting arcs; about half of the states carry an emitting
    self-loop (HMM self-transitions); `eps_frac` of the remaining arcs are
    input-epsilon arcs that only go to higher state ids (acyclic, as
    LatticeFasterDecoder requires: TopSortTokens asserts on epsilon loops,
    lattice-faster-decoder.cc:904-905)."""
    if num_tids is None:
        num_tids = 2 * num_pdfs
    deg = rng.geometric(1.0 / mean_degree, size=num_states).astype(np.int64)
    deg = np.minimum(deg, 64)
    deg[0] = max(int(deg[0]), start_degree)
    offsets = np.zeros(num_states + 1, np.int64)
    offsets[1:] = np.cumsum(deg)
    A = int(offsets[-1])
    src = np.repeat(np.arange(num_states, dtype=np.int64), deg)
    first = np.zeros(A, bool)
    first[offsets[:-1]] = True
    if locality is None:
        nxt = rng.integers(0, num_states, A)
    else:
        nxt = (src + rng.integers(-locality, locality + 1, A)) % num_states
    # self-loops on the first arc of ~half the states (never on the start state)
    selfloop = first & (rng.random(A) < 0.5) & (src != 0)
    nxt = np.where(selfloop, src, nxt)
    # epsilon arcs: not the first arc of a state, strictly forward
    fwd = src + 1 + rng.integers(0, max(1, num_states // 50), A)
    is_eps = (rng.random(A) < eps_frac) & ~first & (fwd < num_states) & (src != 0)
    nxt = np.where(is_eps, fwd, nxt)
    ilabel = np.where(is_eps, 0, rng.integers(1, num_tids + 1, A)).astype(np.int32)
    olabel = np.where(rng.random(A) < 0.1, rng.integers(1, num_words + 1, A), 0).astype(np.int32)
    weight = (rng.random(A) * weight_max).astype(np.float32)
    final = np.full(num_states, np.inf, np.float32)
    fin = rng.random(num_states) < final_frac
    final[fin] = (rng.random(int(fin.sum())) * 5.0).astype(np.float32)
    tid2pdf = np.zeros(num_tids + 1, np.int32)
    tid2pdf[1:] = rng.integers(0, num_pdfs, num_tids)
    return dict(num_states=num_states, start=0, arc_offsets=offsets.astype(np.int64),
                ilabel=ilabel, olabel=olabel, weight=weight, nextstate=nxt.astype(np.int32),
                final=final, tid2pdf=tid2pdf)


def make_hclg_like(rng, num_states, num_pdfs, self_loop_floor=0.5, **kw):
    """make_graph with HCLG-like arc costs: HMM-transition-sized costs U[0, 1.5) on
    ordinary arcs, self-loops no cheaper than `self_loop_floor` (a -log
    self-transition probability is never ~0; without the floor a random graph grows
    zero-cost self-loop attractors and the active-token population collapses),
    LM-sized costs U[2, 10) on word-emitting (olabel != 0) arcs.  With this graph
    and the decode.sh options (beam 15, max-active 7000) the search stays in the
    max-active-bound regime (~10 k tokens, ~18 k arcs per frame), which puts the
    single-thread CPU reference path at RTF ~0.9 — the order of the reference's
    published 1.62 on 2015 hardware (src/doc/online_decoding.dox:288-304)."""
    g = make_graph(rng, num_states, num_pdfs=num_pdfs, weight_max=1.0, **kw)
    A = len(g["ilabel"])
    src = np.repeat(np.arange(num_states), np.diff(g["arc_offsets"]))
    selfloop = g["nextstate"] == src
    has_o = g["olabel"] != 0
    w = rng.random(A) * 1.5
    w = np.where(selfloop, self_loop_floor + rng.random(A) * (1.5 - self_loop_floor), w)
    w = np.where(has_o & ~selfloop, 2.0 + rng.random(A) * 8.0, w)
    g["weight"] = w.astype(np.float32)
    return g


def utterance_lengths(rng, n_utts, mean=740, max_len=3500, min_len=100):
    """Length profile of LibriSpeech test-clean."""
    sigma = 0.6
    mu = np.log(mean) - sigma * sigma / 2
    T = np.exp(rng.normal(mu, sigma, n_utts))
    return np.clip(T, min_len, max_len).astype(np.int64)


def make_loglikes(rng, T, num_pdfs, peak=6.0, acwt=0.1, stickiness=0.9):
    """Synthetic scaled acoustic log-likelihood matrix with a realistic shape:
    one 'true' pdf per frame that persists over time, log-softmax normalised,
    minus log-prior ~ uniform, times acwt (decodable-am-nnet.h:60-69 output)."""
    x = rng.standard_normal((T, num_pdfs)).astype(np.float32)
    cur = rng.integers(0, num_pdfs)
    for t in range(T):
        if rng.random() > stickiness:
            cur = rng.integers(0, num_pdfs)
        x[t, cur] += peak
    x -= np.log(np.exp(x - x.max(1, keepdims=True)).sum(1, keepdims=True)) + x.max(1, keepdims=True)
    x -= np.float32(np.log(1.0 / num_pdfs))
    return (x * np.float32(acwt)).astype(np.float32)
