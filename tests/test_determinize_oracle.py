"""CPU: the differential oracle of lattice determinization.

oracle/determinize_oracle.cc restates DeterminizeLatticePhonePrunedWrapper and everything under it
(lat/determinize-lattice-pruned.cc, push-lattice.cc, minimize-lattice.cc) line by line; the product
(csrc/kh_determinize.hip) implements the same algorithm on its own data structures.  PARITY UNPINNED by the reference
(src/lat needs OpenFst: not compilable here, and it holds no determinized golden lattice), so

  * the ORACLE is pinned by the enumeration of every path of small lattices: for every option set its output is
    deterministic on words and holds, for every word sequence, the cost and the transition-id string of that sequence's
    best raw path (the reference's own test, lat/determinize-lattice-pruned-test.cc, checks determinism +
    RandEquivalent to the input);
  * the PRODUCT is diffed against the oracle: same number of states and arcs and the same weighted language with the
    same alignments (state numbering aside), on random lattices with finite beams, on the decoder oracle's raw
    lattices of an HCLG-structured graph, for phone_determinize / word_determinize / minimize in every combination
    the wrapper distinguishes, and with the memory limit that makes it retry on a pruned lattice.

The reference's defaults are phone_determinize = true, word_determinize = true, minimize = false
(determinize-lattice-pruned.h:163-167)."""
import importlib

import numpy as np
import pytest

import lattice_equiv as LE
from oracle import binding as B
from test_determinize import random_word_lattice, enumerate_raw, enumerate_clat, assert_deterministic

api = importlib.import_module("old-kaldi-git_amd.api")
workloads = importlib.import_module("old-kaldi-git_amd.workloads")

OPTION_SETS = [dict(phone_determinize=False), dict(phone_determinize=True), dict(phone_determinize=True, minimize=True),
               dict(phone_determinize=False, minimize=True), dict(phone_determinize=True, word_determinize=False)]


def random_tid_phone(rng, n_tid=40):
    tp = np.zeros(n_tid, np.int32)
    tp[1::3] = rng.integers(1, 5, len(tp[1::3]))     # a third of the transition-ids start a phone
    return tp


def same_result(co, cp, delta=1e-3):
    assert co["n_states"] == cp["n_states"] and len(co["arc_src"]) == len(cp["arc_src"]), \
        (co["n_states"], cp["n_states"], len(co["arc_src"]), len(cp["arc_src"]))
    assert sum(len(x) for x in co["arc_string"]) == sum(len(x) for x in cp["arc_string"])
    res = LE.compare_deterministic(co, cp, delta=delta)
    assert LE.deterministic_equal(res), res


@pytest.mark.parametrize("seed", range(10))
def test_oracle_against_path_enumeration(seed):
    rng = np.random.default_rng(seed)
    L = random_word_lattice(rng, n_frames=int(rng.integers(3, 7)), width=3)
    tp = random_tid_phone(rng)
    raw = enumerate_raw(L)
    for kw in OPTION_SETS[:4]:
        for faithful in (True, False):
            C = B.determinize_lattice_phone_pruned(L, 1e9, tp, faithful=faithful, **kw)
            assert C["ok"]
            assert_deterministic(C)
            got = enumerate_clat(C)
            assert set(got) == set(raw)
            for words, (c, g, a, tids) in got.items():
                rc, rg, ra, rt = raw[words]
                assert abs(c - rc) < 1e-4 and abs(g - rg) < 1e-4 and abs(a - ra) < 1e-4, (kw, words)
                assert tids == rt, (kw, words)


@pytest.mark.parametrize("seed", range(40))
def test_product_equals_oracle_on_random_lattices(seed):
    rng = np.random.default_rng(100 + seed)
    L = random_word_lattice(rng, int(rng.integers(2, 10)), int(rng.integers(1, 5)))
    tp = random_tid_phone(rng)
    beam = float(rng.choice([0.5, 2.0, 5.0, 1e9]))
    for kw in OPTION_SETS:
        co = B.determinize_lattice_phone_pruned(L, beam, tp, **kw)
        cp = api.determinize_lattice_pruned(L, beam, tid_phone=tp, **kw)
        assert co["ok"] == cp["complete"]
        if kw.get("word_determinize", True):
            same_result(co, cp)
        else:   # the first pass alone: not deterministic on words; same size, same best path cost
            assert co["n_states"] == cp["n_states"] and len(co["arc_src"]) == len(cp["arc_src"])


def test_product_equals_oracle_on_decoder_lattices_and_the_state_sharing_quirk(monkeypatch):
    """Raw lattices of the decoder oracle on an HCLG-structured graph (its transition-ids = {forward, self-loop} per pdf: a
    one-state phone per pdf), the recipe's lattice beam.  Also: this version of the reference shares output states only
    through the initial-subset table (MinimalToStateId lacks the `return`, determinize-lattice-pruned.cc:528-557);
    KH_DETERMINIZE_SHARE_MINIMAL=1 / faithful=False share by minimal subset - fewer states, and a language that can only
    grow (a shared state keeps the arcs its best path was allowed)."""
    rng = np.random.default_rng(303)
    n_pdf = 600
    g = workloads.make_hclg_structured(rng, 200_000, n_pdf)
    tp = np.zeros(2 * n_pdf + 1, np.int32)
    tp[1::2] = 1 + np.arange(n_pdf)             # transition-id 2 * pdf + 1 enters the state, 2 * pdf + 2 loops on it
    cfg = api.decoder_config(beam=13.0, max_active=2000, min_active=200, lattice_beam=7.0)
    n_states = []
    for q in workloads.sample_paths(rng, g, [120, 70]):
        x = (rng.standard_normal((len(q), n_pdf)) * 0.28 - 0.37).astype(np.float32)
        x[np.arange(len(q)), q] = (0.5 + 0.3 * rng.standard_normal(len(q))).astype(np.float32)
        oc = B.DecoderOracle(g, cfg, "canonical")
        assert oc.decode(x)
        L = oc.raw_lattice()
        for kw in OPTION_SETS[:3]:
            co = B.determinize_lattice_phone_pruned(L, 7.0, tp, **kw)
            cp = api.determinize_lattice_pruned(L, 7.0, tid_phone=tp, **kw)
            same_result(co, cp, delta=1e-2)
        monkeypatch.setenv("KH_DETERMINIZE_SHARE_MINIMAL", "1")
        shared = api.determinize_lattice_pruned(L, 7.0)
        monkeypatch.delenv("KH_DETERMINIZE_SHARE_MINIMAL")
        same_result(B.determinize_lattice_phone_pruned(L, 7.0, None, phone_determinize=False, faithful=False), shared, delta=1e-2)
        faithful = api.determinize_lattice_pruned(L, 7.0)
        n_states.append((faithful["n_states"], shared["n_states"]))
        res = LE.compare_deterministic(faithful, shared, delta=1e-2)
        assert res["only1"] == 0 and res["conflict"] == 0 and res["final_mismatch"] == 0, res
    assert all(a >= b for a, b in n_states), n_states


def test_memory_limit_retries_on_a_pruned_lattice_like_the_oracle():
    """max_mem stops a pass early; below half of the beam the wrapper prunes the lattice to a narrower beam and tries
    again (determinize-lattice-pruned.cc:1211-1241): same result, same `complete` flag as the oracle."""
    rng = np.random.default_rng(7)
    L = random_word_lattice(rng, 14, 4, n_words=6)
    seen = set()
    for max_mem in (2000, 20000, 200000):
        co = B.determinize_lattice_phone_pruned(L, 6.0, None, max_mem=max_mem, phone_determinize=False)
        cp = api.determinize_lattice_pruned(L, 6.0, max_mem=max_mem)
        assert co["ok"] == cp["complete"]
        seen.add(cp["complete"])
        assert co["n_states"] == cp["n_states"] and len(co["arc_src"]) == len(cp["arc_src"]), (max_mem, co["n_states"], cp["n_states"])
        if cp["n_states"]:
            assert LE.deterministic_equal(LE.compare_deterministic(co, cp))
    assert seen == {False, True}, seen


@pytest.mark.parametrize("case", [(300, 10, 0.0, 10.0, 40), (3000, 40, 0.3, 8.0, 40)])
def test_memory_limit_stops_either_pass_where_the_oracle_stops(case):
    """max_mem reached on dense decoder lattices, in the phone pass, in the word pass, in both: the pass stops at the same
    output state as the oracle (same sizes, same language, same `complete`).  That point depends on how many strings the
    epsilon closure has interned for candidates it later improves on, i.e. on the order in which it settles the states:
    the product numbers the states as the reference's TopSort calls do (determinize-lattice-pruned.cc:1410, :1417 always,
    :1505 only for a lattice that is not sorted as it stands; fst/topsort.h = reverse depth-first finishing order).  Round 6:
    tests/test_gpu_determinize.py found rounds 3-5's order (Kahn's) stopping 50 states away from the oracle."""
    n_states, n_pdf, eps, lat_beam, T = case
    rng = np.random.default_rng(950 + n_states)
    g = workloads.make_hclg_like(rng, n_states, n_pdf, eps_frac=eps)
    tp = np.zeros(int(g["ilabel"].max()) + 1, np.int32)
    tp[1::3] = rng.integers(1, 5, len(tp[1::3]))
    cfg = api.decoder_config(beam=11.0, max_active=800, min_active=20, lattice_beam=lat_beam)
    od = B.DecoderOracle(g, cfg, "reference")
    assert od.decode(workloads.make_loglikes(rng, T, n_pdf))
    raw = od.raw_lattice()
    stopped = 0
    for max_mem in (5000000, 500000):
        for kw in OPTION_SETS:
            phone = kw.get("phone_determinize", True)
            co = B.determinize_lattice_phone_pruned(raw, lat_beam, tp if phone else None, max_mem=max_mem, **kw)
            cp = api.determinize_lattice_pruned(raw, lat_beam, tid_phone=tp if phone else None, max_mem=max_mem, **kw)
            assert co["ok"] == cp["complete"], (max_mem, kw)
            stopped += not cp["complete"]
            assert co["n_states"] == cp["n_states"] and len(co["arc_src"]) == len(cp["arc_src"]), (max_mem, kw, co["n_states"], cp["n_states"])
            if kw.get("word_determinize", True) and cp["n_states"]:
                assert LE.deterministic_equal(LE.compare_deterministic(co, cp, delta=1e-2)), (max_mem, kw)
    assert stopped >= 2, stopped
