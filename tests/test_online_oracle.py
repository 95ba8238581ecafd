"""Oracle restatement of the online call sequence (LatticeFasterOnlineDecoder,
decoder/lattice-faster-online-decoder.cc:55-72,747-769,775-790): advancing in chunks
must give the lattice of the one-shot Decode(), and an un-finalized GetRawLattice
keeps every token alive on the last frame."""
import importlib

import numpy as np

from oracle import binding as B

workloads = importlib.import_module("old-kaldi-git_amd.workloads")


def test_chunked_advance_equals_one_shot_decode():
    rng = np.random.default_rng(5)
    g = workloads.make_hclg_like(rng, 3000, 60)
    ll = workloads.make_loglikes(rng, 83, 60)
    cfg = B.decoder_config(beam=10.0, max_active=400, min_active=50, lattice_beam=5.0, prune_interval=10)
    for mode in ("reference", "canonical"):
        one = B.DecoderOracle(g, cfg, mode)
        assert one.decode(ll)
        chunked = B.DecoderOracle(g, cfg, mode)
        chunked.begin(ll)
        t = 0
        for n in (1, 7, 25, 0, 13, 100):
            t = chunked.advance(n)
        assert t == 83
        chunked.finalize()
        chunked.snapshot(True)
        assert chunked.raw_lattice().key() == one.raw_lattice().key()
        assert np.array_equal(chunked.best_path()["words"], one.best_path()["words"])


def test_unfinalized_lattice_has_every_last_frame_token_final():
    rng = np.random.default_rng(6)
    g = workloads.make_hclg_like(rng, 3000, 60)
    ll = workloads.make_loglikes(rng, 40, 60)
    cfg = B.decoder_config(beam=10.0, max_active=400, min_active=50, lattice_beam=5.0, prune_interval=10)
    d = B.DecoderOracle(g, cfg, "canonical")
    d.begin(ll)
    assert d.advance(17) == 17
    d.snapshot(False)  # use_final_probs = false: every token of frame 17 is final with weight One
    L = d.raw_lattice()
    last = L["state_frame"] == 17
    assert last.any() and np.all(L["state_final"][last] == 0.0) and np.all(np.isinf(L["state_final"][~last]))
    d.snapshot(True)   # with final probs: only HCLG-final states (or all, if none is final)
    Lf = d.raw_lattice()
    assert np.array_equal(Lf["state_frame"], L["state_frame"])  # same tokens, same links
    assert np.array_equal(Lf["arc_src"], L["arc_src"])
    fin = np.asarray(g["final"], np.float32)[Lf["state_hclg"][last]]
    if np.isfinite(fin).any():
        assert np.array_equal(Lf["state_final"][last], fin)
    else:
        assert np.all(Lf["state_final"][last] == 0.0)
    # decoding continues unharmed after the snapshots
    assert d.advance(-1) == 40
    d.finalize()
    d.snapshot(True)
    one = B.DecoderOracle(g, cfg, "canonical")
    assert one.decode(ll)
    assert d.raw_lattice().key() == one.raw_lattice().key()
