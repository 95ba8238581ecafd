"""GPU: tools/nnet_latgen_faster.py — the reference binary's command line over the
library, through the reference's file formats: final.mdl (TransitionModel + the AmNnet
bytes written by the REFERENCE, tests/golden/kaldi_io/am_nnet_body_bin), HCLG.fst,
feature archive in, lattice / words / alignment archives out; checked against the CPU
oracle run on the same arrays."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "kaldi_io")


def test_nnet_latgen_faster_files_in_files_out(api, oracle, tmp_path, monkeypatch):
    from oracle import binding
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    import nnet_latgen_faster as tool
    net, priors = make_golden.kaldi_io_net(np.random.default_rng(12))     # what am_nnet_body_bin holds
    n_pdf, acwt = 5, 0.2
    rng = np.random.default_rng(21)
    # TransitionModel: one phone per pdf, one emitting HMM state with a self-loop and an exit
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf,
                entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    pdf_of_phone = rng.permutation(n_pdf)
    triples = [(p + 1, 0, int(pdf_of_phone[p])) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    g = workloads.make_hclg_like(rng, 400, n_pdf, final_frac=0.2)
    g["tid2pdf"] = np.concatenate([[-1], np.repeat(pdf_of_phone, 2)]).astype(np.int32)
    monkeypatch.chdir(tmp_path)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_nnet_body_bin"), "rb").read())
    with open("HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    utts = {"spk1-utt%d" % i: rng.standard_normal((T, 6)).astype(np.float32) for i, T in enumerate((37, 5, 64))}
    with kio.TableWriter("feats.ark", "feats.scp") as w:
        for k, m in utts.items():
            w.write(k, m)
        w.write("empty", np.zeros((0, 6), np.float32))
    det_opts = ["--beam=9", "--max-active=300", "--lattice-beam=5", "--acoustic-scale=%g" % acwt, "--allow-partial=true"]
    opts = det_opts + ["--determinize-lattice=false"]
    assert tool.main(opts + ["final.mdl", "HCLG.fst", "ark:feats.ark", "ark:lat.ark", "ark:words.ark", "ark,t:ali.txt"]) == 0
    # the binaries' default: determinized CompactLattices (binary and text)
    assert tool.main(det_opts + ["final.mdl", "HCLG.fst", "ark:feats.ark", "ark:clat.ark"]) == 0
    assert tool.main(det_opts + ["final.mdl", "HCLG.fst", "ark:feats.ark", "ark,t:clat.txt"]) == 0
    assert tool.main(opts + ["--batch-frames=1", "final.mdl", "HCLG.fst", "scp:feats.scp", "ark,t:lat.txt"]) == 0
    lats = dict(kio.read_ark("lat.ark", kind="lattice"))
    lats_t = dict(kio.read_ark("lat.txt", kind="lattice"))
    words = dict(kio.read_ark("words.ark", kind="int32_vector"))
    alis = dict(kio.read_ark("ali.txt", kind="int32_vector"))
    assert sorted(lats) == sorted(utts) == sorted(lats_t) == sorted(words) == sorted(alis)
    cfg = binding.decoder_config(beam=9.0, max_active=300, lattice_beam=5.0)
    for k, x in utts.items():
        ll = oracle.decodable_am_nnet(net, priors, acwt, x)
        oc = binding.DecoderOracle(g, cfg, "canonical")
        oc.decode(ll)
        want, best = oc.raw_lattice(), oc.best_path()
        got = lats[k]
        assert got["num_states"] == len(want["state_frame"])
        order = np.argsort(want["arc_src"], kind="stable")     # the file groups arcs by source state
        for a, b in (("arc_src", "arc_src"), ("arc_dst", "arc_dst"), ("arc_il", "arc_il"), ("arc_ol", "arc_ol")):
            assert np.array_equal(got[a], want[b][order]), (k, a)
        # forward pass: 1e-4 on the scaled log-likelihoods (north_star), so on every cost
        np.testing.assert_allclose(got["arc_g"], want["arc_g"][order], atol=1e-4)
        np.testing.assert_allclose(got["arc_a"] * acwt, want["arc_a"][order], atol=2e-4)
        fin = np.isfinite(want["state_final"])
        assert np.array_equal(np.isfinite(got["state_final"]), fin)
        np.testing.assert_allclose(got["state_final"][fin], want["state_final"][fin], atol=1e-4)
        assert np.array_equal(words[k], best["words"]) and np.array_equal(alis[k], best["alignment"])
        # text lattice = the same lattice at 7 significant digits
        assert np.array_equal(lats_t[k]["arc_dst"], got["arc_dst"]) and np.array_equal(lats_t[k]["arc_il"], got["arc_il"])
        np.testing.assert_allclose(lats_t[k]["arc_a"], got["arc_a"], rtol=1e-6, atol=1e-6)
    # CompactLattices: deterministic on words; the best path carries the decoder's words, its
    # alignment (one transition-id per frame) and its cost (acoustic part unscaled); the text
    # form holds the same lattice
    clats = dict(kio.read_ark("clat.ark", kind="compact_lattice"))
    clats_t = dict(kio.read_ark("clat.txt", kind="compact_lattice"))
    assert sorted(clats) == sorted(utts) == sorted(clats_t)
    for k, x in utts.items():
        Cl, Ct = clats[k], clats_t[k]
        assert len(set(zip(Cl["arc_src"].tolist(), Cl["arc_label"].tolist()))) == len(Cl["arc_src"])
        assert Cl["n_states"] == Ct["n_states"] and np.array_equal(Cl["arc_dst"], Ct["arc_dst"])
        assert all(np.array_equal(p, q) for p, q in zip(Cl["arc_string"], Ct["arc_string"]))
        np.testing.assert_allclose(Cl["arc_a"], Ct["arc_a"], rtol=1e-6, atol=1e-5)
        n = Cl["n_states"]
        dist, back = np.full(n, np.inf), {}
        dist[0] = 0.0
        changed = True
        while changed:
            changed = False
            for j in range(len(Cl["arc_src"])):
                c = dist[Cl["arc_src"][j]] + Cl["arc_g"][j] + Cl["arc_a"][j] * acwt
                if c < dist[Cl["arc_dst"][j]] - 1e-9:
                    dist[Cl["arc_dst"][j]] = c
                    back[int(Cl["arc_dst"][j])] = j
                    changed = True
        c, st = min((dist[q] + Cl["final_g"][q] + Cl["final_a"][q] * acwt, q) for q in range(n) if np.isfinite(Cl["final_g"][q]))
        w, tids = [], list(Cl["final_string"][st])
        while st != 0:
            j = back[st]
            w.append(int(Cl["arc_label"][j]))
            tids = list(Cl["arc_string"][j]) + tids
            st = int(Cl["arc_src"][j])
        assert w[::-1] == words[k].tolist()
        assert len(tids) == len(x)


def test_gmm_latgen_faster_files_in_files_out(api, oracle, tmp_path, monkeypatch):
    """gmm-latgen-faster (configs 1 and 2): final.mdl = TransitionModel + the AmDiagGmm bytes
    written by the REFERENCE (tests/golden/kaldi_io/am_gmm_body_bin)."""
    from oracle import binding
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gmm_latgen_faster as tool
    am = workloads.make_am_gmm(np.random.default_rng(13), 5, 17, 6)     # what am_gmm_body_bin holds
    mi, iv = workloads.gmm_inv_params(am)
    n_pdf, acwt = 5, 0.1
    rng = np.random.default_rng(31)
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf,
                entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    pdf_of_phone = rng.permutation(n_pdf)
    triples = [(p + 1, 0, int(pdf_of_phone[p])) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    g = workloads.make_hclg_like(rng, 300, n_pdf, final_frac=0.2)
    g["tid2pdf"] = np.concatenate([[-1], np.repeat(pdf_of_phone, 2)]).astype(np.int32)
    monkeypatch.chdir(tmp_path)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_gmm_body_bin"), "rb").read())
    with open("HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    utts = {"u%d" % i: rng.standard_normal((T, 6)).astype(np.float32) for i, T in enumerate((41, 9))}
    with kio.TableWriter("feats.ark") as w:
        for k, m in utts.items():
            w.write(k, m)
    opts = ["--beam=10", "--max-active=200", "--lattice-beam=6", "--acoustic-scale=%g" % acwt, "--allow-partial=true",
            "--determinize-lattice=false"]
    assert tool.main(opts + ["final.mdl", "HCLG.fst", "ark:feats.ark", "ark:lat.ark", "ark,t:words.txt"]) == 0
    lats = dict(kio.read_ark("lat.ark", kind="lattice"))
    words = dict(kio.read_ark("words.txt", kind="int32_vector"))
    gconsts, _ = oracle.gmm_compute_gconsts(am["weights"], mi, iv)
    cfg = binding.decoder_config(beam=10.0, max_active=200, lattice_beam=6.0)
    for k, x in utts.items():
        ll = (oracle.am_gmm_loglikes(x, gconsts, mi, iv, am["pdf_offsets"], -1.0) * np.float32(acwt)).astype(np.float32)
        oc = binding.DecoderOracle(g, cfg, "canonical")
        oc.decode(ll)
        want, best = oc.raw_lattice(), oc.best_path()
        got = lats[k]
        order = np.argsort(want["arc_src"], kind="stable")
        assert got["num_states"] == len(want["state_frame"])
        for key in ("arc_dst", "arc_il", "arc_ol"):
            assert np.array_equal(got[key], want[key][order]), (k, key)
        np.testing.assert_allclose(got["arc_a"] * acwt, want["arc_a"][order], atol=2e-4)
        assert np.array_equal(words[k], best["words"])
