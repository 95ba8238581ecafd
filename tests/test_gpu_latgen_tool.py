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


@pytest.mark.parametrize("rule", ["canonical", "reference"])
def test_nnet_latgen_faster_files_in_files_out(api, oracle, tmp_path, monkeypatch, rule):
    """rule = reference: the tool's default, the lattices of LatticeFasterDecoder's own iteration order (oracle mode 0) -
    what nnet2bin/nnet-latgen-faster.cc:139-160 itself writes; max-active binds in this test, so the two orders are
    different searches."""
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
    if rule == "canonical":      # (the tool's default is the reference's own order)
        det_opts.append("--canonical-order=true")
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
        oc = binding.DecoderOracle(g, cfg, rule)
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


def test_nnet_latgen_faster_with_the_recipe_command_line(api, oracle, tmp_path, monkeypatch, capfd):
    """The literal argv of steps/nnet2/decode.sh:130-136 (egs/wsj/s5): --minimize, --word-symbol-table, features through
    an "ark,s,cs:... |" pipeline, lattices into "ark:|gzip -c > lat.JOB.gz"; the model through a pipe as well (the recipes
    pass "nnet-am-copy ... - |" style models); --config; ParseOptions' error / usage exit codes."""
    import gzip
    from oracle import binding
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    import nnet_latgen_faster as tool
    net, priors = make_golden.kaldi_io_net(np.random.default_rng(12))
    n_pdf, acwt = 5, 0.2
    rng = np.random.default_rng(22)
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf,
                entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    pdf_of_phone = rng.permutation(n_pdf)
    triples = [(p + 1, 0, int(pdf_of_phone[p])) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    g = workloads.make_hclg_like(rng, 400, n_pdf, final_frac=0.2)
    g["tid2pdf"] = np.concatenate([[-1], np.repeat(pdf_of_phone, 2)]).astype(np.int32)
    monkeypatch.chdir(tmp_path)
    os.makedirs("exp/decode", exist_ok=True)
    os.makedirs("graph", exist_ok=True)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_nnet_body_bin"), "rb").read())
    with open("graph/HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    n_words = int(g["olabel"].max())
    with open("graph/words.txt", "w") as f:
        f.write("<eps> 0\n" + "".join("W%d %d\n" % (i, i) for i in range(1, n_words + 1)))
    utts = {"spk1-utt%d" % i: rng.standard_normal((T, 6)).astype(np.float32) for i, T in enumerate((37, 5, 64))}
    with kio.TableWriter("feats.ark", "feats.scp") as w:
        for k, m in utts.items():
            w.write(k, m)
    argv = ["--minimize=false", "--max-active=300", "--min-active=200", "--beam=9", "--lattice-beam=5", "--acoustic-scale=%g" % acwt,
            "--allow-partial=true", "--word-symbol-table=graph/words.txt", "cat final.mdl |", "graph/HCLG.fst",
            "ark,s,cs:cat feats.ark | cat |", "ark:|gzip -c > exp/decode/lat.1.gz"]
    capfd.readouterr()
    assert tool.main(argv) == 0
    err = capfd.readouterr().err
    # the command line is echoed, the best paths are printed through the symbol table, the binary's summary lines
    assert err.startswith("nnet-latgen-faster --minimize=false --max-active=300") and "'ark:|gzip -c > exp/decode/lat.1.gz'" in err
    assert "LOG (nnet-latgen-faster:main()) Done 3 utterances, failed for 0" in err
    assert "LOG (nnet-latgen-faster:main()) Overall log-likelihood per frame is" in err
    # the same through plain files + a config file: identical archive bytes
    with open("decode.conf", "w") as f:
        f.write("--max-active=300 # recipe\n--min-active=200\n--beam=9\n--lattice-beam=5\n--allow-partial=true\n")
    assert tool.main(["--config=decode.conf", "--acoustic-scale=%g" % acwt, "--print-args=false", "final.mdl", "graph/HCLG.fst",
                      "scp:feats.scp", "ark:lat.ark", "ark,t:words.txt"]) == 0
    assert gzip.open("exp/decode/lat.1.gz").read() == open("lat.ark", "rb").read()
    clats = dict(kio.read_ark("lat.ark", kind="compact_lattice"))
    words = dict(kio.read_ark("words.txt", kind="int32_vector"))
    cfg = binding.decoder_config(beam=9.0, max_active=300, lattice_beam=5.0)
    api_mod = pkg("api")
    tp = api_mod.tid_phone_map(kio.read_nnet2_model("final.mdl")[0])
    assert np.array_equal(tp[1:], np.ravel(np.stack([np.zeros(n_pdf, np.int32), np.arange(1, n_pdf + 1)], 1)))
    for k, x in utts.items():
        ll = oracle.decodable_am_nnet(net, priors, acwt, x)
        oc = binding.DecoderOracle(g, cfg, "reference")
        oc.decode(ll)
        assert np.array_equal(words[k], oc.best_path()["words"])
        assert "%s %s\n" % (k, "".join("W%d " % w for w in words[k])) in err     # utt W3 W17 ... (decoder-wrappers.cc:247-256)
        # the CompactLattice = the reference's determinization (phone + word passes: its defaults) of the oracle's raw lattice
        want = binding.determinize_lattice_phone_pruned(oc.raw_lattice(), 5.0, tp)
        got = dict(clats[k])
        got["arc_a"] = got["arc_a"] * np.float32(acwt)
        got["final_a"] = got["final_a"] * np.float32(acwt)
        import lattice_equiv as LE
        assert got["n_states"] == want["n_states"] and len(got["arc_src"]) == len(want["arc_src"])
        assert LE.deterministic_equal(LE.compare_deterministic(got, want, delta=2e-3)), k
    # --minimize=true: fewer or as many states, the same language
    assert tool.main(argv[:0] + ["--minimize=true"] + argv[1:-1] + ["ark:lat_min.ark"]) == 0
    for k, c in kio.read_ark("lat_min.ark", kind="compact_lattice"):
        assert c["n_states"] <= clats[k]["n_states"]
        assert LE.deterministic_equal(LE.compare_deterministic(c, clats[k], delta=1e-2))
    # exit codes: usage (too few arguments) 1, an invalid option / unreadable model 255 ("return -1"), --help 0
    assert tool.main(["final.mdl"]) == 1
    assert tool.main(["--no-such-option=1", "final.mdl", "graph/HCLG.fst", "ark:feats.ark", "ark:x.ark"]) == 255
    assert tool.main(["nope.mdl", "graph/HCLG.fst", "ark:feats.ark", "ark:x.ark"]) == 255
    with pytest.raises(SystemExit) as e:
        tool.main(["--help"])
    assert e.value.code == 0


def test_lattice_to_post_tool(api, tmp_path, monkeypatch):
    """tools/lattice_to_post.py = latbin/lattice-to-post.cc: CompactLattices (as the decoders write them, gzipped, through
    a pipe) and state-level lattices in, Posteriors + per-utterance log-likelihoods out, --acoustic-scale / --lm-scale;
    against the CPU oracle's LatticeForwardBackward on the same (scaled, top-sorted) lattice."""
    import gzip
    from oracle import binding
    kio = pkg("kaldi_io")
    api_mod = pkg("api")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lattice_to_post as tool
    from test_determinize import random_word_lattice
    rng = np.random.default_rng(5)
    monkeypatch.chdir(tmp_path)
    raws = {"utt%d" % i: random_word_lattice(rng, int(rng.integers(4, 12)), 3, eps_frac=0.0) for i in range(5)}
    clats = {k: api_mod.determinize_lattice_pruned(L, 1e9) for k, L in raws.items()}
    with gzip.open("lat.1.gz", "wb") as f:
        for k, c in clats.items():
            f.write(k.encode() + b" \0B")
            kio.write_compact_lattice(f, c, True)
    with kio.TableWriter("raw.ark", kind="lattice") as w:
        for k, L in raws.items():
            L = dict(L)
            L["state_frame"] = np.zeros(len(L["state_final"]), np.int32)
            w.write(k, L)
    acwt, lmwt = 0.5, 1.5
    assert tool.main(["--acoustic-scale=%g" % acwt, "--lm-scale=%g" % lmwt, "ark:gunzip -c lat.1.gz|", "ark:1.post", "ark,t:1.like"]) == 0
    assert tool.main(["--acoustic-scale=%g" % acwt, "--lm-scale=%g" % lmwt, "--print-args=false", "ark:raw.ark", "ark,t:raw.post"]) == 0
    assert tool.main(["--acoustic-scale=0", "ark:raw.ark", "ark:x.post"]) == 255
    posts = dict(pkg("kaldi_cli").SequentialTableReader("ark:1.post", "posterior"))
    posts_raw = dict(pkg("kaldi_cli").SequentialTableReader("ark,t:raw.post", "posterior"))
    likes = {l.split()[0]: float(l.split()[1]) for l in open("1.like")}
    for k in raws:
        for src, got in ((kio.compact_lattice_to_lattice(clats[k]), posts[k]), (dict(raws[k], num_states=len(raws[k]["state_final"])), posts_raw[k])):
            csr = tool.top_sorted_csr(src, lmwt, acwt)
            want = binding.lattice_forward_backward(csr)
            # the Posterior the reference builds from the arc posteriors (lattice-functions.cc:333-352): frame = the time of
            # the arc's source state, pairs sorted by transition-id and merged
            arc_src = np.repeat(np.arange(csr["n_states"]), np.diff(csr["arc_offsets"]))
            T = int(want["state_times"].max())
            want_post = [dict() for _ in range(T)]
            for j, tid in enumerate(csr["arc_ilabel"].tolist()):
                if tid != 0:
                    d = want_post[int(want["state_times"][arc_src[j]])]
                    d[tid] = float(np.float32(d.get(tid, 0.0)) + np.float32(want["arc_post"][j]))
            assert len(got) == T
            for a, b in zip(got, want_post):
                b = {t: w for t, w in b.items() if np.float32(w) != 0.0}      # MergePairVectorSumming drops zero entries (util/stl-utils.h)
                assert [t for t, _ in a] == sorted(b)
                np.testing.assert_allclose([w for _, w in a], [b[t] for t in sorted(b)], atol=2e-6)
        assert abs(likes[k] - binding.lattice_forward_backward(tool.top_sorted_csr(kio.compact_lattice_to_lattice(clats[k]), lmwt, acwt))["tot_like"]) < 1e-4


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
        oc = binding.DecoderOracle(g, cfg, "reference")
        oc.decode(ll)
        want, best = oc.raw_lattice(), oc.best_path()
        got = lats[k]
        order = np.argsort(want["arc_src"], kind="stable")
        assert got["num_states"] == len(want["state_frame"])
        for key in ("arc_dst", "arc_il", "arc_ol"):
            assert np.array_equal(got[key], want[key][order]), (k, key)
        np.testing.assert_allclose(got["arc_a"] * acwt, want["arc_a"][order], atol=2e-4)
        assert np.array_equal(words[k], best["words"])


def test_cfg1_yesno_mono_gmm_at_its_stated_shape(api, oracle, tmp_path, monkeypatch):
    """BASELINE config 1 at the shape SURVEY.md 8(d)-1 states: monophone DiagGmm system, 39-dim features (13 MFCC +
    delta + delta-delta), 400 Gaussians in total (egs/yesno/s5/run.sh:32-34 --totgauss 400) over 12 pdfs (4 phones x
    3 HMM states), 30 utterances x 600 frames, a yes/no word-loop graph (epsilon arcs out of the start state); through
    gmm-latgen-faster's command line with the binary's default determinized output.  Every utterance against the
    oracle chain: words and alignment of the best path, the determinized CompactLattice equivalent (every word
    sequence, its cost and alignment) to the determinized oracle lattice."""
    import lattice_equiv as LE
    from oracle import binding
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gmm_latgen_faster as tool
    rng = np.random.default_rng(1001)
    D, n_pdf, acwt = 39, 12, 0.1
    am = workloads.make_am_gmm(rng, n_pdf, 400, D)
    mi, iv = workloads.gmm_inv_params(am)
    g, topo, triples, log_probs = workloads.make_word_loop_graph([0, 1, 2, 0], hmm_states=3)   # SIL, YES, NO, a noise phone
    monkeypatch.chdir(tmp_path)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        kio.write_am_diag_gmm(f, dict(weights=am["weights"], means_invvars=mi, inv_vars=iv, pdf_offsets=am["pdf_offsets"], dim=D), True)
    with open("HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    # features: a random phone sequence, every frame drawn from one Gaussian of its pdf
    utts = {}
    for i in range(30):
        pdfs = []
        while len(pdfs) < 600:
            p = int(rng.integers(0, 4))
            for k in range(3):
                pdfs += [3 * p + k] * int(rng.integers(2, 12))
        pdfs = np.array(pdfs[:600])
        comp = am["pdf_offsets"][pdfs] + (rng.random(600) * np.diff(am["pdf_offsets"])[pdfs]).astype(np.int64)
        utts["yesno_%02d" % i] = (am["means"][comp] + rng.standard_normal((600, D)) * np.sqrt(am["vars"][comp]) * 1.5).astype(np.float32)
    with kio.TableWriter("feats.ark") as w:
        for k, m in utts.items():
            w.write(k, m)
    # egs/yesno/s5/run.sh decodes with steps/decode.sh defaults: beam 13, lattice-beam 6, max-active 7000, acwt 0.083333 (0.1 here)
    opts = ["--beam=13", "--max-active=7000", "--lattice-beam=6", "--acoustic-scale=%g" % acwt, "--allow-partial=true"]
    assert tool.main(opts + ["final.mdl", "HCLG.fst", "ark:feats.ark", "ark:lat.ark", "ark:words.ark", "ark:ali.ark"]) == 0
    clats = dict(kio.read_ark("lat.ark", kind="compact_lattice"))
    words = dict(kio.read_ark("words.ark", kind="int32_vector"))
    alis = dict(kio.read_ark("ali.ark", kind="int32_vector"))
    assert sorted(clats) == sorted(utts)
    gconsts, _ = oracle.gmm_compute_gconsts(am["weights"], mi, iv)
    cfg = binding.decoder_config(beam=13.0, max_active=7000, lattice_beam=6.0)
    tid_phone = api.tid_phone_map(kio.read_gmm_model("final.mdl")[0])
    assert int((tid_phone != 0).sum()) == 4      # one phone-initial forward transition per phone
    n_words = 0
    for k, x in utts.items():
        ll = (oracle.am_gmm_loglikes(x, gconsts, mi, iv, am["pdf_offsets"], -1.0) * np.float32(acwt)).astype(np.float32)
        oc = binding.DecoderOracle(g, cfg, "reference")
        assert oc.decode(ll)
        best = oc.best_path()
        assert np.array_equal(words[k], best["words"]) and np.array_equal(alis[k], best["alignment"]), k
        assert len(alis[k]) == 600
        n_words += len(words[k])
        # (the binary's det_opts defaults: phone + word passes; the oracle is the line-by-line restatement of the reference's)
        want = binding.determinize_lattice_phone_pruned(oc.raw_lattice(), 6.0, tid_phone)
        got = dict(clats[k])
        got["arc_a"] = got["arc_a"] * np.float32(acwt)      # the file holds unscaled acoustic costs
        got["final_a"] = got["final_a"] * np.float32(acwt)
        res = LE.compare_deterministic(got, want, delta=2e-2)   # (frame log-likelihoods to 1e-4 x 600 frames)
        assert LE.deterministic_equal(res), (k, res)
    assert n_words > 30 * 10


def test_online2_wav_nnet2_latgen_faster_files_in_files_out(api, oracle, tmp_path, monkeypatch):
    """tools/online2_wav_nnet2_latgen_faster.py --online=false: wave files + the online2 configuration
    files in, CompactLattices out; checked against the chain of oracles (MFCC, iVector in the
    use_most_recent + greedy mode with the adaptation state carried from a speaker's utterance to the
    next and LimitFrames in between, network, reference-order decoder) on the same waveforms."""
    from oracle import binding
    from oracle import ivector_oracle as IO
    from test_feature_oracle import wave
    from test_gpu_online2_pipeline import compact_best_path
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    tool = importlib.import_module("tools.online2_wav_nnet2_latgen_faster")
    net, priors = make_golden.kaldi_io_net(np.random.default_rng(12))     # 6-dim input: 4 MFCCs + 2 iVector dims (the constant part)
    n_pdf, acwt = 5, 0.2
    rng = np.random.default_rng(22)
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf,
                entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    pdf_of_phone = rng.permutation(n_pdf)
    triples = [(p + 1, 0, int(pdf_of_phone[p])) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    g = workloads.make_hclg_like(rng, 400, n_pdf, final_frac=0.2)
    g["tid2pdf"] = np.concatenate([[-1], np.repeat(pdf_of_phone, 2)]).astype(np.int32)
    ie = workloads.make_ivector_extractor(rng, base_dim=4, splice=1, feat_dim=5, num_gauss=6, ivector_dim=2, prior_offset=3.0)
    ie.update(greedy_most_recent=True, ivector_period=10, max_count=0.0)
    mfcc_kw = dict(num_bins=10, num_ceps=4, low_freq=20.0, high_freq=0.0)
    monkeypatch.chdir(tmp_path)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_nnet_body_bin"), "rb").read())
    with open("HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    inv = 1.0 / ie["ubm_vars"].astype(np.float64)
    for name, writer in (("final.mat", lambda f: kio.write_matrix(f, ie["lda_mat"])),
                         ("global_cmvn.stats", lambda f: kio.write_matrix(f, np.asarray(ie["global_cmvn_stats"], np.float64))),
                         ("final.dubm", lambda f: kio.write_diag_gmm(f, ie["ubm_weights"], (ie["ubm_means"] * inv).astype(np.float32),
                                                                     inv.astype(np.float32))),
                         ("final.ie", lambda f: kio.write_ivector_extractor(
                             f, dict(w=np.zeros((0, 0)), w_vec=np.log(ie["ubm_weights"].astype(np.float64)), M=ie["M"],
                                     Sigma_inv=ie["Sigma_inv"], prior_offset=ie["prior_offset"])))):
        with open(name, "wb") as f:
            f.write(b"\0B")
            writer(f)
    open("mfcc.conf", "w").write("--use-energy=false   # only non-default options\n--num-mel-bins=10\n--num-ceps=4\n--dither=0\n")
    open("splice.conf", "w").write("--left-context=1\n--right-context=1\n")
    open("online_cmvn.conf", "w").write("# defaults\n")
    open("ivector_extractor.conf", "w").write(
        "--splice-config=splice.conf\n--cmvn-config=online_cmvn.conf\n--lda-matrix=final.mat\n--global-cmvn-stats=global_cmvn.stats\n"
        "--diag-ubm=final.dubm\n--ivector-extractor=final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n"
        "--max-remembered-frames=5\n--max-count=0\n")
    open("online_nnet2_decoding.conf", "w").write(
        "--feature-type=mfcc\n--mfcc-config=mfcc.conf\n--ivector-extraction-config=ivector_extractor.conf\n"
        "--beam=9\n--max-active=300\n--lattice-beam=5\n--acoustic-scale=%g\n" % acwt)
    waves = {"utt%d" % i: np.trunc(wave(30 + i, n)) for i, n in enumerate((16000, 6400, 8000))}
    with open("wav.scp", "w") as f:
        for k, w in waves.items():
            with open(k + ".wav", "wb") as wf:
                kio.write_wave(wf, 16000.0, w)
            f.write("%s %s.wav\n" % (k, k))
        f.write("missing nowhere.wav\n")
    open("spk2utt", "w").write("spkA utt0 utt1\nutt2 utt2\nghost ghost\n")
    assert tool.main(["--config=online_nnet2_decoding.conf", "--online=false", "final.mdl", "HCLG.fst", "ark:spk2utt", "scp:wav.scp",
                      "ark:clat.ark"]) == 0
    clats = dict(kio.read_ark("clat.ark", kind="compact_lattice"))
    assert sorted(clats) == sorted(waves)
    ko = binding.OracleLib("ko")
    cfg = binding.decoder_config(beam=9.0, max_active=300, lattice_beam=5.0)
    spk_state, different = {"spkA": None, "utt2": None}, 0.0
    for k, w in waves.items():
        m = ko.mfcc_compute(w.astype(np.float32), **mfcc_kw)
        spk = "spkA" if k in ("utt0", "utt1") else k
        iv, st = IO.extract(m, ie, spk_state[spk], True)
        IO.limit_frames(st, ie, 5.0)
        spk_state[spk] = st
        if k == "utt1":                     # the carried state matters: from the prior the rows differ
            different = np.abs(iv - IO.extract(m, ie)).max()
        x = np.concatenate([m, iv], 1)
        ll = oracle.decodable_am_nnet(net, priors, acwt, x)
        oc = binding.DecoderOracle(g, cfg, "reference")
        assert oc.decode(ll)
        best = oc.best_path()
        C = clats[k]
        cl = dict(arc_src=C["arc_src"], arc_dst=C["arc_dst"], arc_label=C["arc_label"], arc_g=C["arc_g"], arc_a=C["arc_a"] * acwt,
                  arc_string=C["arc_string"], final_g=C["final_g"], final_a=C["final_a"] * acwt, final_string=C["final_string"])
        words, ali, cost = compact_best_path(cl)
        assert words == [int(v) for v in best["words"]] and ali == [int(v) for v in best["alignment"]], k
        assert abs(cost - (best["graph_cost"] + best["acoustic_cost"])) < 5e-3
    assert different > 0.05


def online2_setup(tmp_path, monkeypatch, greedy):
    """The files of an online2 decoding directory (tiny model); returns what the oracle chains need."""
    from test_feature_oracle import wave
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    net, priors = make_golden.kaldi_io_net(np.random.default_rng(12))     # 6-dim input: 4 MFCCs + 2 iVector dims (the constant part)
    n_pdf, acwt = 5, 0.2
    rng = np.random.default_rng(22)
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf, entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    pdf_of_phone = rng.permutation(n_pdf)
    triples = [(p + 1, 0, int(pdf_of_phone[p])) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    g = workloads.make_hclg_like(rng, 400, n_pdf, final_frac=0.2)
    g["tid2pdf"] = np.concatenate([[-1], np.repeat(pdf_of_phone, 2)]).astype(np.int32)
    ie = workloads.make_ivector_extractor(rng, base_dim=4, splice=1, feat_dim=5, num_gauss=6, ivector_dim=2, prior_offset=3.0)
    ie.update(greedy_most_recent=greedy, ivector_period=10, max_count=0.0)
    monkeypatch.chdir(tmp_path)
    with open("final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_nnet_body_bin"), "rb").read())
    with open("HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    inv = 1.0 / ie["ubm_vars"].astype(np.float64)
    for name, writer in (("final.mat", lambda f: kio.write_matrix(f, ie["lda_mat"])),
                         ("global_cmvn.stats", lambda f: kio.write_matrix(f, np.asarray(ie["global_cmvn_stats"], np.float64))),
                         ("final.dubm", lambda f: kio.write_diag_gmm(f, ie["ubm_weights"], (ie["ubm_means"] * inv).astype(np.float32),
                                                                     inv.astype(np.float32))),
                         ("final.ie", lambda f: kio.write_ivector_extractor(
                             f, dict(w=np.zeros((0, 0)), w_vec=np.log(ie["ubm_weights"].astype(np.float64)), M=ie["M"],
                                     Sigma_inv=ie["Sigma_inv"], prior_offset=ie["prior_offset"])))):
        with open(name, "wb") as f:
            f.write(b"\0B")
            writer(f)
    open("mfcc.conf", "w").write("--use-energy=false   # only non-default options\n--num-mel-bins=10\n--num-ceps=4\n--dither=0\n")
    open("splice.conf", "w").write("--left-context=1\n--right-context=1\n")
    open("online_cmvn.conf", "w").write("# defaults\n")
    open("ivector_extractor.conf", "w").write(
        "--splice-config=splice.conf\n--cmvn-config=online_cmvn.conf\n--lda-matrix=final.mat\n--global-cmvn-stats=global_cmvn.stats\n"
        "--diag-ubm=final.dubm\n--ivector-extractor=final.ie\n--num-gselect=5\n--min-post=0.025\n--posterior-scale=0.1\n"
        "--max-remembered-frames=5\n--max-count=0\n")
    open("online_nnet2_decoding.conf", "w").write(
        "--feature-type=mfcc\n--mfcc-config=mfcc.conf\n--ivector-extraction-config=ivector_extractor.conf\n"
        "--beam=9\n--max-active=300\n--lattice-beam=5\n--acoustic-scale=%g\n" % acwt)
    waves = {"utt%d" % i: np.trunc(wave(30 + i, n)) for i, n in enumerate((16000, 6400, 8000))}
    with open("wav.scp", "w") as f:
        for k, w in waves.items():
            with open(k + ".wav", "wb") as wf:
                kio.write_wave(wf, 16000.0, w)
            f.write("%s cat %s.wav |\n" % (k, k) if k == "utt1" else "%s %s.wav\n" % (k, k))    # one entry is a command
    open("spk2utt", "w").write("spkA utt0 utt1\nutt2 utt2\n")
    with open("words.txt", "w") as f:
        f.write("<eps> 0\n" + "".join("W%d %d\n" % (i, i) for i in range(1, int(g["olabel"].max()) + 1)))
    return dict(net=net, priors=priors, g=g, ie=ie, waves=waves, acwt=acwt, n_pdf=n_pdf, pdf_of_phone=pdf_of_phone,
                mfcc_kw=dict(num_bins=10, num_ceps=4, low_freq=20.0, high_freq=0.0))


def test_online2_wav_nnet2_latgen_faster_online_true_with_endpointing(api, oracle, tmp_path, monkeypatch):
    """steps/online/nnet2/decode.sh:118-125 in the binary's DEFAULT mode, --online=true: iVectors estimated online (per
    period, from the frames so far), the waveform consumed in --chunk-length chunks, --do-endpointing with the rules of
    online2/online-endpoint.cc tested after every chunk.  Against an oracle CHAIN that really runs chunk by chunk: the
    numpy iVector specification in its online mode, the network oracle, the decoder oracle's InitDecoding / AdvanceDecoding
    sequence fed exactly the frames DecodableNnet2Online::NumFramesReady() reports after each chunk (restated here from
    online-feature.cc / online-nnet2-decodable.cc:69-83), best-path traceback + FinalRelativeCost() -> the five rules,
    FinalizeDecoding, and the reference's determinization restated.  Same endpoint chunk, same lattice."""
    import gzip
    import lattice_equiv as LE
    from oracle import binding
    from oracle import ivector_oracle as IO
    kio = pkg("kaldi_io")
    S = online2_setup(tmp_path, monkeypatch, greedy=False)
    tool = importlib.import_module("tools.online2_wav_nnet2_latgen_faster")
    acwt, g, ie = S["acwt"], S["g"], S["ie"]
    phone_of_pdf = np.empty(S["n_pdf"], np.int64)
    phone_of_pdf[S["pdf_of_phone"]] = np.arange(1, S["n_pdf"] + 1)
    sil = {1, 2, 3}
    common = ["--config=online_nnet2_decoding.conf", "--word-symbol-table=words.txt", "final.mdl", "HCLG.fst", "ark:spk2utt", "scp:wav.scp"]
    # (1) no endpointing, default chunk length; (2) endpointing: a 5-frame trailing-silence rule and a 0.3 s length rule
    assert tool.main(["--online=true"] + common + ["ark:|gzip -c > lat.1.gz"]) == 0
    ep = ["--do-endpointing=true", "--endpoint.silence-phones=1:2:3", "--endpoint.rule1.min-trailing-silence=0.05",
          "--endpoint.rule2.min-trailing-silence=100", "--endpoint.rule3.min-trailing-silence=100", "--endpoint.rule4.min-trailing-silence=100",
          "--endpoint.rule5.min-utterance-length=0.6", "--chunk-length=0.07"]
    assert tool.main(ep + common + ["ark:ep.ark"]) == 0
    plain = dict(kio.read_ark(gzip.open("lat.1.gz"), kind="compact_lattice"))
    endp = dict(kio.read_ark("ep.ark", kind="compact_lattice"))
    assert sorted(plain) == sorted(S["waves"]) == sorted(endp)

    ko = binding.OracleLib("ko")
    cfg = binding.decoder_config(beam=9.0, max_active=300, lattice_beam=5.0)
    tm = kio.read_nnet2_model("final.mdl")[0]
    tp = api.tid_phone_map(tm)
    rc = max(max(c["context"]) for c in S["net"] if c["type"] == "splice")     # the network's right context

    def ready(n, finished):      # OnlineMfcc -> OnlineSpliceFrames (right context 1) -> OnlineAppendFeature -> DecodableNnet2Online
        base = 0 if n < 400 else 1 + (n - 400) // 160
        feats = base if finished else max(0, base - 1)
        if feats == 0:
            return 0
        return feats if finished else max(0, feats - rc)

    def rules(num_frames, trailing, rel_cost, r1_sil, r5_len):
        utt_len, trail = np.float32(num_frames) * np.float32(0.01), np.float32(trailing) * np.float32(0.01)
        rule1 = trail >= np.float32(r1_sil)                                       # must_contain_nonsilence = false, any cost
        rule5 = utt_len >= np.float32(r5_len)
        return bool(rule1 or rule5)

    n_stopped = 0
    for mode, got_all, chunk, do_ep in (("plain", plain, 800, False), ("endpoint", endp, int(16000 * 0.07), True)):
        spk_state = {"spkA": None, "utt2": None}
        for k, w in S["waves"].items():
            spk = "spkA" if k in ("utt0", "utt1") else k
            n = len(w)
            offs = list(range(chunk, n, chunk)) + [n]
            m_full = ko.mfcc_compute(w.astype(np.float32), **S["mfcc_kw"])
            iv_full, _ = IO.extract(m_full, ie, spk_state[spk], True)
            ll_full = oracle.decodable_am_nnet(S["net"], S["priors"], acwt, np.concatenate([m_full, iv_full], 1))
            od = binding.DecoderOracle(g, cfg, "reference")
            od.begin(ll_full)
            decoded, stop_at = 0, None
            for ci, o in enumerate(offs):
                want = ready(o, o == n)
                if want > decoded:
                    decoded = od.advance(want - decoded)
                    assert decoded == want
                if do_ep and decoded > 0:
                    od.snapshot(use_final_probs=False)
                    ali = od.best_path()["alignment"]
                    trailing = 0
                    for tid in ali[::-1]:
                        if int(phone_of_pdf[g["tid2pdf"][tid]]) in sil:
                            trailing += 1
                        else:
                            break
                    if rules(decoded, trailing, od.final_relative_cost(), 0.05, 0.6):
                        stop_at = ci
                        break
            od.finalize()
            od.snapshot(True)
            n_stopped += stop_at is not None
            # the speaker's adaptation state: from what the pipeline had consumed (the whole waveform unless endpointed)
            w_used = w if stop_at is None else w[:offs[stop_at]]
            m_used = ko.mfcc_compute(w_used.astype(np.float32), **S["mfcc_kw"])
            _, st = IO.extract(m_used, ie, spk_state[spk], True)
            IO.limit_frames(st, ie, 5.0)
            spk_state[spk] = st
            want_c = binding.determinize_lattice_phone_pruned(od.raw_lattice(), 5.0, tp)
            got = dict(got_all[k])
            got["arc_a"] = got["arc_a"] * np.float32(acwt)
            got["final_a"] = got["final_a"] * np.float32(acwt)
            assert sum(len(x) for x in got["arc_string"]) + sum(len(x) for x in got["final_string"]) >= 0
            # costs: the iVector rows agree with the numpy specification to 2e-4, the frame log-likelihoods to 1e-4 (north_star):
            # a few 1e-3 over a path of this length; states, words and alignments exactly
            res = LE.compare_deterministic(got, want_c, delta=3e-2, strings=False)
            assert got["n_states"] == want_c["n_states"] and LE.deterministic_equal(res), (mode, k, {a: b for a, b in res.items() if b}, decoded)
            # alignments: the best path's exactly; the others wherever two alignments of a word sequence are not tied to within
            # the float32 noise of the feature pipeline (this 5-pdf model has many near-ties)
            from test_gpu_online2_pipeline import compact_best_path
            best = od.best_path()
            words, ali, cost = compact_best_path(got)
            assert words == [int(v) for v in best["words"]] and ali == [int(v) for v in best["alignment"]], (mode, k)
            assert abs(cost - (best["graph_cost"] + best["acoustic_cost"])) < 1e-2
            # the number of frames in the lattice = the frames decoded when the utterance stopped
            wl = LE.WordLattice.from_compact(got)
            istr, _, _ = LE.rand_path(wl, np.random.default_rng(0))
            assert len(istr) == decoded, (mode, k, len(istr), decoded)
    assert n_stopped >= 1       # the endpointing run did stop an utterance early (and the lattices above have that many frames)


@pytest.mark.parametrize("endpointing", [False, True])
def test_online2_silence_weighting_of_the_ivector_statistics(api, oracle, tmp_path, monkeypatch, endpointing):
    """--ivector-silence-weighting.* (egs/librispeech/s5/local/online/run_nnet2_ms.sh:216 -> steps/online/nnet2/decode.sh:101-104):
    per chunk the decoder's traceback re-weights the frames in the iVector statistics (online2-wav-nnet2-latgen-faster.cc:239-244),
    so the iVector of a period, the network's input and the search depend on one another chunk by chunk.  The oracle chain runs
    exactly that loop per utterance: numpy OnlineIvectorFeature with UpdateFrameWeights, OnlineSilenceWeighting restated line by
    line, the network oracle on the rows that became ready, the decoder oracle's AdvanceDecoding; speaker state chained
    (GetAdaptationState with the weighted statistics).  Same lattices; with endpointing the same stopping chunk."""
    import lattice_equiv as LE
    from oracle import binding
    from oracle import ivector_oracle as IO
    from test_gpu_online2_pipeline import compact_best_path
    kio = pkg("kaldi_io")
    S = online2_setup(tmp_path, monkeypatch, greedy=False)
    tool = importlib.import_module("tools.online2_wav_nnet2_latgen_faster")
    acwt, g, ie = S["acwt"], S["g"], S["ie"]
    sw, msd, chunk_secs = 0.25, 4, 0.07
    common = ["--config=online_nnet2_decoding.conf", "final.mdl", "HCLG.fst", "ark:spk2utt", "scp:wav.scp"]
    opts = ["--ivector-silence-weighting.silence-phones=1:2:3", "--ivector-silence-weighting.silence-weight=%g" % sw,
            "--ivector-silence-weighting.max-state-duration=%d" % msd, "--chunk-length=%g" % chunk_secs]
    if endpointing:
        opts += ["--do-endpointing=true", "--endpoint.silence-phones=1:2:3", "--endpoint.rule1.min-trailing-silence=0.05",
                 "--endpoint.rule2.min-trailing-silence=100", "--endpoint.rule3.min-trailing-silence=100",
                 "--endpoint.rule4.min-trailing-silence=100", "--endpoint.rule5.min-utterance-length=0.6"]
    assert tool.main(opts + common + ["ark:sw.ark"]) == 0
    # needs the online estimation: refused with --online=false
    assert tool.main(["--online=false"] + opts + common + ["ark:x.ark"]) == 255
    got_all = dict(kio.read_ark("sw.ark", kind="compact_lattice"))
    assert sorted(got_all) == sorted(S["waves"])

    ko = binding.OracleLib("ko")
    cfg = binding.decoder_config(beam=9.0, max_active=300, lattice_beam=5.0)
    tm = kio.read_nnet2_model("final.mdl")[0]
    tp = api.tid_phone_map(tm)
    t2ph = tm["tid2phone"]
    ctx = [c["context"] for c in S["net"] if c["type"] == "splice"]
    L, R = -min(min(c) for c in ctx), max(max(c) for c in ctx)
    sil = {1, 2, 3}
    chunk = int(16000 * chunk_secs)
    spk_state = {"spkA": None, "utt2": None}
    n_stopped = 0
    n_downweighted = 0
    for k, w in S["waves"].items():
        spk = "spkA" if k in ("utt0", "utt1") else k
        n = len(w)
        offs = list(range(chunk, n, chunk)) + [n]
        m_full = ko.mfcc_compute(w.astype(np.float32), **S["mfcc_kw"])
        T = len(m_full)
        feat = IO.OnlineIvectorFeature(m_full, ie, spk_state[spk])
        swo = IO.OnlineSilenceWeighting(t2ph, sil, sw, msd)
        rows = np.zeros((T, m_full.shape[1] + 2), np.float32)
        rows[:, :m_full.shape[1]] = m_full
        ll = np.zeros((T, S["n_pdf"]), np.float32)
        od = binding.DecoderOracle(g, cfg, "reference")
        od.begin(ll)
        ll = od._ll                                    # the rows the decoder reads: filled as they are computed
        decoded, filled, stop_at, base_used = 0, 0, None, T
        for ci, o in enumerate(offs):
            fin = o == n
            base = 0 if o < 400 else 1 + (o - 400) // 160
            feat_ready = base if fin else max(0, base - 1)                       # OnlineSpliceFrames of the iVector: right context 1
            want = 0 if feat_ready == 0 else (feat_ready if fin else max(0, feat_ready - R))
            if decoded > 0:
                od.snapshot(use_final_probs=False)
                ali = od.best_path()["alignment"]
            else:
                ali = []
            swo.compute_current_traceback(ali)
            feat.update_frame_weights(swo.get_delta_weights(feat_ready), feat_ready)
            if want > decoded:
                for t in range(filled, feat_ready):
                    rows[t, m_full.shape[1]:] = feat.get_frame(t)
                filled = feat_ready
                idx = np.clip(np.arange(decoded - L, want + R), 0, feat_ready - 1)
                out = oracle.decodable_am_nnet(S["net"], S["priors"], acwt, rows[idx])
                ll[decoded:want] = out[L:L + want - decoded]
                decoded = od.advance(want - decoded)
                assert decoded == want
            if endpointing and decoded > 0:
                od.snapshot(use_final_probs=False)
                ali = od.best_path()["alignment"]
                trailing = 0
                for tid in ali[::-1]:
                    if int(t2ph[tid]) in sil:
                        trailing += 1
                    else:
                        break
                utt_len, trail = np.float32(decoded) * np.float32(0.01), np.float32(trailing) * np.float32(0.01)
                if trail >= np.float32(0.05) or utt_len >= np.float32(0.6):
                    stop_at, base_used = ci, base
                    break
        od.finalize()
        od.snapshot(True)
        n_stopped += stop_at is not None
        n_downweighted += int(sum(1 for x in swo.weight if x != 1.0))
        # GetAdaptationState: CMVN over the MFCC frames accepted, the weighted iVector statistics as they stand; LimitFrames
        _, st = IO.extract(m_full[:max(1, base_used)], ie, spk_state[spk], True)
        st.update(quad=feat.quad.copy(), lin=feat.lin.copy(), num_frames=feat.num_frames)
        IO.limit_frames(st, ie, 5.0)
        spk_state[spk] = st
        want_c = binding.determinize_lattice_phone_pruned(od.raw_lattice(), 5.0, tp)
        got = dict(got_all[k])
        got["arc_a"] = got["arc_a"] * np.float32(acwt)
        got["final_a"] = got["final_a"] * np.float32(acwt)
        res = LE.compare_deterministic(got, want_c, delta=3e-2, strings=False)
        assert got["n_states"] == want_c["n_states"] and LE.deterministic_equal(res), (k, {a: b for a, b in res.items() if b}, decoded)
        best = od.best_path()
        words, ali, cost = compact_best_path(got)
        assert words == [int(v) for v in best["words"]] and ali == [int(v) for v in best["alignment"]], k
        assert abs(cost - (best["graph_cost"] + best["acoustic_cost"])) < 1e-2
        wl = LE.WordLattice.from_compact(got)
        istr, _, _ = LE.rand_path(wl, np.random.default_rng(0))
        assert len(istr) == decoded, (k, len(istr), decoded)
    assert n_downweighted > 20                       # the weighting did something in this run
    assert (n_stopped >= 1) == endpointing


def test_reference_named_executables_on_path_and_a_table_of_graphs(api, oracle, tmp_path):
    """SURVEY 8 f4 leftovers: (1) `nnet-latgen-faster` BY NAME, found on PATH (bin/), with the literal argv of
    steps/nnet2/decode.sh:130-136 - nothing of the recipe line changes but PATH; (2) the fsts-rspecifier branch
    (nnet-latgen-faster.cc:140-176): a table of per-utterance decoding graphs, a new decoder per utterance, features looked
    up by the graph's key, an utterance without features counted as failed."""
    import gzip
    import subprocess
    from oracle import binding
    kio, workloads = pkg("kaldi_io"), pkg("workloads")
    cli = pkg("kaldi_cli")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    net, priors = make_golden.kaldi_io_net(np.random.default_rng(12))
    n_pdf, acwt = 5, 0.2
    rng = np.random.default_rng(31)
    topo = dict(phones=list(range(1, n_pdf + 1)), phone2idx=[-1] + [0] * n_pdf, entries=[[(0, [(0, 0.5), (1, 0.5)]), (-1, [])]])
    triples = [(p + 1, 0, p) for p in range(n_pdf)]
    log_probs = np.concatenate([[0.0], np.full(2 * n_pdf, np.log(0.5))]).astype(np.float32)
    tid2pdf = np.concatenate([[-1], np.repeat(np.arange(n_pdf), 2)]).astype(np.int32)
    g = workloads.make_hclg_like(rng, 300, n_pdf, final_frac=0.2)
    os.makedirs(tmp_path / "graph")
    os.makedirs(tmp_path / "exp" / "decode")
    with open(tmp_path / "final.mdl", "wb") as f:
        f.write(b"\0B")
        kio.write_transition_model(f, topo, triples, log_probs, True)
        f.write(open(os.path.join(GOLD, "am_nnet_body_bin"), "rb").read())
    with open(tmp_path / "graph" / "HCLG.fst", "wb") as f:
        kio.write_fst(f, g)
    n_words = int(g["olabel"].max())
    with open(tmp_path / "graph" / "words.txt", "w") as f:
        f.write("<eps> 0\n" + "".join("W%d %d\n" % (i, i) for i in range(1, n_words + 1)))
    utts = {"spk1-utt%d" % i: rng.standard_normal((T, 6)).astype(np.float32) for i, T in enumerate((33, 12, 50))}
    with kio.TableWriter(str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")) as w:
        for k, m in utts.items():
            w.write(k, m)
    env = dict(os.environ, PATH=os.path.join(ROOT, "bin") + os.pathsep + os.environ["PATH"], PYTHON=sys.executable)
    # (1) decode.sh:130-136 as the recipe writes it (JOB = 1, $thread_string empty)
    line = ("nnet-latgen-faster --minimize=false --max-active=300 --min-active=200 --beam=9 --lattice-beam=5 "
            "--acoustic-scale=%g --allow-partial=true --word-symbol-table=graph/words.txt final.mdl graph/HCLG.fst "
            "\"ark,s,cs:cat feats.ark |\" \"ark:|gzip -c > exp/decode/lat.1.gz\"" % acwt)
    p = subprocess.run(line, shell=True, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "LOG (nnet-latgen-faster:main()) Done 3 utterances, failed for 0" in p.stderr
    clats = dict(kio.read_ark(gzip.open(tmp_path / "exp" / "decode" / "lat.1.gz"), kind="compact_lattice"))
    assert sorted(clats) == sorted(utts)
    for name in ("gmm-latgen-faster", "online2-wav-nnet2-latgen-faster", "lattice-to-post"):   # the others resolve and print their usage
        q = subprocess.run([name], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
        assert q.returncode == 1 and "Usage" in q.stderr, (name, q.stderr[-500:])
    # (2) a table of graphs: each utterance its own (here: differently seeded) graph; one key has no features
    graphs = {k: workloads.make_hclg_like(np.random.default_rng(100 + i), 120 + 40 * i, n_pdf, final_frac=0.3) for i, k in enumerate(utts)}
    graphs["spk9-missing"] = graphs["spk1-utt0"]
    w = cli.TableWriter("ark:" + str(tmp_path / "graphs.fsts"), "fst")
    for k, gg in graphs.items():
        w.write(k, gg)
    assert w.close()
    line = ("nnet-latgen-faster --determinize-lattice=false --beam=9 --lattice-beam=5 --acoustic-scale=%g --allow-partial=true final.mdl "
            "ark:graphs.fsts scp:feats.scp ark:raw.ark ark,t:words.txt ark,t:ali.txt" % acwt)
    p = subprocess.run(line, shell=True, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "Not decoding utterance spk9-missing because no features available." in p.stderr
    assert "Done 3 utterances, failed for 1" in p.stderr
    lats = dict(kio.read_ark(str(tmp_path / "raw.ark"), kind="lattice"))
    ali = dict(kio.read_ark(str(tmp_path / "ali.txt"), kind="int32_vector"))
    cfg = binding.decoder_config(beam=9.0, lattice_beam=5.0)
    nnet_oracle = oracle
    for k, m in utts.items():
        gg = dict(graphs[k], tid2pdf=tid2pdf)
        ll = nnet_oracle.decodable_am_nnet(net, priors, acwt, m)
        od = binding.DecoderOracle(gg, cfg, "reference")
        assert od.decode(ll)
        want = od.best_path()
        assert np.array_equal(ali[k], want["alignment"]), k
        assert len(lats[k]["arc_src"]) == len(od.raw_lattice()["arc_src"]), k
