#!/usr/bin/env python3
"""nnet-latgen-faster / gmm-latgen-faster on the MI355X path: the reference binaries'
command line (nnet2bin/nnet-latgen-faster.cc:40-190, gmmbin/gmm-latgen-faster.cc:36-180)
over the library, reading and writing the reference's own file formats
(old-kaldi-git_amd/kaldi_io.py).  Run through nnet_latgen_faster.py / gmm_latgen_faster.py.

  {nnet,gmm}_latgen_faster.py [options] <model-in> <fst-in> <features-rspecifier> \\
      <lattice-wspecifier> [<words-wspecifier> [<alignments-wspecifier>]]

  <model-in>              final.mdl (TransitionModel + AmNnet | AmDiagGmm), binary or text
  <fst-in>                HCLG.fst (OpenFst vector or const, StdArc)
  <features-rspecifier>   ark:FILE | scp:FILE
  <lattice-wspecifier>    ark:FILE | ark,t:FILE   CompactLattices (--determinize-lattice=true, the
                          binaries' default: DeterminizeLatticePhonePrunedWrapper, decoder-wrappers.cc:264-274,
                          on host threads) or state-level lattices (--determinize-lattice=false)
  words / alignments      ark:FILE | ark,t:FILE   Int32Vector tables

Differences from the binary: utterances are decoded in batches (--batch-frames) — one
forward pass and one decoder launch per batch — instead of one at a time; the results per
utterance are the same (tests/test_gpu_latgen_tool.py).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def parse_specifier(spec, writing):
    """ClassifyWspecifier / ClassifyRspecifier (util/kaldi-table.cc) for the forms supported here."""
    opts, _, path = spec.partition(":")
    opts = opts.split(",")
    kind = opts[0]
    if kind not in ("ark", "scp") or not path or (writing and kind != "ark"):
        raise SystemExit("unsupported %sspecifier: %s" % ("w" if writing else "r", spec))
    return kind, path, "t" in opts[1:]


def main(argv=None, kind="nnet2"):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # LatticeFasterDecoderConfig::Register (lattice-faster-decoder.h:68-91) + the binary's own options
    ap.add_argument("--beam", type=float, default=16.0)
    ap.add_argument("--max-active", type=int, default=2147483647)
    ap.add_argument("--min-active", type=int, default=200)
    ap.add_argument("--lattice-beam", type=float, default=10.0)
    ap.add_argument("--prune-interval", type=int, default=25)
    ap.add_argument("--beam-delta", type=float, default=0.5)
    ap.add_argument("--hash-ratio", type=float, default=2.0)
    ap.add_argument("--acoustic-scale", type=float, default=0.1)
    ap.add_argument("--allow-partial", type=lambda s: s.lower() in ("true", "1", "t"), default=False)
    ap.add_argument("--determinize-lattice", type=lambda s: s.lower() in ("true", "1", "t"), default=True,
                    help="If true, determinize the lattice (lattice-determinization, keeping only best pdf-sequence for each word-sequence).")
    ap.add_argument("--delta", type=float, default=2.0 ** -10, help="Tolerance used in determinization")
    ap.add_argument("--max-mem", type=int, default=50000000, help="Maximum approximate memory usage in determinization")
    ap.add_argument("--batch-frames", type=int, default=200000, help="frames per forward / decoder launch")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("model")
    ap.add_argument("fst")
    ap.add_argument("features")
    ap.add_argument("lattices")
    ap.add_argument("words", nargs="?")
    ap.add_argument("alignments", nargs="?")
    a = ap.parse_args(argv)

    import torch
    api = importlib.import_module("old-kaldi-git_amd.api")
    kio = importlib.import_module("old-kaldi-git_amd.kaldi_io")
    api.select_gpu(a.gpu)
    t_start = time.time()

    if kind == "nnet2":
        tm, comps, priors = kio.read_nnet2_model(a.model)
        nnet = api.Nnet(comps, priors)
        input_dim = nnet.input_dim()

        def score(feats, off):     # DecodableAmNnet (decodable-am-nnet.h:60-69), batched
            return nnet.compute(feats, off, pad_input=True, epilogue=True, prob_scale=a.acoustic_scale)[0]
    else:
        tm, am = kio.read_gmm_model(a.model)
        gconsts, _ = api.gmm_compute_gconsts(am["weights"], am["means_invvars"], am["inv_vars"])   # DiagGmm::Read :755
        gmm = api.AmDiagGmm(gconsts, am["means_invvars"], am["inv_vars"], am["pdf_offsets"])
        input_dim = am["dim"]

        def score(feats, off):     # DecodableAmDiagGmmScaled::LogLikelihood (decodable-am-diag-gmm.h:142-145)
            ll = gmm.pdf_log_likelihoods(feats)
            api.scale(ll, a.acoustic_scale)
            return ll
    graph = kio.read_fst(a.fst)
    graph["tid2pdf"] = tm["tid2pdf"]
    if int(graph["ilabel"].max(initial=0)) >= len(tm["tid2pdf"]):
        raise SystemExit("HCLG has transition-ids the model does not define")
    fst = api.Fst(graph)
    cfg = api.decoder_config(beam=a.beam, max_active=a.max_active, min_active=a.min_active,
                             lattice_beam=a.lattice_beam, prune_interval=a.prune_interval,
                             beam_delta=a.beam_delta, hash_ratio=a.hash_ratio)

    kind, path, _ = parse_specifier(a.features, False)
    reader = kio.read_ark(path) if kind == "ark" else kio.read_scp(path)
    _, lat_path, lat_text = parse_specifier(a.lattices, True)
    lat_w = kio.TableWriter(lat_path, kind="compact_lattice" if a.determinize_lattice else "lattice", binary=not lat_text)
    words_w = ali_w = None
    if a.words:
        _, p, t = parse_specifier(a.words, True)
        words_w = kio.TableWriter(p, kind="int32_vector", binary=not t)
    if a.alignments:
        _, p, t = parse_specifier(a.alignments, True)
        ali_w = kio.TableWriter(p, kind="int32_vector", binary=not t)

    tot_like, frame_count, num_success, num_fail = 0.0, 0, 0, 0
    dec = None

    def flush(batch):
        nonlocal tot_like, frame_count, num_success, num_fail, dec
        if not batch:
            return
        off = np.concatenate([[0], np.cumsum([len(m) for _, m in batch])]).astype(np.int32)
        feats = torch.from_numpy(np.concatenate([m for _, m in batch], 0).astype(np.float32)).cuda()
        loglikes = score(feats, off)
        max_T = int(np.diff(off).max())
        if dec is None or len(batch) > dec._max_batch or max_T > dec._max_frames:
            dec = api.LatticeFasterDecoder(fst, cfg, max_batch=max(len(batch), 64), max_frames=max(max_T, 1024))
            dec._max_batch, dec._max_frames = max(len(batch), 64), max(max_T, 1024)
        dec.decode(loglikes, off)
        dec.prepare()
        for u, (utt, m) in enumerate(batch):
            st = dec.stats(u)
            if st["status"] != 0 or st["num_tokens"] == 0:     # Decode() returned false (decoder-wrappers.cc:213)
                print("WARNING Failed to decode file %s" % utt, file=sys.stderr)
                num_fail += 1
                continue
            if not st["reached_final"]:
                if a.allow_partial:
                    print("WARNING Outputting partial output for utterance %s since no final-state reached" % utt,
                          file=sys.stderr)
                else:
                    print("WARNING Not producing output for utterance %s since no final-state reached and "
                          "--allow-partial=false." % utt, file=sys.stderr)
                    num_fail += 1
                    continue
            best = dec.get_best_path(u)
            num_frames = len(best["alignment"])
            if words_w:
                words_w.write(utt, best["words"])
            if ali_w:
                ali_w.write(utt, best["alignment"])
            like = -(best["graph_cost"] + best["acoustic_cost"])
            lat = dec.get_raw_lattice(u)
            if a.determinize_lattice:         # decoder-wrappers.cc:264-279
                clat = api.determinize_lattice_pruned(lat, a.lattice_beam, a.delta, a.max_mem)
                if not clat["complete"]:
                    print("WARNING Determinization finished earlier than the beam for utterance %s" % utt, file=sys.stderr)
                if a.acoustic_scale != 0.0:   # "We'll write the lattice without acoustic scaling."
                    inv = np.float32(1.0 / a.acoustic_scale)
                    clat["arc_a"] = (clat["arc_a"] * inv).astype(np.float32)
                    clat["final_a"] = (clat["final_a"] * inv).astype(np.float32)
                lat_w.write(utt, clat)
            else:
                if a.acoustic_scale != 0.0:   # "We'll write the lattice without acoustic scaling." :283-285
                    lat["arc_a"] = (lat["arc_a"] * np.float32(1.0 / a.acoustic_scale)).astype(np.float32)
                lat_w.write(utt, lat)
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (utt, like / max(num_frames, 1), num_frames),
                  file=sys.stderr)
            tot_like += like
            frame_count += len(m)
            num_success += 1

    batch, frames = [], 0
    for utt, m in reader:
        if m.shape[0] == 0:
            print("WARNING Zero-length utterance: %s" % utt, file=sys.stderr)
            num_fail += 1
            continue
        if m.shape[1] != input_dim:
            raise SystemExit("feature dimension %d of %s does not match the model's input %d" % (m.shape[1], utt, input_dim))
        batch.append((utt, m))
        frames += m.shape[0]
        if frames >= a.batch_frames:
            flush(batch)
            batch, frames = [], 0
    flush(batch)
    for w in (lat_w, words_w, ali_w):
        if w:
            w.close()
    elapsed = time.time() - t_start
    # :179-186
    print("LOG Time taken %gs: real-time factor assuming 100 frames/sec is %g" % (elapsed, elapsed * 100.0 / max(frame_count, 1)),
          file=sys.stderr)
    print("LOG Done %d utterances, failed for %d" % (num_success, num_fail), file=sys.stderr)
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(frame_count, 1), frame_count),
          file=sys.stderr)
    return 0 if num_success != 0 else 1


if __name__ == "__main__":
    sys.exit(main())
