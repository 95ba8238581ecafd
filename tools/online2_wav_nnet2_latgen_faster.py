#!/usr/bin/env python3
"""online2-wav-nnet2-latgen-faster --online=false on the MI355X path: the reference binary's command
line (online2bin/online2-wav-nnet2-latgen-faster.cc:78-300) over the library — waveform -> MFCC ->
iVector (use_most_recent_ivector + greedy_ivector_extractor, what --online=false sets, :148-152) ->
[mfcc, ivector] -> nnet2 -> LatticeFasterDecoder -> pruned determinization -> CompactLattice — reading
the reference's own files (final.mdl, HCLG.fst, conf/*.conf, final.mat, global_cmvn.stats, final.dubm,
final.ie, RIFF wave files) with old-kaldi-git_amd/kaldi_io.py.

  online2_wav_nnet2_latgen_faster.py [options] <nnet2-in> <fst-in> <spk2utt-rspecifier> \\
      <wav-rspecifier> <lattice-wspecifier>

  options: --config=FILE (one --name=value per line, e.g. conf/online_nnet2_decoding.conf),
           --mfcc-config, --ivector-extraction-config, --feature-type=mfcc, --online=false,
           --beam --max-active --min-active --lattice-beam --acoustic-scale (OnlineNnet2DecodingConfig)
  <spk2utt-rspecifier>  ark:FILE  "spk utt1 utt2 ..." per line ("utt utt" to decode utterance by utterance)
  <wav-rspecifier>      scp:FILE  "utt path.wav" per line (files; commands ending in | are not run)
  <lattice-wspecifier>  ark:FILE | ark,t:FILE  CompactLattices, acoustic costs unscaled (:61-66 of the binary's GetLattice use)

Differences from the binary, stated: the utterances are processed as one batch per stage (one MFCC /
forward / decoder launch; the iVectors in rounds — the r-th utterance of every speaker together, with
the adaptation state SetAdaptationState / GetAdaptationState carry from one utterance of a speaker to
the next, --max-remembered-frames); --online=true (chunk-wise estimates, endpointing, silence
weighting) is not implemented and is refused.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _bool(s):
    return str(s).lower() in ("true", "1", "t", "yes")


def mfcc_kwargs(conf):
    """MfccOptions::Register + FrameExtractionOptions + MelBanksOptions names -> api.Mfcc arguments."""
    names = {"sample-frequency": ("samp_freq", float), "frame-length": ("frame_length_ms", float), "frame-shift": ("frame_shift_ms", float),
             "preemphasis-coefficient": ("preemph_coeff", float), "remove-dc-offset": ("remove_dc_offset", _bool),
             "window-type": ("window_type", str), "num-mel-bins": ("num_bins", int), "low-freq": ("low_freq", float),
             "high-freq": ("high_freq", float), "num-ceps": ("num_ceps", int), "cepstral-lifter": ("cepstral_lifter", float),
             "snip-edges": ("snip_edges", _bool), "use-energy": ("use_energy", _bool), "raw-energy": ("raw_energy", _bool),
             "energy-floor": ("energy_floor", float), "htk-compat": ("htk_compat", _bool), "dither": ("dither", float)}
    # the reference's struct defaults (feature-mfcc.h:54, feature-functions.h:91) where they differ from api.Mfcc's
    kw = dict(use_energy=True, dither=1.0)
    for k, v in conf.items():
        if k not in names:
            raise SystemExit("mfcc config: unsupported option --" + k)
        kw[names[k][0]] = names[k][1](v)
    return kw


def ivector_info(conf_path, kio, online):
    """OnlineIvectorExtractionInfo::Init (online2/online-ivector-feature.cc:26-68) from ivector_extractor.conf."""
    c = kio.read_config_file(conf_path)
    need = ("lda-matrix", "global-cmvn-stats", "diag-ubm", "ivector-extractor", "splice-config", "cmvn-config")
    for k in need:
        if k not in c:
            raise SystemExit("--%s option must be set (%s)" % (k, conf_path))
    splice = kio.read_config_file(c["splice-config"])
    cmvn = kio.read_config_file(c["cmvn-config"])
    lda = kio.read_kaldi_object(c["lda-matrix"], kio.read_matrix).astype(np.float32)
    gstats = kio.read_kaldi_object(c["global-cmvn-stats"], kio.read_matrix).astype(np.float64)
    w, mi, iv = kio.read_kaldi_object(c["diag-ubm"], kio.read_diag_gmm)
    ie = kio.read_kaldi_object(c["ivector-extractor"], kio.read_ivector_extractor)
    var = 1.0 / iv.astype(np.float64)
    most_recent = _bool(c.get("use-most-recent-ivector", "true")) or _bool(c.get("greedy-ivector-extractor", "false"))
    if online:
        raise SystemExit("--online=true is not implemented: run with --online=false")
    del most_recent   # --online=false sets both flags (online2-wav-nnet2-latgen-faster.cc:148-152)
    return dict(lda_mat=lda, global_cmvn_stats=gstats, splice_left=int(splice.get("left-context", 4)),
                splice_right=int(splice.get("right-context", 4)), cmn_window=int(cmvn.get("cmn-window", 600)),
                speaker_frames=int(cmvn.get("speaker-frames", 600)), global_frames=int(cmvn.get("global-frames", 200)),
                normalize_mean=_bool(cmvn.get("norm-mean", "true")), normalize_variance=_bool(cmvn.get("norm-vars", "false")),
                ubm_weights=w, ubm_means=(mi.astype(np.float64) * var).astype(np.float32), ubm_vars=var.astype(np.float32),
                M=ie["M"], Sigma_inv=ie["Sigma_inv"], prior_offset=ie["prior_offset"],
                ivector_period=int(c.get("ivector-period", 10)), num_gselect=int(c.get("num-gselect", 5)),
                min_post=float(c.get("min-post", 0.025)), posterior_scale=float(c.get("posterior-scale", 0.1)),
                max_count=float(c.get("max-count", 0.0)), num_cg_iters=15, greedy_most_recent=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    # --config=FILE: its lines are options of this command line (ParseOptions::ReadConfigFile)
    kio = importlib.import_module("old-kaldi-git_amd.kaldi_io")
    expanded = []
    for a in argv:
        if a.startswith("--config="):
            for k, v in kio.read_config_file(a.split("=", 1)[1]).items():
                expanded.append("--%s=%s" % (k, v))
        else:
            expanded.append(a)
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--feature-type", default="mfcc")
    ap.add_argument("--mfcc-config", default="")
    ap.add_argument("--ivector-extraction-config", default="")
    ap.add_argument("--online", type=_bool, default=True)
    ap.add_argument("--do-endpointing", type=_bool, default=False)
    ap.add_argument("--chunk-length", type=float, default=0.05)
    ap.add_argument("--beam", type=float, default=16.0)
    ap.add_argument("--max-active", type=int, default=2147483647)
    ap.add_argument("--min-active", type=int, default=200)
    ap.add_argument("--lattice-beam", type=float, default=10.0)
    ap.add_argument("--prune-interval", type=int, default=25)
    ap.add_argument("--beam-delta", type=float, default=0.5)
    ap.add_argument("--hash-ratio", type=float, default=2.0)
    ap.add_argument("--acoustic-scale", type=float, default=0.1)
    ap.add_argument("--max-nnet-batch-size", type=int, default=256)
    ap.add_argument("--determinize-lattice", type=_bool, default=True)
    ap.add_argument("--delta", type=float, default=2.0 ** -10)
    ap.add_argument("--max-mem", type=int, default=50000000)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("nnet2")
    ap.add_argument("fst")
    ap.add_argument("spk2utt")
    ap.add_argument("wav")
    ap.add_argument("lattices")
    a = ap.parse_args(expanded)
    if a.feature_type != "mfcc":
        raise SystemExit("Invalid feature type: %s (only mfcc is implemented)" % a.feature_type)
    if a.do_endpointing:
        raise SystemExit("--do-endpointing=true is not implemented")

    import torch
    api = importlib.import_module("old-kaldi-git_amd.api")
    lf = importlib.import_module("tools.latgen_faster")
    api.select_gpu(a.gpu)
    t_start = time.time()
    mfcc = api.Mfcc(**mfcc_kwargs(kio.read_config_file(a.mfcc_config) if a.mfcc_config else {}))
    ivec = api.OnlineIvectorExtractor(ivector_info(a.ivector_extraction_config, kio, a.online)) if a.ivector_extraction_config else None
    if ivec is None and a.online:
        raise SystemExit("--online=true is not implemented: run with --online=false")
    max_rem = float(kio.read_config_file(a.ivector_extraction_config).get("max-remembered-frames", 1000)) if a.ivector_extraction_config else 1000.0
    pipe = api.OnlineNnet2FeaturePipeline(mfcc, ivec)
    tm, comps, priors = kio.read_nnet2_model(a.nnet2)
    nnet = api.Nnet(comps, priors)
    if nnet.input_dim() != pipe.dim():
        raise SystemExit("feature dimension %d does not match the network's input %d" % (pipe.dim(), nnet.input_dim()))
    graph = kio.read_fst(a.fst)
    graph["tid2pdf"] = tm["tid2pdf"]
    fst = api.Fst(graph)
    cfg = api.decoder_config(beam=a.beam, max_active=a.max_active, min_active=a.min_active, lattice_beam=a.lattice_beam,
                             prune_interval=a.prune_interval, beam_delta=a.beam_delta, hash_ratio=a.hash_ratio)
    kind, spk_path, _ = lf.parse_specifier(a.spk2utt, False)
    wkind, wav_path, _ = lf.parse_specifier(a.wav, False)
    if kind != "ark" or wkind != "scp":
        raise SystemExit("spk2utt must be ark:FILE and the waveforms scp:FILE")
    wav_of = {}
    with open(wav_path) as f:
        for line in f:
            k, _, v = line.strip().partition(" ")
            if k:
                wav_of[k] = v.strip()
    utts, spk_of, num_err = [], [], 0
    with open(spk_path) as f:
        for line in f:
            toks = line.split()
            for utt in toks[1:]:
                if utt not in wav_of:
                    print("WARNING Did not find audio for utterance %s" % utt, file=sys.stderr)
                    num_err += 1
                elif wav_of[utt].endswith("|"):
                    raise SystemExit("wav.scp commands are not run: " + wav_of[utt])
                else:
                    utts.append(utt)
                    spk_of.append(toks[0])
    _, lat_path, lat_text = lf.parse_specifier(a.lattices, True)
    lat_w = kio.TableWriter(lat_path, kind="compact_lattice", binary=not lat_text)
    waves = []
    for utt in utts:
        rate, data = kio.read_wave(wav_of[utt])
        if rate != mfcc.samp_freq:
            raise SystemExit("Sampling frequency mismatch, expected %g, got %g" % (mfcc.samp_freq, rate))   # online-feature.cc AcceptWaveform
        waves.append(torch.from_numpy(np.ascontiguousarray(data[0])).cuda())    # channel zero (:196-198)
    feats, off = pipe.compute(waves, speakers=spk_of, max_remembered_frames=max_rem)
    keep = [u for u in range(len(utts)) if off[u + 1] > off[u]]
    for u in range(len(utts)):
        if off[u + 1] == off[u]:
            print("WARNING no frames for utterance %s" % utts[u], file=sys.stderr)
            num_err += 1
    num_done, tot_like, num_frames = 0, 0.0, 0
    if keep:
        off_k = np.concatenate([[0], np.cumsum([off[u + 1] - off[u] for u in keep])]).astype(np.int32)
        ll, ll_off = nnet.compute(feats, off_k, True, epilogue=True, prob_scale=a.acoustic_scale)
        dec = api.LatticeFasterDecoder(fst, cfg, max_batch=len(keep), max_frames=int(np.diff(ll_off).max()))
        dec.decode(ll, np.asarray(ll_off, np.int32))
        dec.prepare()
        for j, u in enumerate(keep):
            st = dec.stats(j)
            if st["status"] != 0 or st["num_tokens"] == 0:
                print("WARNING Failed to decode utterance %s" % utts[u], file=sys.stderr)
                num_err += 1
                continue
            raw = dec.get_raw_lattice(j)      # GetLattice(end_of_utterance = true): final-probs applied, then determinized
            clat = api.determinize_lattice_pruned(raw, a.lattice_beam, a.delta, a.max_mem)
            best = dec.get_best_path(j)       # (= CompactLatticeShortestPath of clat: GetDiagnosticsAndPrintOutput :35-70)
            like = -(best["graph_cost"] + best["acoustic_cost"])
            n = len(best["alignment"])
            if a.acoustic_scale != 0.0:       # "we want the output lattices to have un-scaled acoustics" :289-291
                inv = np.float32(1.0 / a.acoustic_scale)
                clat["arc_a"] = (clat["arc_a"] * inv).astype(np.float32)
                clat["final_a"] = (clat["final_a"] * inv).astype(np.float32)
            lat_w.write(utts[u], clat)
            print("LOG Decoded utterance %s" % utts[u], file=sys.stderr)
            tot_like += like
            num_frames += n
            num_done += 1
    lat_w.close()
    elapsed = time.time() - t_start
    print("LOG Decoded %d utterances, %d with errors." % (num_done, num_err), file=sys.stderr)
    print("LOG Overall likelihood per frame was %g per frame over %d frames." % (tot_like / max(num_frames, 1), num_frames), file=sys.stderr)
    print("LOG Time taken %gs: real-time factor assuming 100 frames/sec is %g" % (elapsed, elapsed * 100.0 / max(num_frames, 1)),
          file=sys.stderr)
    return 0 if num_done != 0 else 1


if __name__ == "__main__":
    sys.exit(main())
