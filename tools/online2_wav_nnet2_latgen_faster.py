#!/usr/bin/env python3
"""online2-wav-nnet2-latgen-faster on the MI355X path: the reference binary's command line
(online2bin/online2-wav-nnet2-latgen-faster.cc:78-300), so that steps/online/nnet2/decode.sh:118-125 runs unchanged:

  online2-wav-nnet2-latgen-faster --online=$online --do-endpointing=$do_endpointing \\
     --config=$srcdir/conf/online_nnet2_decoding.conf --max-active=$max_active --beam=$beam --lattice-beam=$lattice_beam \\
     --acoustic-scale=$acwt --word-symbol-table=$graphdir/words.txt $srcdir/final.mdl $graphdir/HCLG.fst \\
     $spk2utt_rspecifier "$wav_rspecifier" "ark:|gzip -c > $dir/lat.JOB.gz"

waveform -> MFCC -> iVector -> [mfcc, ivector] -> nnet2 -> LatticeFasterOnlineDecoder -> pruned determinization ->
CompactLattice, reading the reference's own files (final.mdl, HCLG.fst, conf/*.conf, final.mat, global_cmvn.stats,
final.dubm, final.ie, RIFF waves from files, archives or pipes) through old-kaldi-git_amd/kaldi_io.py / kaldi_cli.py.

  --online=true  (the default)  iVectors estimated online: frame t uses the estimate of period t / ivector-period, computed
                 from the frames up to it (OnlineIvectorFeature::GetFrame, online-ivector-feature.cc:250-282); the waveform is
                 consumed in chunks of --chunk-length seconds; with --do-endpointing=true the rules of
                 online2/online-endpoint.{h,cc} are tested after every chunk and stop the utterance.
  --online=false the binary's :148-152: use_most_recent_ivector + greedy_ivector_extractor, one chunk.

How it runs here (old-kaldi-git_amd/online2.py): the r-th utterance of every speaker forms a round (the speaker's
adaptation state, SetAdaptationState / GetAdaptationState :199,283, chains the rounds); a round's features and network
outputs are computed for all its utterances at once — every stage is causal, so these are the values the reference computes
chunk by chunk — and the decoder's streams consume them in lockstep, chunk index by chunk index, exactly as many frames as
DecodableNnet2Online::NumFramesReady() would report after that chunk.  Without endpointing the chunking cannot change the
result and the round is decoded in one launch.

Silence weighting (--ivector-silence-weighting.silence-phones / .silence-weight / .max-state-duration, what
egs/librispeech/s5/local/online/run_nnet2_ms.sh:216 turns on): the one stage whose values depend on the chunking, because the
weight of a frame in the iVector statistics comes from the decoder's traceback at the time.  Such a round really runs chunk
by chunk (WeightedRound below): traceback -> OnlineSilenceWeighting -> UpdateFrameWeights, then the iVectors of the new
feature frames (api.OnlineIvectorStreams: one launch for all advancing utterances), the network on the new rows
(api.DecodableNnet2Online) and AdvanceDecoding.

Not implemented (an error, not a silent difference): plp / fbank / pitch features; silence weighting with --online=false.
Stated difference (unweighted runs): after an ENDPOINTED utterance the speaker's adaptation state is computed from the
truncated waveform; the reference's iVector statistics lag the last `splice right-context` frames behind that (the weighted
path keeps the statistics as the updates left them, as the reference does)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

USAGE = ("Reads in wav file(s) and simulates online decoding with neural nets\n"
         "(nnet2 setup), with optional iVector-based speaker adaptation and\n"
         "optional endpointing.  Note: some configuration values and inputs are\n"
         "set via config files whose filenames are passed as options\n"
         "\n"
         "Usage: online2-wav-nnet2-latgen-faster [options] <nnet2-in> <fst-in> "
         "<spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>\n"
         "The spk2utt-rspecifier can just be <utterance-id> <utterance-id> if\n"
         "you want to decode utterance by utterance.\n"
         "See egs/rm/s5/local/run_online_decoding_nnet2.sh for example\n"
         "See also online2-wav-nnet2-latgen-threaded\n")


def _bool(s):
    return str(s).lower() in ("true", "1", "t", "")


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
        k = k.replace("_", "-").lower()
        if k not in names:
            raise ValueError("mfcc config: unsupported option --" + k)
        kw[names[k][0]] = names[k][1](v)
    return kw


def ivector_info(conf_path, kio, cli, online):
    """OnlineIvectorExtractionInfo::Init (online2/online-ivector-feature.cc:26-68) from ivector_extractor.conf."""
    c = {k.replace("_", "-").lower(): v for k, v in kio.read_config_file(conf_path).items()}
    for k in ("lda-matrix", "global-cmvn-stats", "diag-ubm", "ivector-extractor", "splice-config", "cmvn-config"):
        if k not in c:
            raise cli.KaldiError("--%s option must be set (%s)" % (k, conf_path))
    splice = kio.read_config_file(c["splice-config"])
    cmvn = kio.read_config_file(c["cmvn-config"])
    lda = cli.read_kaldi_object(c["lda-matrix"], kio.read_matrix).astype(np.float32)
    gstats = cli.read_kaldi_object(c["global-cmvn-stats"], kio.read_matrix).astype(np.float64)
    w, mi, iv = cli.read_kaldi_object(c["diag-ubm"], kio.read_diag_gmm)
    ie = cli.read_kaldi_object(c["ivector-extractor"], kio.read_ivector_extractor)
    var = 1.0 / iv.astype(np.float64)
    # --online=false sets both flags (:148-152); --online=true keeps the file's (their defaults are false)
    most_recent = (not online) or _bool(c.get("use-most-recent-ivector", "false"))
    greedy = (not online) or _bool(c.get("greedy-ivector-extractor", "false"))
    if most_recent != greedy:
        raise cli.KaldiError("--use-most-recent-ivector and --greedy-ivector-extractor are implemented only together "
                             "(both true: what --online=false sets; both false: online estimation)")
    return dict(lda_mat=lda, global_cmvn_stats=gstats, splice_left=int(splice.get("left-context", 4)),
                splice_right=int(splice.get("right-context", 4)), cmn_window=int(cmvn.get("cmn-window", 600)),
                speaker_frames=int(cmvn.get("speaker-frames", 600)), global_frames=int(cmvn.get("global-frames", 200)),
                normalize_mean=_bool(cmvn.get("norm-mean", "true")), normalize_variance=_bool(cmvn.get("norm-vars", "false")),
                ubm_weights=w, ubm_means=(mi.astype(np.float64) * var).astype(np.float32), ubm_vars=var.astype(np.float32),
                M=ie["M"], Sigma_inv=ie["Sigma_inv"], prior_offset=ie["prior_offset"],
                ivector_period=int(c.get("ivector-period", 10)), num_gselect=int(c.get("num-gselect", 5)),
                min_post=float(c.get("min-post", 0.025)), posterior_scale=float(c.get("posterior-scale", 0.1)),
                max_count=float(c.get("max-count", 0.0)), num_cg_iters=int(c.get("num-cg-iters", 15)), greedy_most_recent=bool(greedy),
                max_remembered_frames=float(c.get("max-remembered-frames", 1000)))


def main(argv=None):
    cli = importlib.import_module("old-kaldi-git_amd.kaldi_cli")
    prog = "online2-wav-nnet2-latgen-faster"
    argv = [prog] + list(sys.argv[1:] if argv is None else argv)
    try:
        return run(cli, argv, prog)
    except (cli.KaldiError, ValueError) as e:
        sys.stderr.write("ERROR (%s) %s\n" % (prog, e))
        return 255
    finally:
        cli.stop_pipe_helper()


def run(cli, argv, prog):
    cli.start_pipe_helper()               # before anything initialises the GPU
    kio = importlib.import_module("old-kaldi-git_amd.kaldi_io")
    online2 = importlib.import_module("old-kaldi-git_amd.online2")
    lf = importlib.import_module("tools.latgen_faster")
    po = cli.ParseOptions(USAGE)
    t_start = time.time()
    po.register("chunk-length", 0.05, "Length of chunk size in seconds, that we process.  Set to <= 0 to use all input in one chunk.", float)
    po.register("word-symbol-table", "", "Symbol table for words [for debug output]")
    po.register("do-endpointing", False, "If true, apply endpoint detection")
    po.register("online", True, "You can set this to false to disable online iVector estimation and have all the data for each "
                "utterance used, even at utterance start.  This is useful where you just want the best results and don't care about "
                "online operation.  Setting this to false has the same effect as setting --use-most-recent-ivector=true and "
                "--greedy-ivector-extractor=true in the file given to --ivector-extraction-config, and --chunk-length=-1.")
    po.register("num-threads-startup", 8, "Number of threads used when initializing iVector extractor.", int)
    # OnlineNnet2FeaturePipelineConfig::Register online-nnet2-feature-pipeline.h:92-110
    po.register("feature-type", "mfcc", "Base feature type [mfcc, plp, fbank]")
    po.register("mfcc-config", "", "Configuration file for MFCC features (e.g. conf/mfcc.conf)")
    po.register("plp-config", "", "Configuration file for PLP features (e.g. conf/plp.conf)")
    po.register("fbank-config", "", "Configuration file for filterbank features (e.g. conf/fbank.conf)")
    po.register("add-pitch", False, "Append pitch features to raw MFCC/PLP/filterbank features [but not for iVector extraction]")
    po.register("online-pitch-config", "", "Configuration file for online pitch features, if --add-pitch=true (e.g. conf/online_pitch.conf)")
    po.register("ivector-extraction-config", "", "Configuration file for online iVector extraction, see class "
                "OnlineIvectorExtractionConfig in the code")
    # OnlineSilenceWeightingConfig::RegisterWithPrefix("ivector-silence-weighting") online-ivector-feature.h:301-362
    po.register("ivector-silence-weighting.silence-phones", "", "(RE weighting in iVector estimation for online decoding) List of integer "
                "ids of silence phones, separated by colons (or commas).  Data that (according to the traceback of the decoder) "
                "corresponds to these phones will be downweighted by --silence-weight.")
    po.register("ivector-silence-weighting.silence-weight", 1.0, "(RE weighting in iVector estimation for online decoding) Weighting factor "
                "for frames that the decoder trace-back identifies as silence; only relevant if the --silence-phones option is set.", float)
    po.register("ivector-silence-weighting.max-state-duration", -1.0, "(RE weighting in iVector estimation for online decoding) Maximum "
                "allowed duration of a single transition-id; runs with durations longer than this will be weighted down to the silence-weight.", float)
    # OnlineNnet2DecodingConfig::Register: LatticeFasterDecoderConfig + DecodableNnet2OnlineOptions
    lf.register_decoder_options(po)
    po.register("acoustic-scale", 0.1, "Scaling factor for acoustic likelihoods", float)
    po.register("pad-input", True, "If true, pad acoustic features with required acoustic context past edges of file.")
    po.register("max-nnet-batch-size", 256, "Maximum batch size we use in neural-network decodable object, in cases where we are not "
                "constrained by currently available frames (this will rarely make a difference)", int)
    endpoint = online2.OnlineEndpointConfig()
    endpoint.register(po)
    po.register("reference-order", True, "[MI355X] decode in LatticeFasterDecoder's own iteration order (HashList order, running next_cutoff): the lattices the "
                "reference binary itself writes, bit for bit (the default since round 6)")
    po.register("canonical-order", False, "[MI355X] opt out of --reference-order: the order-independent acceptance rule (a cheaper kernel; same 1-best on every "
                "recipe-like case measured, 0-6 % different raw-lattice arcs; DESIGN.md).  KH_DECODER_ORDER=reference|canonical in the "
                "environment overrides both")
    po.register("gpu", -1, "[MI355X] device ordinal (CuDevice::SelectGpuId); -1: LOCAL_RANK, else 0", int)
    po.register("world", 0, "[MI355X] number of ranks sharing the job (default: WORLD_SIZE, else 1): the SPEAKERS of the spk2utt "
                "table are dealt to the ranks longest-first (a speaker's adaptation state chains its utterances); every JOB in the "
                "arguments becomes rank + 1", int)
    po.register("rank", -1, "[MI355X] this process's rank (default: RANK, else 0)", int)
    po.register("dry-run", False, "[MI355X] read the inputs and the shard, decode nothing (multi-rank plumbing test without a GPU)")
    po.read(argv)
    cli.set_program_name(prog)
    if po.num_args() != 5:
        po.print_usage()
        return 1
    sharding = importlib.import_module("old-kaldi-git_amd.sharding")
    rank, world = sharding.tool_ranks(po["world"], po["rank"])
    if world > 1:
        if "JOB" not in po.get_arg(5):
            raise cli.KaldiError("--world=%d: the lattice table needs JOB in its name (lat.JOB.gz), or the ranks overwrite each other" % world)
        shared_spk2utt = "JOB" not in po.get_arg(3)
        po.positional = sharding.job_substitute(po.positional, rank)
    endpoint.read(po)
    nnet2_rx, fst_rx, spk2utt_rspec, wav_rspec, clat_wspec = (po.get_arg(i) for i in range(1, 6))
    online, do_endpointing, acwt = po["online"], po["do-endpointing"], po["acoustic-scale"]
    if po["feature-type"] != "mfcc" or po["add-pitch"]:
        raise cli.KaldiError("Invalid feature type: %s%s (mfcc without pitch is what is implemented)" % (po["feature-type"], " + pitch" if po["add-pitch"] else ""))
    if not po["pad-input"]:
        raise cli.KaldiError("--pad-input=false is not implemented")
    sw_opts = (po["ivector-silence-weighting.silence-phones"], po["ivector-silence-weighting.silence-weight"],
               po["ivector-silence-weighting.max-state-duration"])
    weighting = sw_opts[0] != "" and sw_opts[1] != 1.0                         # OnlineSilenceWeightingConfig::Active()
    if weighting and (not online or not po["ivector-extraction-config"]):
        raise cli.KaldiError("silence weighting of the iVector statistics needs --online=true and an iVector extractor "
                             "(--online=false estimates one iVector per utterance: --use-most-recent-ivector)")
    chunk_secs = po["chunk-length"] if online else -1.0                        # :148-152
    mfcc_conf = kio.read_config_file(po["mfcc-config"]) if po["mfcc-config"] else {}
    mfcc_kw = mfcc_kwargs(mfcc_conf)
    info = ivector_info(po["ivector-extraction-config"], kio, cli, online) if po["ivector-extraction-config"] else None
    tm, (comps, priors) = cli.read_kaldi_object(nnet2_rx, lambda s, b: (kio.read_transition_model(s, b), kio.read_am_nnet(s, b)))
    graph = cli.read_kaldi_object(fst_rx, lambda s, b: kio.read_fst(s))
    graph["tid2pdf"] = tm["tid2pdf"]
    word_syms = cli.read_symbol_table(po["word-symbol-table"]) if po["word-symbol-table"] != "" else None
    # tables: the speakers in order, every waveform (channel zero, :196-198)
    spk2utt = list(cli.SequentialTableReader(spk2utt_rspec, "token_vector"))
    if world > 1 and shared_spk2utt:      # one table for all ranks: this rank's speakers (a JOB table is the rank's own split)
        spk2utt = sharding.partition_speakers(spk2utt, world)[rank]
    wav_reader = cli.RandomAccessTableReader(wav_rspec, "wave")
    clat_w = cli.TableWriter(clat_wspec, "compact_lattice")
    num_err = 0
    queues = []                           # per speaker: [(utt, samp_freq, samples)]
    for spk, uttlist in spk2utt:
        q = []
        for utt in uttlist:
            if not wav_reader.has_key(utt):
                cli.warn("Did not find audio for utterance " + utt)
                num_err += 1
                continue
            rate, data = wav_reader.value(utt)
            q.append((utt, float(rate), np.ascontiguousarray(data[0], np.float32)))
        queues.append((spk, q))

    if po["dry-run"]:
        n_utts = sum(len(q) for _, q in queues)
        n_samp = sum(len(w) for _, q in queues for _, _, w in q)
        clat_w.close()
        cli.log("Decoded %d utterances, %d with errors." % (n_utts, num_err))
        return finish(cli, sharding, world, "gloo", time.time() - t_start, [0.0, n_samp, n_utts, num_err])

    # ---- the GPU from here on
    import torch
    api = importlib.import_module("old-kaldi-git_amd.api")
    api.select_gpu(po["gpu"] if po["gpu"] >= 0 else int(os.environ.get("LOCAL_RANK", "0")))
    mfcc = api.Mfcc(**mfcc_kw)
    ivec = api.OnlineIvectorExtractor(info) if info is not None else None
    pipe = api.OnlineNnet2FeaturePipeline(mfcc, ivec)
    nnet = api.Nnet(comps, priors)
    if nnet.input_dim() != pipe.dim():
        raise cli.KaldiError("feature dimension %d does not match the network's input %d" % (pipe.dim(), nnet.input_dim()))
    fst = api.Fst(graph)
    cfg = lf.decoder_config(api, po)
    det_opts = lf.determinize_options(api, po, tm)
    max_rem = info["max_remembered_frames"] if info is not None else 1000.0
    frame_shift = mfcc_kw.get("frame_shift_ms", 10.0) * 0.001
    num_done, tot_like, num_frames = 0, 0.0, 0
    state = {}                            # speaker -> adaptation state
    n_rounds = max((len(q) for _, q in queues), default=0)
    for r in range(n_rounds):
        batch = [(spk, q[r]) for spk, q in queues if r < len(q)]
        for _, (utt, rate, _) in batch:
            if rate != mfcc.samp_freq:
                raise cli.KaldiError("Sampling frequency mismatch, expected %g, got %g" % (mfcc.samp_freq, rate))   # AcceptWaveform
        waves = [torch.from_numpy(w).cuda() for _, (_, _, w) in batch]
        speakers = [spk for spk, _ in batch]
        # the schedule of every utterance: samples accepted after chunk k -> frames the decoder may consume
        sched, feat_sched, base_sched = [], [], []   # per utterance and chunk: NumFramesReady() of the decodable / the feature pipeline / OnlineMfcc
        for _, (_, rate, w) in batch:
            n = len(w)
            chunk = max(1, int(rate * chunk_secs)) if chunk_secs > 0 else max(n, 1)
            offs = list(range(chunk, n, chunk)) + [n] if n > 0 else []
            feat_sched.append([online2.frames_ready_after(o, o == n, rate, mfcc_kw, info["splice_right"] if info is not None else None, 0, True, 0)
                               for o in offs])
            base_sched.append([online2.frames_ready_after(o, o == n, rate, mfcc_kw, None, 0, True, 0) for o in offs])
            sched.append([online2.frames_ready_after(o, o == n, rate, mfcc_kw, info["splice_right"] if info is not None else None,
                                                     nnet.right_context(), True, nnet.left_context()) for o in offs])
        wround = None
        if weighting:
            wround = WeightedRound(api, online2, pipe, ivec, nnet, waves, speakers, state, acwt, po["max-nnet-batch-size"], tm["tid2phone"],
                                   sw_opts, feat_sched, base_sched)
            feats, off, new_state = wround.feats, wround.off, {}
        else:
            feats, off, new_state = compute_features(api, pipe, ivec, waves, speakers, state, max_rem)
        keep = [u for u in range(len(batch)) if off[u + 1] > off[u]]
        for u in range(len(batch)):
            if off[u + 1] == off[u]:
                cli.warn("no frames for utterance %s" % batch[u][1][0])
                num_err += 1
        if not keep:
            state.update(new_state)
            continue
        off_k = np.concatenate([[0], np.cumsum([off[u + 1] - off[u] for u in keep])]).astype(np.int32)
        stopped = [None] * len(keep)
        if wround is not None:
            # the iVector rows depend on the decoder's traceback: features, network and decoder advance chunk by chunk
            odec = api.LatticeFasterOnlineDecoder(fst, cfg, num_streams=len(keep), max_frames=int(np.diff(off_k).max()),
                                                  exact_reference_order=bool(po["reference-order"]) and not bool(po["canonical-order"]))
            wround.begin(keep, odec)
            decoded, stopped = online2.simulate(odec, None, off_k, [sched[u] for u in keep], endpoint if do_endpointing else None,
                                                tm["tid2phone"], frame_shift, before_advance=wround.before_advance, rows_of=wround.rows_of)
            new_state = wround.adaptation_states(keep, stopped, max_rem)
            get = lambda j: (odec.stats(j), odec.get_raw_lattice(j), odec.get_best_path(j))
        else:
            rows = torch.cat([feats[off[u]:off[u + 1]] for u in keep], 0) if len(keep) != len(batch) else feats
            ll, ll_off = nnet.compute(rows, off_k, True, epilogue=True, prob_scale=acwt)
            ll_off = np.asarray(ll_off, np.int32)
        if wround is not None:
            pass
        elif do_endpointing:
            odec = api.LatticeFasterOnlineDecoder(fst, cfg, num_streams=len(keep), max_frames=int(np.diff(ll_off).max()),
                                                  exact_reference_order=bool(po["reference-order"]) and not bool(po["canonical-order"]))
            decoded, stopped = online2.simulate(odec, ll, ll_off, [sched[u] for u in keep], endpoint, tm["tid2phone"], frame_shift)
            get = lambda j: (odec.stats(j), odec.get_raw_lattice(j), odec.get_best_path(j))
        else:
            dec = api.LatticeFasterDecoder(fst, cfg, max_batch=len(keep), max_frames=int(np.diff(ll_off).max()),
                                           exact_reference_order=bool(po["reference-order"]) and not bool(po["canonical-order"]))
            dec.set_determinize(True, **det_opts)
            dec.decode(ll, ll_off)
            dec.prepare()
            get = lambda j: (dec.stats(j), None, dec.get_best_path(j))
        redo = []                         # endpointed utterances: their adaptation state comes from the truncated waveform
        for j, u in enumerate(keep):
            utt = batch[u][1][0]
            st, raw, best = get(j)
            if st["status"] != 0 or st["num_tokens"] == 0:
                cli.warn("Failed to decode utterance " + utt)
                num_err += 1
                continue
            # GetLattice(end_of_utterance = true): final-probs applied, then determinized (online-nnet2-decoding.cc:47-66)
            clat = api.determinize_lattice_pruned(raw, det_opts["beam"], det_opts["delta"], det_opts["max_mem"], tid_phone=det_opts["tid_phone"],
                                                  phone_determinize=det_opts["phone_determinize"], word_determinize=det_opts["word_determinize"],
                                                  minimize=det_opts["minimize"]) if raw is not None else dec.get_compact_lattice(j)
            # GetDiagnosticsAndPrintOutput :35-70 (the best path of the CompactLattice = the decoder's)
            like = -(best["graph_cost"] + best["acoustic_cost"])
            n = len(best["alignment"])
            cli.vlog(2, "Likelihood per frame for utterance %s is %g over %d frames." % (utt, like / max(n, 1), n))
            if word_syms is not None:
                sys.stderr.write(utt + " " + "".join(word_syms[int(w)] + " " for w in best["words"]) + "\n")
            if acwt != 0.0:               # "we want to output the lattice with un-scaled acoustics" :286-289
                inv = np.float32(1.0 / acwt)
                clat["arc_a"] = (clat["arc_a"] * inv).astype(np.float32)
                clat["final_a"] = (clat["final_a"] * inv).astype(np.float32)
            clat_w.write(utt, clat)
            cli.log("Decoded utterance " + utt)
            tot_like += like
            num_frames += n
            num_done += 1
            if stopped[j] is not None and wround is None:
                redo.append((u, stopped[j]))
        if redo:                           # (see the module docstring: "Stated difference")
            cut = []
            for u, k in redo:
                rate, w = batch[u][1][1], batch[u][1][2]
                chunk = max(1, int(rate * chunk_secs)) if chunk_secs > 0 else len(w)
                cut.append(torch.from_numpy(w[:min(len(w), (k + 1) * chunk)]).cuda())
            _, _, st2 = compute_features(api, pipe, ivec, cut, [speakers[u] for u, _ in redo], state, max_rem)
            new_state.update(st2)
        state.update(new_state)
    ok = clat_w.close()
    elapsed = time.time() - t_start
    cli.log("Decoded %d utterances, %d with errors." % (num_done, num_err))
    cli.log("Overall likelihood per frame was %g per frame over %d frames." % (tot_like / max(num_frames, 1), num_frames))
    cli.vlog(1, "Time taken %gs: real-time factor assuming 100 frames/sec is %g" % (elapsed, elapsed * 100.0 / max(num_frames, 1)))
    if not ok:
        raise cli.KaldiError("error closing the lattice table " + clat_wspec)
    return finish(cli, sharding, world, "nccl", elapsed, [tot_like, num_frames, num_done, num_err])


def finish(cli, sharding, world, backend, elapsed, totals):
    """With a launcher's rendezvous: the totals over all ranks from rank 0 (one all-reduce of four numbers)."""
    tot_like, num_frames, num_done, num_err = totals
    if sharding.init_tool_group(world, backend):
        import torch.distributed as dist
        tot = sharding.reduce_decode_totals(num_frames, tot_like, num_done, num_err, elapsed, device="cuda" if backend == "nccl" else "cpu")
        if dist.get_rank() == 0:
            cli.log("All %d ranks: decoded %d utterances, %d with errors; overall likelihood per frame was %g over %d frames"
                    % (world, tot["num_success"], tot["num_fail"], tot["loglike_per_frame"], int(tot["frames"])))
        dist.barrier()
        dist.destroy_process_group()
    return 0 if num_done != 0 else 1


class WeightedRound:
    """One round (at most one utterance per speaker) with --ivector-silence-weighting active: the loop body of
    online2-wav-nnet2-latgen-faster.cc:226-262 for all its utterances at once.  Per chunk and live utterance: the decoder's
    traceback -> OnlineSilenceWeighting -> UpdateFrameWeights (:239-244); then AdvanceDecoding pulls the new rows of
    DecodableNnet2Online, whose input rows [mfcc, ivector] pull OnlineIvectorFeature::GetFrame up to the last feature frame
    that is ready - api.OnlineIvectorStreams.get_frames for all advancing utterances in one launch, one network pass over
    their new rows, one launch of the online decode kernel (online2.simulate)."""

    def __init__(self, api, online2, pipe, ivec, nnet, waves, speakers, state, acwt, max_nnet_batch, tid2phone, sw_opts, feat_ready, base_ready):
        import torch
        self.api, self.online2, self.ivec, self.nnet, self.torch = api, online2, ivec, nnet, torch
        self.acwt, self.max_nnet_batch, self.tid2phone, self.sw_opts = acwt, max_nnet_batch, tid2phone, sw_opts
        self.speakers, self.state_in, self.waves = speakers, state, waves
        self.feat_ready_all = feat_ready           # [utterance][chunk] = OnlineNnet2FeaturePipeline::NumFramesReady()
        self.base_ready_all = base_ready           # ... = OnlineMfcc::NumFramesReady() (what OnlineCmvn has seen)
        base = [pipe.mfcc.compute(w) for w in waves]
        lens = np.array([b.shape[0] for b in base], np.int64)
        self.off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows, self.d0 = int(self.off[-1]), pipe.mfcc.num_ceps
        stride = (pipe.dim() + 3) // 4 * 4
        self.feats = torch.zeros((max(rows, 1), stride), dtype=torch.float32, device="cuda")[:rows, :pipe.dim()]
        for u, b in enumerate(base):
            if b.shape[0]:
                self.feats[self.off[u]:self.off[u + 1], :self.d0] = b
        self.base = base

    def begin(self, keep, odec):
        torch, api = self.torch, self.api
        self.keep, self.odec = keep, odec
        lens = [int(self.off[u + 1] - self.off[u]) for u in keep]
        self.boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        self.base_cat = torch.cat([self.base[u] for u in keep], 0).contiguous()
        # the rows of the kept utterances, [mfcc | ivector]; the iVector block is the streams' output matrix
        self.rows = self.feats if len(keep) == len(self.base) else torch.cat([self.feats[self.off[u]:self.off[u + 1]] for u in keep], 0)
        if len(keep) != len(self.base):
            stride = (self.rows.shape[1] + 3) // 4 * 4
            full = torch.zeros((self.rows.shape[0], stride), dtype=torch.float32, device="cuda")[:, :self.rows.shape[1]]
            full.copy_(self.rows)
            self.rows = full
        self.st_in = np.stack([self.state_in[self.speakers[u]] if self.speakers[u] in self.state_in else self.ivec.fresh_state(1)[0]
                               for u in keep])
        self.streams = api.OnlineIvectorStreams(self.ivec, self.base_cat, self.boff, self.rows[:, self.d0:], state=self.st_in)
        self.dn = api.DecodableNnet2Online(self.nnet, len(keep), max(lens), acoustic_scale=self.acwt, pad_input=True,
                                           max_nnet_batch_size=self.max_nnet_batch)
        self.sw = [self.online2.OnlineSilenceWeighting(self.tid2phone, *self.sw_opts) for _ in keep]
        self.given = [0] * len(keep)
        self.feat_ready = [self.feat_ready_all[u] for u in keep]
        self.k = -1

    def before_advance(self, k, now, decoded):
        self.k = k
        for j in now:
            ali = self.odec.get_best_path(j, use_final_probs=False)["alignment"] if decoded[j] > 0 else []
            self.sw[j].compute_current_traceback(ali)
            ready = self.feat_ready[j][k]
            self.streams.update_frame_weights(j, self.sw[j].get_delta_weights(ready), ready)

    def rows_of(self, todo, first, end):
        k = self.k
        ready = [self.feat_ready[j][k] for j in todo]
        self.streams.get_frames(todo, [r - 1 for r in ready])
        self.dn.accept_features_many(todo, self.rows, [int(self.boff[j] + self.given[j]) for j in todo],
                                     [r - self.given[j] for j, r in zip(todo, ready)],
                                     [k == len(self.feat_ready[j]) - 1 for j in todo])
        for j, r in zip(todo, ready):
            self.given[j] = r
        parts = [[] for _ in todo]
        cur = list(first)
        while True:                                   # (more than --max-nnet-batch-size new frames: several ComputeForFrame calls)
            idx = [i for i in range(len(todo)) if cur[i] < end[i]]
            if not idx:
                break
            out = self.dn.compute([todo[i] for i in idx], [cur[i] for i in idx])
            for i, o in zip(idx, out):
                assert o.shape[0] > 0 and cur[i] + o.shape[0] <= end[i], "NumFramesReady() of the decodable and the schedule disagree"
                parts[i].append(o)
                cur[i] += o.shape[0]
        return [p[0] if len(p) == 1 else self.torch.cat(p, 0) for p in parts]

    def adaptation_states(self, keep, stopped, max_remembered_frames):
        """GetAdaptationState :283 per utterance: cmvn_->GetState(NumFramesReady() - 1) = the CMVN statistics over the MFCC frames
        accepted when the utterance ended (all of them unless it endpointed at chunk stopped[j]), the iVector statistics as
        the weighted updates left them (weights still queued are not in them, as in the reference)."""
        accepted = [max(1, self.base_ready_all[u][stopped[j]] if stopped[j] is not None else int(self.off[u + 1] - self.off[u]))
                    for j, u in enumerate(keep)]
        cut = self.torch.cat([self.base[u][:accepted[j]] for j, u in enumerate(keep)], 0).contiguous()
        coff = np.concatenate([[0], np.cumsum(accepted)]).astype(np.int32)
        _, st = self.ivec.extract(cut, coff, state=self.st_in, return_state=True)
        st = self.streams.get_stats(st)
        self.ivec.limit_frames(st, max_remembered_frames)
        return {self.speakers[u]: st[j] for j, u in enumerate(keep)}


def compute_features(api, pipe, ivec, waves, speakers, state, max_remembered_frames):
    """One round (at most one utterance per speaker): [mfcc, ivector] rows of every waveform, starting from the speakers'
    adaptation states; returns (features, row offsets, {speaker: state after its utterance})."""
    import torch
    base = [pipe.mfcc.compute(w) for w in waves]
    lens = np.array([b.shape[0] for b in base], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows, d0 = int(off[-1]), pipe.mfcc.num_ceps
    stride = (pipe.dim() + 3) // 4 * 4
    out = torch.empty((max(rows, 1), stride), dtype=torch.float32, device="cuda")[:rows, :pipe.dim()]
    for u, b in enumerate(base):
        if b.shape[0]:
            out[off[u]:off[u + 1], :d0] = b
    new_state = {}
    if ivec is None or rows == 0:
        return out, off, new_state
    live = [u for u in range(len(waves)) if lens[u] > 0]
    feats = torch.cat([base[u] for u in live], 0).contiguous()
    boff = np.concatenate([[0], np.cumsum([lens[u] for u in live])]).astype(np.int32)
    st_in = np.stack([state[speakers[u]] if speakers[u] in state else ivec.fresh_state(1)[0] for u in live])
    iv, st = ivec.extract(feats, boff, state=st_in, return_state=True)
    ivec.limit_frames(st, max_remembered_frames)
    for j, u in enumerate(live):
        out[off[u]:off[u + 1], d0:] = iv[boff[j]:boff[j + 1]]
        new_state[speakers[u]] = st[j]
    return out, off, new_state


if __name__ == "__main__":
    sys.exit(main())
