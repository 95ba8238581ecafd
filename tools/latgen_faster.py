#!/usr/bin/env python3
"""nnet-latgen-faster / gmm-latgen-faster on the MI355X path: the reference binaries' command line
(nnet2bin/nnet-latgen-faster.cc:40-190, gmmbin/gmm-latgen-faster.cc:36-180) over the library, so that the recipe line of
steps/nnet2/decode.sh:130-136 runs unchanged:

  nnet-latgen-faster --minimize=$minimize --max-active=$max_active --min-active=$min_active --beam=$beam \\
     --lattice-beam=$lattice_beam --acoustic-scale=$acwt --allow-partial=true --word-symbol-table=$graphdir/words.txt \\
     "$model" $graphdir/HCLG.fst "$feats" "ark:|gzip -c > $dir/lat.JOB.gz"

with "$feats" = "ark,s,cs:apply-cmvn ... scp:feats.scp ark:- | splice-feats ... |".  Everything the binary's ParseOptions
does (old-kaldi-git_amd/kaldi_cli.py): --config, --verbose, --help, --print-args, bool forms, `_` for `-`; every
rspecifier / wspecifier / rxfilename / wxfilename form (ark / scp / ark,scp with their options, "-", pipes, file:offset);
every option LatticeFasterDecoderConfig and DeterminizeLatticePhonePrunedOptions register
(lattice-faster-decoder.h:67-91, determinize-lattice-pruned.h:168-186); the binary's log lines and exit codes (0 if an
utterance was decoded, 1 if none, 255 on an error - "return -1").  Run through nnet_latgen_faster.py / gmm_latgen_faster.py.

Differences from the binary: utterances are decoded in batches (--batch-frames, not a reference option) — one forward
pass and one decoder launch per batch, the determinization of a batch on host threads while its decode kernel runs — the
results per utterance are the same (tests/test_gpu_latgen_tool.py); a table of per-utterance FSTs as <fst-in> is refused.
Pipe children are started by a helper process forked before the GPU is initialised (kaldi_cli.start_pipe_helper)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

USAGE = {
    "nnet2": ("Generate lattices using neural net model.\n"
              "Usage: nnet-latgen-faster [options] <nnet-in> <fst-in|fsts-rspecifier> <features-rspecifier>"
              " <lattice-wspecifier> [ <words-wspecifier> [<alignments-wspecifier>] ]\n"),
    "gmm": ("Generate lattices using GMM-based model.\n"
            "Usage: gmm-latgen-faster [options] model-in (fst-in|fsts-rspecifier) features-rspecifier"
            " lattice-wspecifier [ words-wspecifier [alignments-wspecifier] ]\n"),
}


def register_decoder_options(po):
    """LatticeFasterDecoderConfig::Register (lattice-faster-decoder.h:67-86) incl. det_opts
    (DeterminizeLatticePhonePrunedOptions::Register, determinize-lattice-pruned.h:168-186)."""
    po.register("delta", 0.0009765625, "Tolerance used in determinization", float)
    po.register("max-mem", 50000000, "Maximum approximate memory usage in determinization (real usage might be many times this).", int)
    po.register("phone-determinize", True, "If true, do an initial pass of determinization on both phones and words (see also "
                "--word-determinize)")
    po.register("word-determinize", True, "If true, do a second pass of determinization on words only (see also --phone-determinize)")
    po.register("minimize", False, "If true, push and minimize after determinization.")
    po.register("beam", 16.0, "Decoding beam.", float)
    po.register("max-active", 2147483647, "Decoder max active states.", int)
    po.register("min-active", 200, "Decoder minimum #active states.", int)
    po.register("lattice-beam", 10.0, "Lattice generation beam", float)
    po.register("prune-interval", 25, "Interval (in frames) at which to prune tokens", int)
    po.register("determinize-lattice", True, "If true, determinize the lattice (in a special sense, keeping only best "
                "pdf-sequence for each word-sequence).")
    po.register("beam-delta", 0.5, "Increment used in decoding-- this parameter is obscure and relates to a speedup in the way the "
                "max-active constraint is applied.  Larger is more accurate.", float)
    po.register("hash-ratio", 2.0, "Setting used in decoder to control hash behavior", float)


def decoder_config(api, po):
    return api.decoder_config(beam=po["beam"], max_active=po["max-active"], min_active=po["min-active"],
                              lattice_beam=po["lattice-beam"], prune_interval=po["prune-interval"],
                              beam_delta=po["beam-delta"], hash_ratio=po["hash-ratio"])


def determinize_options(api, po, tm):
    return dict(beam=po["lattice-beam"], delta=po["delta"], max_mem=po["max-mem"], tid_phone=api.tid_phone_map(tm),
                phone_determinize=po["phone-determinize"], word_determinize=po["word-determinize"], minimize=po["minimize"])


def write_utterance(cli, api, dec, u, utt, num_rows, po, writers, word_syms, totals, prog):
    """DecodeUtteranceLatticeFaster decoder-wrappers.cc:197-293 behind Decode(): outputs + log lines of one utterance.
    totals = [tot_like, frame_count, num_success, num_fail]."""
    lat_w, words_w, ali_w = writers
    where = "DecodeUtteranceLatticeFaster()"
    st = dec.stats(u)
    if st["status"] != 0 or st["num_tokens"] == 0:
        cli.warn("Failed to decode file " + utt, where)
        totals[3] += 1
        return
    if not st["reached_final"]:
        if po["allow-partial"]:
            cli.warn("Outputting partial output for utterance %s since no final-state reached\n" % utt, where)
        else:
            cli.warn("Not producing output for utterance %s since no final-state reached and --allow-partial=false.\n" % utt, where)
            totals[3] += 1
            return
    best = dec.get_best_path(u)
    num_frames = len(best["alignment"])
    words_w.write(utt, best["words"])
    ali_w.write(utt, best["alignment"])
    if word_syms is not None:
        names = []
        for w in best["words"].tolist():
            if w not in word_syms:
                raise cli.KaldiError("Word-id %d not in symbol table." % w)
            names.append(word_syms[w])
        sys.stderr.write(utt + " " + "".join(n + " " for n in names) + "\n")
    like = -(best["graph_cost"] + best["acoustic_cost"])
    acwt = po["acoustic-scale"]
    if po["determinize-lattice"]:         # decoder-wrappers.cc:264-279
        clat = dec.get_compact_lattice(u)
        if not clat["complete"]:
            cli.warn("Determinization finished earlier than the beam for utterance " + utt, where)
        if acwt != 0.0:                    # "We'll write the lattice without acoustic scaling."
            inv = np.float32(1.0 / acwt)
            clat["arc_a"] = (clat["arc_a"] * inv).astype(np.float32)
            clat["final_a"] = (clat["final_a"] * inv).astype(np.float32)
        lat_w.write(utt, clat)
    else:
        lat = dec.get_raw_lattice(u)
        if acwt != 0.0:                    # :283-285
            lat["arc_a"] = (lat["arc_a"] * np.float32(1.0 / acwt)).astype(np.float32)
        lat_w.write(utt, lat)
    cli.log("Log-like per frame for utterance %s is %g over %d frames." % (utt, like / max(num_frames, 1), num_frames), where=where)
    cli.vlog(2, "Cost for utterance %s is %g + %g" % (utt, best["graph_cost"], best["acoustic_cost"]), where)
    totals[0] += like
    totals[1] += num_rows
    totals[2] += 1


def main(argv=None, kind="nnet2"):
    cli = importlib.import_module("old-kaldi-git_amd.kaldi_cli")
    prog = "nnet-latgen-faster" if kind == "nnet2" else "gmm-latgen-faster"
    argv = [prog] + list(sys.argv[1:] if argv is None else argv)
    try:
        return run(cli, argv, kind, prog)
    except cli.KaldiError as e:           # "catch(const std::exception &e) { std::cerr << e.what(); return -1; }"
        sys.stderr.write("ERROR (%s) %s\n" % (prog, e))
        leave_group_after_failure()
        return 255
    finally:
        cli.stop_pipe_helper()


def leave_group_after_failure():
    """A rank that fails before the summary (a feature-dimension mismatch, an unreadable model) still joins the reduction -
    with nothing decoded and one failure - so that its peers, started by a launcher that does not kill siblings, do not wait
    for it until the backend's timeout.  Best effort: errors here are not reported over the original one."""
    try:
        sharding = importlib.import_module("old-kaldi-git_amd.sharding")
        rank, world = sharding.tool_ranks()
        # the backend the surviving ranks use in finish(): gloo for --dry-run, else nccl (which needs this rank's device)
        backend = "gloo" if any(a.startswith("--dry-run") and not a.endswith("=false") for a in sys.argv[1:]) else "nccl"
        if world > 1 and backend == "nccl":
            api = importlib.import_module("old-kaldi-git_amd.api")
            api.select_gpu(int(os.environ.get("LOCAL_RANK", "0")))
        if sharding.init_tool_group(world, backend):
            import torch.distributed as dist
            sharding.reduce_decode_totals(0, 0.0, 0, 1, 0.0, device="cuda" if backend == "nccl" else "cpu")
            dist.barrier()
            dist.destroy_process_group()
    except Exception:   # noqa: BLE001
        pass


def run(cli, argv, kind, prog):
    cli.start_pipe_helper()               # before anything initialises the GPU
    po = cli.ParseOptions(USAGE[kind])
    t_start = time.time()
    register_decoder_options(po)
    po.register("acoustic-scale", 0.1, "Scaling factor for acoustic likelihoods", float)
    po.register("word-symbol-table", "", "Symbol table for words [for debug output]")
    po.register("allow-partial", False, "If true, produce output even if end state was not reached.")
    # not options of the reference binary:
    po.register("batch-frames", 200000, "[MI355X] frames per forward pass / decoder launch", int)
    po.register("reference-order", True, "[MI355X] decode in LatticeFasterDecoder's own iteration order (HashList order, running next_cutoff): the lattices the "
                "reference binary itself writes, bit for bit (the default since round 6)")
    po.register("canonical-order", False, "[MI355X] opt out of --reference-order: the order-independent acceptance rule (a cheaper kernel; same 1-best on every "
                "recipe-like case measured, 0-6 % different raw-lattice arcs; DESIGN.md).  KH_DECODER_ORDER=reference|canonical in the "
                "environment overrides both")
    po.register("gpu", -1, "[MI355X] device ordinal (CuDevice::SelectGpuId); -1: LOCAL_RANK, else 0", int)
    po.register("world", 0, "[MI355X] number of ranks sharing the job (default: WORLD_SIZE, else 1).  Rank r is the recipe's JOB "
                "r + 1: every JOB in the arguments becomes r + 1 (run.pl JOB=1:$nj); a feature table without JOB is taken "
                "round-robin by utterance", int)
    po.register("rank", -1, "[MI355X] this process's rank (default: RANK, else 0)", int)
    po.register("dry-run", False, "[MI355X] read the inputs and the shard, decode nothing, write nothing into the lattices "
                "(multi-rank plumbing test without a GPU)")
    sharding = importlib.import_module("old-kaldi-git_amd.sharding")
    # run.pl replaces JOB in the WHOLE command line (options too: --config=$dir/JOB/decode.conf, --word-symbol-table=...):
    # so --world / --rank are looked at first, the substitution is applied to every argument, then the options are parsed
    pre = cli.ParseOptions("")
    pre.register("world", 0, "", int)
    pre.register("rank", -1, "", int)
    w_opt = r_opt = None
    for a in argv[1:]:
        if a.startswith("--world="):
            w_opt = int(a.split("=", 1)[1])
        elif a.startswith("--rank="):
            r_opt = int(a.split("=", 1)[1])
    rank, world = sharding.tool_ranks(w_opt or 0, -1 if r_opt is None else r_opt)
    raw_args = list(argv)
    if world > 1:
        argv = [argv[0]] + sharding.job_substitute(argv[1:], rank)
    po.read(argv)
    cli.set_program_name(prog)
    if po.num_args() < 4 or po.num_args() > 6:
        po.print_usage()
        return 1
    raw_pos = [a for a in raw_args[1:] if not a.startswith("--")]
    had_job = ["JOB" in a for a in raw_pos] + [False] * 6
    if world > 1:
        # an EMPTY optional output ("" for the words table) is legal and writes nothing: only real outputs need JOB
        outs = [i for i in range(3, po.num_args()) if po.get_arg(i + 1) != ""]
        if not all(had_job[i] for i in outs):
            raise cli.KaldiError("--world=%d: every output table needs JOB in its name (lat.JOB.gz), or the ranks overwrite each other" % world)
    round_robin = world > 1 and not had_job[2]
    model_rx, fst_rx, feats_rspec, lat_wspec = (po.get_arg(i) for i in (1, 2, 3, 4))
    words_wspec, ali_wspec = po.get_opt_arg(5), po.get_opt_arg(6)
    fst_table = cli.classify_rspecifier(fst_rx)[0] is not None   # ":103 ClassifyRspecifier(fst_in_str) == kNoRspecifier"
    if fst_table and round_robin:
        raise cli.KaldiError("--world=%d with a table of decoding graphs: give every rank its own graph table (JOB in its name)" % world)

    kio = importlib.import_module("old-kaldi-git_amd.kaldi_io")
    # model (":76-83 Input ki(model_in_filename, &binary); trans_model.Read; am.Read")
    if kind == "nnet2":
        tm, (comps, priors) = cli.read_kaldi_object(model_rx, lambda s, b: (kio.read_transition_model(s, b), kio.read_am_nnet(s, b)))
    else:
        tm, am = cli.read_kaldi_object(model_rx, lambda s, b: (kio.read_transition_model(s, b), kio.read_am_diag_gmm(s, b)))
    determinize = po["determinize-lattice"]
    lat_w = cli.TableWriter(lat_wspec, "compact_lattice" if determinize else "lattice")
    if not lat_w.is_open():
        raise cli.KaldiError("Could not open table for writing lattices: " + lat_wspec)
    words_w, ali_w = cli.TableWriter(words_wspec, "int32_vector"), cli.TableWriter(ali_wspec, "int32_vector")
    word_syms = cli.read_symbol_table(po["word-symbol-table"]) if po["word-symbol-table"] != "" else None

    def check_graph(graph, what):
        graph["tid2pdf"] = tm["tid2pdf"]
        if int(graph["ilabel"].max(initial=0)) >= len(tm["tid2pdf"]):
            raise cli.KaldiError("%s has transition-ids the model does not define" % what)
        return graph
    if fst_table:     # ":140-142 SequentialTableReader<fst::VectorFstHolder> fst_reader; RandomAccessBaseFloatCuMatrixReader feature_reader"
        graph = None
        fst_reader = cli.SequentialTableReader(fst_rx, "fst")
        reader = cli.RandomAccessTableReader(feats_rspec, "matrix")
    else:
        graph = check_graph(cli.read_kaldi_object(fst_rx, lambda s, b: kio.read_fst(s)), "HCLG")
        reader = cli.SequentialTableReader(feats_rspec, "matrix")

    if po["dry-run"]:
        n_utts = n_frames = 0
        if fst_table:
            reader = ((utt, reader.value(utt)) for utt, _ in fst_reader if reader.has_key(utt))
        for k, (utt, m) in enumerate(reader):
            if round_robin and k % world != rank:
                continue
            n_utts += 1
            n_frames += m.shape[0]
        lat_w.close(), words_w.close(), ali_w.close()
        return finish(cli, sharding, world, "gloo", time.time() - t_start, [0.0, n_frames, n_utts, 0], lat_wspec, True)

    # ---- the GPU from here on
    import torch
    api = importlib.import_module("old-kaldi-git_amd.api")
    api.select_gpu(po["gpu"] if po["gpu"] >= 0 else int(os.environ.get("LOCAL_RANK", "0")))
    acwt = po["acoustic-scale"]
    if kind == "nnet2":
        nnet = api.Nnet(comps, priors)
        input_dim = nnet.input_dim()

        def score(feats, off):     # DecodableAmNnet (decodable-am-nnet.h:60-69), batched
            return nnet.compute(feats, off, pad_input=True, epilogue=True, prob_scale=acwt)[0]
    else:
        gconsts, _ = api.gmm_compute_gconsts(am["weights"], am["means_invvars"], am["inv_vars"])   # DiagGmm::Read :755
        gmm = api.AmDiagGmm(gconsts, am["means_invvars"], am["inv_vars"], am["pdf_offsets"])
        input_dim = am["dim"]

        def score(feats, off):     # DecodableAmDiagGmmScaled::LogLikelihood (decodable-am-diag-gmm.h:142-145)
            ll = gmm.pdf_log_likelihoods(feats)
            api.scale(ll, acwt)
            return ll
    cfg = decoder_config(api, po)
    det_opts = determinize_options(api, po, tm)
    totals = [0.0, 0, 0, 0]     # tot_like, frame_count, num_success, num_fail
    state = dict(dec=None, max_batch=0, max_frames=0)
    if fst_table:
        # :140-176 a different graph for every utterance: "LatticeFasterDecoder decoder(fst_reader.Value(), config)" per
        # utterance, features looked up by the graph's key.  (Such graphs are small - an utterance's own alignment or
        # rescoring graph - so there is one device graph, one decoder object and one launch per utterance; the batched path
        # below is for the shared HCLG.)
        for utt, g in fst_reader:
            if not reader.has_key(utt):
                cli.warn("Not decoding utterance %s because no features available." % utt)
                totals[3] += 1
                continue
            m = reader.value(utt)
            if m.shape[0] == 0:
                cli.warn("Zero-length utterance: " + utt)
                totals[3] += 1
                continue
            if m.shape[1] != input_dim:
                raise cli.KaldiError("feature dimension %d of %s does not match the model's input %d" % (m.shape[1], utt, input_dim))
            off = np.array([0, m.shape[0]], np.int32)
            loglikes = score(torch.from_numpy(np.ascontiguousarray(m, np.float32)).cuda(), off)
            dec = api.LatticeFasterDecoder(api.Fst(check_graph(g, "the graph of " + utt)), cfg, max_batch=1, max_frames=int(m.shape[0]),
                                           exact_reference_order=bool(po["reference-order"]) and not bool(po["canonical-order"]))
            dec.set_determinize(determinize, **det_opts)
            dec.decode(loglikes, off)
            dec.prepare()
            write_utterance(cli, api, dec, 0, utt, m.shape[0], po, (lat_w, words_w, ali_w), word_syms, totals, prog)
        ok = lat_w.close()
        words_w.close()
        ali_w.close()
        return finish(cli, sharding, world, "nccl", time.time() - t_start, totals, lat_wspec, ok)
    fst = api.Fst(graph)

    def flush(batch):
        if not batch:
            return
        off = np.concatenate([[0], np.cumsum([len(m) for _, m in batch])]).astype(np.int32)
        feats = torch.from_numpy(np.concatenate([m for _, m in batch], 0).astype(np.float32)).cuda()
        loglikes = score(feats, off)
        max_T = int(np.diff(off).max())
        if state["dec"] is None or len(batch) > state["max_batch"] or max_T > state["max_frames"]:
            state["max_batch"], state["max_frames"] = max(len(batch), 64), max(max_T, 1024)
            state["dec"] = api.LatticeFasterDecoder(fst, cfg, max_batch=state["max_batch"], max_frames=state["max_frames"],
                                                        exact_reference_order=bool(po["reference-order"]) and not bool(po["canonical-order"]))
            state["dec"].set_determinize(determinize, **det_opts)
        dec = state["dec"]
        dec.decode(loglikes, off)
        dec.prepare()
        for u, (utt, m) in enumerate(batch):
            write_utterance(cli, api, dec, u, utt, len(m), po, (lat_w, words_w, ali_w), word_syms, totals, prog)

    batch, frames = [], 0
    for k, (utt, m) in enumerate(reader):
        if round_robin and k % world != rank:
            continue
        if m.shape[0] == 0:
            cli.warn("Zero-length utterance: " + utt)
            totals[3] += 1
            continue
        if m.shape[1] != input_dim:
            raise cli.KaldiError("feature dimension %d of %s does not match the model's input %d" % (m.shape[1], utt, input_dim))
        batch.append((utt, m))
        frames += m.shape[0]
        if frames >= po["batch-frames"]:
            flush(batch)
            batch, frames = [], 0
    flush(batch)
    ok = lat_w.close()
    words_w.close()
    ali_w.close()
    return finish(cli, sharding, world, "nccl", time.time() - t_start, totals, lat_wspec, ok)


def finish(cli, sharding, world, backend, elapsed, totals, lat_wspec, ok):
    """The binary's summary lines (:179-186) for this rank's shard - its own log, like a recipe job's - and, when the ranks
    were started by a launcher, the totals over all of them from rank 0 (one all-reduce of four numbers + MAX of the time)."""
    tot_like, frame_count, num_success, num_fail = totals
    cli.log("Time taken %gs: real-time factor assuming 100 frames/sec is %g" % (elapsed, elapsed * 100.0 / max(frame_count, 1)))
    cli.log("Done %d utterances, failed for %d" % (num_success, num_fail))
    cli.log("Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(frame_count, 1), frame_count))
    if sharding.init_tool_group(world, backend):
        import torch.distributed as dist
        tot = sharding.reduce_decode_totals(frame_count, tot_like, num_success, num_fail, elapsed, device="cuda" if backend == "nccl" else "cpu")
        if dist.get_rank() == 0:
            cli.log("All %d ranks: done %d utterances, failed for %d; overall log-likelihood per frame is %g over %d frames; "
                    "%g frames/s, real-time factor %g" % (world, tot["num_success"], tot["num_fail"], tot["loglike_per_frame"], int(tot["frames"]),
                                                         tot["frames_per_sec"], tot["rtf"]))
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise cli.KaldiError("error closing the lattice table " + lat_wspec)
    return 0 if num_success != 0 else 1


if __name__ == "__main__":
    sys.exit(main())
