"""Secondary legs of bench.py (configs 2, 3, 5 and the online mode of
config 4), each a bounded measurement on cuda:0 with inputs resident in HBM.  run_all()
returns a dict that bench.py attaches to its JSON line as "secondary" — the driver-visible
record of the numbers DESIGN.md quotes.  Product API only (no oracle)."""
import importlib
import os
import sys
import time

import numpy as np

PKG = "old-kaldi-git_amd"


def _timeit(fn, sync, reps=3):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def gmm_cfg2(api, torch, T=200_000):
    """Config 2 (egs/rm tri1): DiagGmm::LogLikelihoods + per-pdf LogSumExp, 39 dims,
    1800 pdfs / 9000 Gaussians, T frames -> the frame x pdf matrix."""
    W = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(1)
    am = W.make_am_gmm(rng, 1800, 9000, 39)
    mi, iv = W.gmm_inv_params(am)
    g, _ = api.gmm_compute_gconsts(am["weights"], mi, iv)
    gmm = api.AmDiagGmm(g, mi, iv, am["pdf_offsets"])
    x = torch.from_numpy(rng.standard_normal((T, 39)).astype(np.float32)).cuda()
    out = torch.empty((T, 1800), dtype=torch.float32, device="cuda")
    sync = lambda: (api.synchronize(), torch.cuda.synchronize())
    dt = _timeit(lambda: gmm.pdf_log_likelihoods(x, out=out), sync)
    flops = T * 9000 * (2 * 78 + 20)
    return {"workload": "rm_tri1_synthetic: %d frames x 39 dims, 1800 pdfs / 9000 Gaussians" % T,
            "ms_per_call": dt * 1e3, "frames_per_s": T / dt, "algorithmic_tflops": flops / dt / 1e12,
            "bound": "mfma", "peak_tflops": 157.3, "frac": flops / dt / 1e12 / 157.3,
            "hbm_bytes_algorithmic": T * (39 * 4 + 1800 * 4) + 2.8e6}


def nnet_cfg3(api, torch, T=120_000):
    """Config 3 (egs/wsj nnet5d): p-norm network forward, 40 -> 360 -> 4 x (2000/400) -> 8000 -> 3400."""
    W = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(2)
    net, priors = W.wsj_nnet5d(rng)
    nnet = api.Nnet(net, priors)
    lens = np.full(120, T // 120, np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = torch.from_numpy(rng.standard_normal((int(off[-1]), 40)).astype(np.float32)).cuda()
    out = torch.empty((int(off[-1]), 3400), dtype=torch.float32, device="cuda")
    sync = lambda: (api.synchronize(), torch.cuda.synchronize())

    def fwd():
        for u0 in range(0, 120, 60):
            sub = (off[u0:u0 + 61] - off[u0]).astype(np.int32)
            nnet.compute(x[off[u0]:off[u0 + 60]], sub, True, epilogue=True, prob_scale=0.1, out=out[off[u0]:off[u0 + 60]])
    dt = _timeit(fwd, sync)
    flops = 2.0 * (360 * 360 + 360 * 2000 + 3 * 400 * 2000 + 400 * 8000) * int(off[-1])
    return {"workload": "wsj_nnet5d_synthetic forward: %d frames" % int(off[-1]), "ms_per_call": dt * 1e3,
            "frames_per_s": int(off[-1]) / dt, "algorithmic_tflops": flops / dt / 1e12, "bound": "mfma",
            "peak_tflops": 157.3, "frac": flops / dt / 1e12 / 157.3}


def cfg3_workload(n_utts, graph_states=2_000_000, seed=33):
    """Config 3 (egs/wsj nnet5d decode) at its sizes: the wsj p-norm network, an HCLG-structured graph
    with its 3400 pdfs, utterances sampled from the graph.  The wsj network has no constant input
    part to carry a per-frame target (bench.py's nnet_a workload uses the iVector dimensions for
    that), so the decoder's scores are synthetic rows that follow the sampled paths (competitors
    N(-0.37, 0.28), the path's pdf 0.5 +- 0.3 after the acoustic scale: the figures the nnet_a workload
    produces) and the forward pass is timed on random features of the same shape.  Pure numpy."""
    W = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(seed)
    net, priors = W.wsj_nnet5d(rng)
    g = W.make_hclg_structured(np.random.default_rng(seed + 2), graph_states, net[-1]["output_dim"])
    urng = np.random.default_rng(seed + 4)
    lens = np.sort(W.utterance_lengths(urng, n_utts))[::-1].copy()
    seqs = W.sample_paths(urng, g, lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = urng.standard_normal((int(off[-1]), 40)).astype(np.float32)

    def scores(u):
        q = seqs[u]
        r = np.random.default_rng(seed * 1000 + u)
        x = (r.standard_normal((len(q), net[-1]["output_dim"])) * 0.28 - 0.37).astype(np.float32)
        x[np.arange(len(q)), q] = (0.5 + 0.3 * r.standard_normal(len(q))).astype(np.float32)
        return x
    return net, priors, g, feats, off, scores


def decode_cfg3(api, torch, n_utts=333):
    """Config 3 end to end: forward + LatticeFasterDecoder (beam 15, max-active 7000, lattice-beam 8,
    acwt 0.1: steps/nnet2/decode.sh) of an eval92-sized set (333 utterances) in one launch."""
    net, priors, g, feats, off, scores = cfg3_workload(n_utts)
    nnet = api.Nnet(net, priors)
    fst = api.Fst(g)
    cfg = api.decoder_config(beam=15.0, max_active=7000, min_active=200, lattice_beam=8.0)
    dec = api.LatticeFasterDecoder(fst, cfg, max_batch=n_utts, max_frames=int(np.diff(off).max()))
    x = torch.from_numpy(feats).cuda()
    ll = torch.empty((int(off[-1]), net[-1]["output_dim"]), dtype=torch.float32, device="cuda")
    sc = torch.empty_like(ll)
    for u in range(n_utts):
        sc[off[u]:off[u + 1]] = torch.from_numpy(scores(u)).cuda()

    def step():
        u0 = 0
        while u0 < n_utts:
            u1 = u0 + 1
            while u1 < n_utts and off[u1 + 1] - off[u0] <= 60000:
                u1 += 1
            nnet.compute(x[off[u0]:off[u1]], (off[u0:u1 + 1] - off[u0]).astype(np.int32), True, epilogue=True, prob_scale=0.1,
                         out=ll[off[u0]:off[u1]])
            u0 = u1
        dec.decode(sc, off)
        dec.prepare()
    dt = _timeit(step, lambda: torch.cuda.synchronize(), reps=2)
    st = [dec.stats(u) for u in range(0, n_utts, 37)]
    arcs = sum(s["num_links"] for s in st) / max(1, sum(s["num_frames"] for s in st))
    return {"workload": "wsj_nnet5d_structured: %d utterances, %d frames, graph %d states / %d arcs, 3400 pdfs; forward on random "
                        "features, decoder scores synthetic along sampled paths" % (n_utts, int(off[-1]), int(g["num_states"]), int(g["arc_offsets"][-1])),
            "ms_per_step": dt * 1e3, "frames_per_s": int(off[-1]) / dt, "kernel_ms": dec.last_kernel_ms(),
            "lattice_arcs_per_frame": arcs, "ok": int(sum(dec.stats(u)["reached_final"] for u in range(n_utts)))}


def lattice_fb_cfg5(api, torch, N=256, T=400):
    """Config 5 (egs/swbd MMI / sMBR): denominator lattices of N utterances (raw lattices of a
    structured-graph decode).  (a) kh_lattice_forward_backward alone: upload, device
    preparation (levels, incoming-arc CSR, validation), sweeps, download;  (b) the whole
    NnetDiscriminativeUpdater::Propagate + LatticeComputations pipeline
    (nnet-compute-discriminative.cc:150-321) per criterion: forward of a wsj-sized p-norm
    network, Lookup, lattice rescoring, forward-backward, posterior merging, CompObjfAndDeriv."""
    import ctypes as C
    capi = importlib.import_module(PKG + ".capi")
    W = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(5)
    P = 2000
    g = W.make_hclg_structured(rng, 1_000_000, P)
    seqs = W.sample_paths(rng, g, [T] * N)
    lls = []
    for q in seqs:
        x = (rng.standard_normal((T, P)) * 0.28 - 0.37).astype(np.float32)
        x[np.arange(T), q] = (0.5 + 0.3 * rng.standard_normal(T)).astype(np.float32)
        lls.append(x)
    flat = torch.from_numpy(np.concatenate(lls)).cuda()
    cfg = api.decoder_config(beam=13.0, max_active=7000, min_active=200, lattice_beam=8.0)
    dec = api.LatticeFasterDecoder(api.Fst(g), cfg, max_batch=N, max_frames=T)
    dec.decode(flat, (np.arange(N + 1) * T).astype(np.int32))
    raw = [dec.get_raw_lattice(u) for u in range(N)]
    lats = [api.lattice_to_csr(L) for L in raw]
    arcs = sum(len(L["arc_ilabel"]) for L in lats)
    states = sum(L["n_states"] for L in lats)
    alis = [dec.get_best_path(u)["alignment"].astype(np.int32) for u in range(N)]
    del dec, flat
    ntid = len(g["tid2pdf"]) - 1
    t2ph = np.concatenate([[0], 1 + (np.arange(ntid) // 6) % 40]).astype(np.int32)
    res = {"workload": "%d lattices x %d frames (structured graph, lattice-beam 8): %d states, %d arcs (%.1f arcs/frame)"
                       % (N, T, states, arcs, arcs / (N * T))}
    # (0) the step right behind the decoder: DeterminizeLatticePhonePrunedWrapper (decoder-wrappers.cc:264-274), host threads
    t0 = time.perf_counter()
    clats = api.determinize_lattices(raw, 8.0)
    dt = time.perf_counter() - t0
    res["determinize"] = {"ms_per_batch": dt * 1e3, "frames_per_s": N * T / dt, "raw_arcs": arcs,
                          "compact_arcs": int(sum(len(c["arc_src"]) for c in clats)),
                          "complete": int(sum(1 for c in clats if c["complete"])), "where": "host threads (as the reference)"}
    del raw, clats
    # (a) the C call alone (upload of the caller's arrays, device preparation, both sweeps, download), at the batch of the
    # recipe (256) and at 2048 lattices (the same lattices 8 times over: the call is bound by the ~levels sequential steps
    # of a lattice x one workgroup each, so throughput comes from the number of lattices in flight)
    ip, fp, dp = capi.c_int32_p, capi.c_float_p, capi.c_double_p

    def fb_leg(batch):
        n, soff, aoff, il, ns, gg, aa, fin = api._cat_lattices(batch)
        post = np.empty(len(il), np.float32)
        tot, ac = np.empty(n), np.empty(n)

        def call():
            api.check(api.lib().kh_lattice_forward_backward(
                n, soff.ctypes.data_as(ip), aoff.ctypes.data_as(capi.c_int64_p), il.ctypes.data_as(ip), ns.ctypes.data_as(ip),
                gg.ctypes.data_as(fp), aa.ctypes.data_as(fp), fin.ctypes.data_as(fp), post.ctypes.data_as(fp),
                tot.ctypes.data_as(dp), ac.ctypes.data_as(dp), None))
        dt = _timeit(call, lambda: None, reps=3)
        split = api.lattice_last_timings()       # HIP events around upload / device preparation / sweeps / download
        na = len(il)
        # the same batch KEPT ON THE DEVICE (kh_lattice_batch_*: one upload + preparation for every computation on it)
        t0 = time.perf_counter()
        batch_h = api.LatticeBatch((n, soff, aoff, il, ns, gg, aa, fin))
        create_ms = (time.perf_counter() - t0) * 1e3
        dt_res = _timeit(lambda: batch_h.forward_backward(), lambda: None, reps=3)
        res_split = api.lattice_last_timings()
        dev_post = torch.empty(na, dtype=torch.float32, device="cuda")
        dt_dev = _timeit(lambda: batch_h.forward_backward_device(dev_post), lambda: None, reps=3)   # posteriors stay in HBM
        del batch_h, dev_post
        # algorithmic bytes: 32 B per arc per sweep (next state, graph + acoustic cost, the incoming-arc entry, the 8-byte
        # alpha / beta of the other end), two sweeps - over the SWEEP KERNEL's own duration
        k_s = max(res_split["sweeps_ms"], 1e-6) * 1e-3
        return {"lattices": n, "ms_per_batch": dt * 1e3, "arcs_per_s": na / dt, "frames_per_s": n * T / dt,
                "call_split_ms": {k: round(v, 3) for k, v in split.items()},
                "resident_batch": {"create_ms": create_ms, "forward_backward_ms": dt_res * 1e3, "arcs_per_s": na / dt_res,
                                   "sweeps_ms": res_split["sweeps_ms"], "download_ms": res_split["download_ms"],
                                   "forward_backward_device_out_ms": dt_dev * 1e3, "arcs_per_s_device_out": na / dt_dev},
                "roofline": {"bound": "hbm", "achieved": na * 64 / k_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": na * 64 / k_s / 8e12, "algorithmic_bytes": na * 64, "kernel_ms": res_split["sweeps_ms"],
                             "note": "the sweep kernel alone (HIP events); the kernel is bound by the dependent chain of a "
                                     "lattice - ~420 levels x 2 sweeps x one double-precision LogAdd latency - not by bytes: "
                                     "throughput comes from the lattices in flight (2048: see the other entry)"}}

    def levels(L):   # longest distance from the start state = the sequential steps of one sweep
        off, nxt = np.asarray(L["arc_offsets"]), np.asarray(L["arc_nextstate"])
        lv = np.zeros(L["n_states"], np.int64)
        for s_ in range(L["n_states"]):
            d = nxt[off[s_]:off[s_ + 1]]
            if len(d):
                np.maximum.at(lv, d, lv[s_] + 1)
        return int(lv.max()) + 1
    lv = [levels(L) for L in lats[:16]]
    res["kh_lattice_forward_backward"] = fb_leg(lats)
    res["kh_lattice_forward_backward"]["levels_per_lattice"] = {"mean_of_16": float(np.mean(lv)), "max_of_16": int(max(lv))}
    res["kh_lattice_forward_backward_2048"] = fb_leg(lats * (2048 // N))
    # (b) the discriminative pipeline
    net, _ = W.make_pnorm_net(rng, feat_dim=40, splice=4, const_dim=0, pnorm_in=2000, pnorm_out=400, n_hidden=4,
                              n_mix=2 * P, n_pdf=P, final_scale=4.0)
    priors = np.full(P, 1.0 / P, np.float32)
    nnet = api.Nnet(net, priors)
    Lc, Rc = nnet.left_context(), nnet.right_context()
    egs = [dict(feats=torch.from_numpy(rng.standard_normal((T + Lc + Rc, 40)).astype(np.float32)).cuda(), num_ali=alis[u],
                den_lat=lats[u], weight=1.0) for u in range(N)]
    for crit in ("mmi", "smbr"):
        fn = lambda: api.discriminative_lattice_computations(nnet, priors, g["tid2pdf"], egs, criterion=crit, acoustic_scale=0.1,
                                                             drop_frames=True, tid2phone=t2ph, silence_phones=[1, 2])
        dt = _timeit(fn, lambda: torch.cuda.synchronize(), reps=2)
        res["pipeline_" + crit] = {"ms_per_batch": dt * 1e3, "frames_per_s": N * T / dt, "arcs_per_s": arcs / dt}
    # a trainer's loop: batch i + 1 begun (its forward pass queued) before batch i is ended - the forward pass runs beside the
    # previous batch's lattice steps (kh_discriminative_lattice_computations_begin / _end)
    def piped(k=6):
        prev = None
        for _ in range(k):
            c = api.discriminative_lattice_computations(nnet, priors, g["tid2pdf"], egs, criterion="mmi", acoustic_scale=0.1,
                                                        drop_frames=True, tid2phone=t2ph, silence_phones=[1, 2], begin=True)
            if prev is not None:
                prev.end()
            prev = c
        prev.end()
    try:
        dt = _timeit(piped, lambda: torch.cuda.synchronize(), reps=2) / 6
        res["pipeline_mmi_two_in_flight"] = {"ms_per_batch": dt * 1e3, "frames_per_s": N * T / dt, "arcs_per_s": arcs / dt}
    except api.KhError as e:     # (KH_LATTICE_ONE_STREAM=1: the halves need the second stream)
        res["pipeline_mmi_two_in_flight"] = {"error": str(e)[:120]}
    # the same with the lattices of the batch concatenated ahead (a data loader's job), and its parts
    cat = api.cat_lattices(lats)
    fn = lambda: api.discriminative_lattice_computations(nnet, priors, g["tid2pdf"], egs, criterion="mmi", acoustic_scale=0.1,
                                                         drop_frames=True, tid2phone=t2ph, silence_phones=[1, 2], den_lats=cat)
    dt = _timeit(fn, lambda: torch.cuda.synchronize(), reps=2)
    res["pipeline_mmi_preconcatenated"] = {"ms_per_batch": dt * 1e3, "frames_per_s": N * T / dt}
    feats = torch.cat([e["feats"] for e in egs], 0)
    foff = (np.arange(N + 1) * (T + Lc + Rc)).astype(np.int32)
    dt = _timeit(lambda: nnet.compute(feats, foff, pad_input=False), lambda: torch.cuda.synchronize(), reps=2)
    res["pipeline_parts"] = {"forward_ms": dt * 1e3}
    t0 = time.perf_counter()
    api.cat_lattices(lats)
    res["pipeline_parts"]["concatenate_lattices_host_ms"] = (time.perf_counter() - t0) * 1e3
    return res


def ivector_f3(api, torch, n_utts=2620, mean_len=740):
    """Online iVector extraction at the reference's default dimensions (online-ivector-feature.h:102-107):
    40-dim base features, +-3 splice, LDA to 40, 512-Gaussian UBM, 100-dim iVector, period 10."""
    W = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(4)
    m = W.make_ivector_extractor(rng)
    ext = api.OnlineIvectorExtractor(m)
    lens = np.clip(rng.gamma(4.0, mean_len / 4.0, n_utts).astype(np.int64), 100, 3500)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = torch.from_numpy(rng.standard_normal((int(off[-1]), 40)).astype(np.float32)).cuda()
    out = torch.empty((int(off[-1]), 100), dtype=torch.float32, device="cuda")
    sync = lambda: (api.synchronize(), torch.cuda.synchronize())
    dt = _timeit(lambda: ext.extract(x, off, out=out), sync, reps=2)
    return {"workload": "%d utterances, %d frames, UBM 512 x 40, iVector dim 100, period 10, 15 CG iterations" % (n_utts, int(off[-1])),
            "ms_per_call": dt * 1e3, "frames_per_s": int(off[-1]) / dt, "longest_utterance_frames": int(lens.max())}


def online2_cfg4(api, torch, workload=None, streams=256, chunks=(5, 50), only_persistent=False):
    """Config 4 in its default mode (online2-wav-nnet2-latgen-faster --online=true, :213-262) as a serving loop: `streams`
    concurrent utterances advance in lockstep, one chunk of audio per step (--chunk-length 0.05 s = 5 frames, and 0.5 s).
    A step = what one chunk triggers for every live stream: the chunk's feature rows (MFCC + iVector: already in HBM, their
    kernels have their own leg) enter DecodableNnet2Online, ComputeForFrame runs the network over the rows that became
    ready (NumFramesReady() holds back the right context until InputFinished()), AdvanceDecoding consumes them in one
    launch of the online decode kernel.  Reported: frames/s over the whole set and the latency of a step (= what a caller
    waits between handing over a chunk and being able to ask for a partial result)."""
    if workload is None:
        bench = importlib.import_module("bench")
        net, priors, g, protos = bench.build_model_and_graph(3456, 2_000_000, False)
        feats, off = bench.build_utterances(3456, 0, 3 * streams, net, g, protos, False)
        dcfg, acwt = bench.DECODE_CFG, bench.ACWT
    else:
        net, priors, g, feats, off, dcfg, acwt = workload
    n = min(streams, len(off) - 1)
    lens = np.diff(off)[:n].astype(np.int64)
    max_t = int(np.diff(off).max())
    nnet = api.Nnet(net, priors)
    fst = api.Fst(g)
    dec = api.LatticeFasterOnlineDecoder(fst, api.decoder_config(**dcfg), num_streams=n, max_frames=max_t)
    n_serve = min(len(off) - 1, 3 * n)            # the continuous loop serves up to three utterances per slot
    off = np.asarray(off[:n_serve + 1])
    x_all = torch.from_numpy(np.ascontiguousarray(feats[:off[-1]])).cuda()
    x = x_all[:off[n]]
    res = {"workload": "%d streams of the headline utterance set (%d frames, longest %d), graph %d states; features resident in HBM"
                       % (n, int(off[n]), max_t, int(g["num_states"]))}
    pipe = api.OnlineNnet2Pipeline(nnet, dec, max_frames=max_t, acoustic_scale=acwt, pad_input=True, max_nnet_batch_size=256)
    n_utts = len(off) - 1
    all_lens = np.diff(off).astype(np.int64)
    def progress(msg):
        if os.environ.get("BENCH_VERBOSE"):
            print("[bench online2] " + msg, file=sys.stderr, flush=True)

    for c in (() if only_persistent else chunks):   # (only_persistent: tools/stress_serving.py runs the serving legs alone)
        progress("launch-per-step loop, chunk %d" % c)
        # CONTINUOUS serving, the step as ONE library call (kh_online_nnet2_step): `n` stream slots; a slot whose utterance
        # has been decoded is finalized and takes the next utterance of the set (InitDecoding), until the set is used up.
        # Throughput and latency are those of the steps in which every slot was busy (the drain at the end is a property of
        # a finite test set, not of a serving loop).
        lat = []
        for rep in range(2):           # the first pass warms the allocator / the kernels
            slot_utt = np.arange(n)                    # utterance a slot works on
            next_utt = n
            given = np.zeros(n, np.int64)
            pipe.reset(list(range(n)))
            lat, n_done = [], 0
            torch.cuda.synchronize()
            live = np.arange(n)
            while len(live):
                t0 = time.perf_counter()
                u = slot_utt[live]
                cnt = np.minimum(c, all_lens[u] - given[live])
                fin = given[live] + cnt == all_lens[u]
                done = pipe.step(live, x_all, off[u].astype(np.int64) + given[live], cnt, fin)
                t_adv = time.perf_counter() - t0
                given[live] += cnt
                ended = live[done >= all_lens[u]]
                if len(ended):
                    dec.finalize_decoding(ended)
                    n_done += len(ended)
                    fresh = []
                    for s in ended:            # the slot's next utterance
                        if next_utt < n_utts and all_lens[next_utt] <= max_t:
                            slot_utt[s], given[s] = next_utt, 0
                            next_utt += 1
                            fresh.append(s)
                        else:
                            slot_utt[s] = -1
                    if fresh:
                        pipe.reset(fresh)
                    live = np.array([s for s in live if slot_utt[s] >= 0], np.int64)
                api.synchronize()
                lat.append((time.perf_counter() - t0, len(u), int(cnt.sum()), t_adv, len(ended)))
        ms = np.array([l[0] for l in lat]) * 1e3
        full = np.array([l[1] == n for l in lat])
        fr = np.array([l[2] for l in lat])
        adv = np.array([l[3] for l in lat]) * 1e3
        n_end = np.array([l[4] for l in lat])
        res["chunk_%d_frames" % c] = {
            "chunk_seconds": c * 0.01, "steps": len(lat), "steps_with_every_slot_busy": int(full.sum()), "utterances_served": int(n_done),
            "frames_per_s": float(fr[full].sum() / (ms[full].sum() * 1e-3)) if full.any() else None,
            "real_time_streams_sustained": float(fr[full].sum() / (ms[full].sum() * 1e-3) / 100.0) if full.any() else None,
            # the chunk's own work (features in -> every slot advanced): what a caller waits for a partial result
            "advance_ms": {"mean": float(adv[full].mean()) if full.any() else None, "p50": float(np.percentile(adv[full], 50)) if full.any() else None,
                           "p95": float(np.percentile(adv[full], 95)) if full.any() else None},
            # ... and the end-of-utterance work done inside the same steps (FinalizeDecoding of the slots that ended +
            # InitDecoding of their next utterance; synchronous here, so it is charged to every slot's step)
            "end_of_utterance_ms_per_utterance": float((ms[full] - adv[full]).sum() / max(1, n_end[full].sum())) if full.any() else None,
            "step_latency_ms": {"mean": float(ms[full].mean()) if full.any() else None,
                                "p50": float(np.percentile(ms[full], 50)) if full.any() else None,
                                "p95": float(np.percentile(ms[full], 95)) if full.any() else None, "max": float(ms.max())},
            "step": "kh_online_nnet2_step (one library call per chunk; finished slots finalized and given the next utterance "
                    "inside the timed step)"}
    R = nnet.right_context()
    ahead = int(os.environ.get("KH_BENCH_SERVE_AHEAD", "1"))

    def serve_leg(c, tag, note):
        # The same service through the decoder's PERSISTENT kernel (kh_online_nnet2_serve_*): no decode launch per chunk, one
        # resident workgroup per stream.  A stream is handed its next chunk while at most ONE chunk of its frames is still
        # waiting to be decoded (the next chunk of audio arrives while the previous one is being worked on) - a
        # stream that is pruning (every prune_interval frames) or finalizing simply misses turns instead of holding the
        # step up for everybody; FinalizeDecoding is requested asynchronously and the slot takes its next utterance when it
        # is acknowledged.  step_call_ms = the host's step (features in -> scores published); chunk_latency_ms = per stream
        # and chunk, from handing the chunk over to NumFramesDecoded() having reached its frames (polled once per turn).
        progress("persistent kernel, chunk %d%s" % (c, tag))
        pipe.serve_start()
        for rep in range(2):
            slot_utt = np.arange(n)
            next_utt = n
            given = np.zeros(n, np.int64)
            state = np.zeros(n, np.int8)           # 0 running, 1 FinalizeDecoding in flight, 2 no utterance left
            out = [[] for _ in range(n)]           # per slot: (time handed over, NumFramesDecoded() that completes the chunk)
            pipe.reset(list(range(n)))
            all_slots = np.arange(n)
            calls, chunk_lat, last_lat, marks = [], [], [], []
            n_done = frames_done = 0
            t_begin = time.perf_counter()
            while (state != 2).any():
                if time.perf_counter() - t_begin > 180.0:
                    pipe.serve_stop()
                    raise RuntimeError("serving leg: no progress within 180 s (%d utterances served)" % n_done)
                dcd, busy = pipe.serve_poll(all_slots)
                now = time.perf_counter()
                u = np.maximum(slot_utt, 0)
                want = np.where(given >= all_lens[u], all_lens[u], np.maximum(0, given - R))   # frames submitted to the decoder
                for sl in np.nonzero(~busy)[0]:
                    q = out[sl]
                    while q and dcd[sl] >= q[0][1]:
                        # (an utterance's LAST chunk is only seen complete once FinalizeDecoding has been acknowledged - the
                        # stream reports busy until then - so its figure includes that call: kept apart)
                        t_h, _, was_last = q.pop(0)
                        (last_lat if was_last else chunk_lat).append(now - t_h)
                for sl in np.nonzero((state == 1) & ~busy)[0]:       # finalized: the slot's next utterance
                    n_done += 1
                    frames_done += int(all_lens[slot_utt[sl]])
                    out[sl] = []
                    if next_utt < n_utts and all_lens[next_utt] <= max_t:
                        slot_utt[sl], given[sl], state[sl] = next_utt, 0, 0
                        want[sl] = dcd[sl] = 0
                        next_utt += 1
                        pipe.reset([int(sl)])
                    else:
                        slot_utt[sl], state[sl] = -1, 2
                marks.append((now, n_done, frames_done, int((state != 2).sum())))
                # a stream takes its next chunk while at most `ahead` chunks of frames are still waiting to be decoded
                ready = np.nonzero((state == 0) & (want - dcd <= ahead * c))[0]
                if len(ready) == 0:
                    time.sleep(5e-5)
                    continue
                t0 = time.perf_counter()
                uu = slot_utt[ready]
                cnt = np.minimum(c, all_lens[uu] - given[ready])
                fin = given[ready] + cnt == all_lens[uu]
                pipe.step(ready, x_all, off[uu].astype(np.int64) + given[ready], cnt, fin)
                given[ready] += cnt
                tgt = np.where(fin, all_lens[uu], np.maximum(0, given[ready] - R))
                for sl, tg, fl in zip(ready.tolist(), tgt.tolist(), fin.tolist()):
                    out[sl].append((t0, tg, bool(fl)))
                ended = ready[fin]
                if len(ended):
                    pipe.serve_finalize(ended)
                    state[ended] = 1
                calls.append((time.perf_counter() - t0, len(ready), int(cnt.sum())))
            total = time.perf_counter() - t_begin
        pipe.serve_stop()
        marks = np.array(marks)
        full = marks[:, 3] == n                      # every slot had an utterance
        if full.any():
            i0, i1 = np.nonzero(full)[0][[0, -1]]
            # frames of the utterances that COMPLETED inside the window in which every slot was busy
            fps = float((marks[i1, 2] - marks[i0, 2]) / max(1e-9, marks[i1, 0] - marks[i0, 0]))
        else:
            fps = None
        cm = np.array([x_[0] for x_ in calls]) * 1e3
        cl = np.array(chunk_lat) * 1e3
        ll_ = np.array(last_lat) * 1e3 if last_lat else np.zeros(1)
        res["chunk_%d_frames_persistent%s" % (c, tag)] = {
            "chunk_seconds": c * 0.01, "utterances_served": int(n_done), "frames_per_s": fps,
            "real_time_streams_sustained": fps / 100.0 if fps else None,
            "frames_per_s_whole_run": float(frames_done / total),
            "step_calls": len(calls), "streams_per_step_call": float(np.mean([x_[1] for x_ in calls])),
            "step_call_ms": {"mean": float(cm.mean()), "p50": float(np.percentile(cm, 50)), "p95": float(np.percentile(cm, 95))},
            "chunk_latency_ms": {"mean": float(cl.mean()), "p50": float(np.percentile(cl, 50)), "p95": float(np.percentile(cl, 95)),
                                 "max": float(cl.max())},
            # the last chunk of an utterance, FinalizeDecoding included (the lazy schedule prunes every frame THERE, once)
            "last_chunk_and_finalize_ms": {"mean": float(ll_.mean()), "p95": float(np.percentile(ll_, 95)), "max": float(ll_.max())},
            "chunks_ahead": ahead,
            "step": "kh_online_nnet2_serve_*: persistent decode kernel, one resident workgroup per stream; a stream gets its next "
                    "chunk while at most `chunks_ahead` chunks of its frames wait to be decoded; FinalizeDecoding asynchronous" + note}
    for c in chunks:
        serve_leg(c, "", "")
    # ... and with the offline kernel's lazy pruning schedule for the streams (kh_online_decoder_set_lazy_prune: nothing is
    # pruned while a stream advances, FinalizeDecoding prunes every frame once; same final lattices and best paths)
    if not os.environ.get("KH_STRESS_SKIP_LAZY"):   # (tools/stress_serving.py --interval-only: the reference schedule's leg alone)
        dec.set_lazy_prune(True)
        for c in chunks:
            serve_leg(c, "_lazy", "; lazy pruning schedule (no PruneActiveTokens every prune_interval frames: "
                                  "FinalizeDecoding prunes every frame once)")
        dec.set_lazy_prune(False)
    for c in (() if only_persistent else chunks[:1]):
        # ... and the same loop through the Python-side DecodableNnet2Online + advance_decoding (round 3's leg)
        dn = api.DecodableNnet2Online(nnet, n, max_t, acoustic_scale=acwt, pad_input=True, max_nnet_batch_size=max(256, c))
        all_streams = list(range(n))
        lat = []
        for rep in range(2):           # the first pass warms the allocator / the kernels
            dn.reset(all_streams)
            dec.init_decoding(all_streams)
            given = np.zeros(n, np.int64)
            decoded = np.zeros(n, np.int64)
            lat = []
            torch.cuda.synchronize()
            t_begin = time.perf_counter()
            while True:
                live = [s for s in all_streams if given[s] < lens[s]]
                if not live:
                    break
                t0 = time.perf_counter()
                cnt = [int(min(c, lens[s] - given[s])) for s in live]
                fin = [given[s] + k == lens[s] for s, k in zip(live, cnt)]
                dn.accept_features_many(live, x, [int(off[s] + given[s]) for s in live], cnt, fin)
                for s, k in zip(live, cnt):
                    given[s] += k
                out = dn.compute(live, [int(decoded[s]) for s in live])
                adv = [(s, o) for s, o in zip(live, out) if o.shape[0] > 0]
                if adv:
                    dec.advance_decoding([s for s, _ in adv], [o for _, o in adv])
                    for s, o in adv:
                        decoded[s] += o.shape[0]
                api.synchronize()
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0, len(live), sum(o.shape[0] for o in out)))
            dec.finalize_decoding(all_streams)
            api.synchronize()
            total = time.perf_counter() - t_begin
        assert np.array_equal(decoded, lens), "every frame decoded"
        ms = np.array([l[0] for l in lat]) * 1e3
        full = np.array([l[1] == n for l in lat])
        ok = sum(dec.stats(s)["reached_final"] for s in range(0, n, max(1, n // 16)))
        res["chunk_%d_frames_python_loop" % c] = {
            "chunk_seconds": c * 0.01, "steps": len(lat), "frames_per_s": float(lens.sum() / total),
            "step_latency_ms": {"mean_all_streams_live": float(ms[full].mean()) if full.any() else None,
                                "p50": float(np.percentile(ms, 50)), "p95": float(np.percentile(ms, 95)), "max": float(ms.max())},
            "reached_final_of_sampled": int(ok), "step": "api.DecodableNnet2Online.compute + advance_decoding from Python"}
    return res


def release_device_memory(api, torch):
    """Between legs: drop what the previous leg left cached (python cycles, torch's caching allocator, the library's own
    block pool) so that the next leg sizes its arenas from the memory that is really free.  The driver's round-3 run lost
    the online leg to exactly that: hipMemGetInfo does not count blocks torch holds in its cache, and with the headline
    run's score matrices (45 GB each) still cached only 64 of 256 stream slots fitted."""
    import gc
    gc.collect()
    api.synchronize()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    api.pool_release()


def run_all(api, torch, main_workload=None, out=None, state=None):
    """main_workload: bench.py's (net, priors, graph, feats, utt offsets, decoder config, acwt) for the online leg.
    out / state: filled in place (results by leg; state["leg"] = the leg that is running) for bench.py's watchdog."""
    out = {} if out is None else out
    for name, fn in (("gmm_cfg2", gmm_cfg2), ("nnet_cfg3", nnet_cfg3), ("decode_cfg3", decode_cfg3), ("lattice_fb_cfg5", lattice_fb_cfg5),
                     ("ivector_f3", ivector_f3), ("online2_cfg4", lambda a, t: online2_cfg4(a, t, main_workload))):
        if state is not None:
            state["leg"] = name
        try:
            release_device_memory(api, torch)
            out[name] = fn(api, torch)
        except Exception as e:  # a secondary leg never fails the headline run
            out[name] = {"error": repr(e)}
    if state is not None:
        state["leg"] = None
    return out


if __name__ == "__main__":
    import json
    import sys
    import torch
    sys.path.insert(0, ".")
    api = importlib.import_module(PKG + ".api")
    api.select_gpu(0)
    print(json.dumps(run_all(api, torch), indent=1))
