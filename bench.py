#!/usr/bin/env python3
"""Headline benchmark: frames/sec decoded + real-time factor, LibriSpeech-shaped
nnet2 decode (BASELINE.json config 4) on N MI355X of one node.

A "step" = one pass of the hot path over one per-GPU shard of utterances, inputs
already resident in HBM:  nnet2 forward of every frame (Splice -> FixedAffine ->
4 x (Affine, Pnorm, Normalize) -> Affine -> Softmax -> SumGroup, then
DecodableAmNnet's floor/log/-log prior/x acwt)  ->  LatticeFasterDecoder over the
whole shard (beam 15, max-active 7000, min-active 200, lattice-beam 8, acwt 0.1:
steps/nnet2/decode.sh:14-20)  ->  raw lattice + best path of every utterance
(what DecodeUtteranceLatticeFaster takes from the decoder,
decoder-wrappers.cc:232-262; lattice determinization is outside SURVEY.md §8).

Workload ("librispeech_nnet_a_synthetic"): no corpus or trained model is
available offline, so everything is seeded synthetic data of the reference
recipe's shape: nnet_a 140 -> 700 -> 4x(3500/350) ->
12000 -> 5800 pdfs with random weights, a random HCLG-like graph with 10 M
states / ~25 M arcs, and per GPU a test-clean-sized set of 2620 utterances
(~1.94 M frames, LibriSpeech-like length distribution).  Multi-GPU = utterance
sharding (weak scaling: every rank decodes its own 2620-utterance set); the only
collective is the final all-reduce of {frames, utterances, tot_like}
(nnet-latgen-faster.cc:100-101,133-135) over RCCL.

RTF = elapsed * 100 / frames (nnet-latgen-faster.cc:179-182).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "old-kaldi-git_amd"

ACWT = 0.1
DECODE_CFG = dict(beam=15.0, max_active=7000, min_active=200, lattice_beam=8.0)


def build_workload(seed, rank, n_utts, graph_states, small=False):
    workloads = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(seed)           # model + graph: identical on every rank
    if small:
        net, priors = workloads.make_pnorm_net(rng, feat_dim=40, splice=2, const_dim=10, pnorm_in=400,
                                               pnorm_out=40, n_hidden=2, n_mix=600, n_pdf=300, final_scale=2.0)
    else:
        net, priors = workloads.librispeech_nnet_a(rng, final_scale=2.0)
    n_pdf = net[-1]["output_dim"]
    g = workloads.make_hclg_like(rng, graph_states, n_pdf)
    urng = np.random.default_rng(seed * 1000 + 17 + rank)   # this rank's utterances
    if small:
        lens = urng.integers(40, 120, n_utts)
    else:
        lens = workloads.utterance_lengths(urng, n_utts)
    lens = np.sort(lens)[::-1].copy()            # longest first
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = urng.standard_normal((int(off[-1]), net[0]["input_dim"])).astype(np.float32)
    return net, priors, g, feats, off


def forward_all(api, torch, nnet, feats_d, off, loglikes, max_rows):
    """nnet forward in groups of utterances (bounds the activation buffers)."""
    u0 = 0
    n = len(off) - 1
    while u0 < n:
        u1 = u0 + 1
        while u1 < n and off[u1 + 1] - off[u0] <= max_rows:
            u1 += 1
        sub = (off[u0:u1 + 1] - off[u0]).astype(np.int32)
        nnet.compute(feats_d[off[u0]:off[u1]], sub, True, epilogue=True, prob_scale=ACWT,
                     out=loglikes[off[u0]:off[u1]])
        u0 = u1


def cpu_baseline(net, priors, g, feats, off, budget_frames):
    """The reference CPU path on a bounded sample of the same workload, 1 thread:
    nnet2 forward through the reference's own compiled CPU code when oracle/_ref
    exists (cblas_sgemm as in the reference), else the restatement; decoding by
    the restatement in reference-iteration-order mode (the reference decoder cannot
    be compiled here).  Returns (frames_per_sec, description)."""
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    from oracle import binding
    fwd = binding.OracleLib("ref") if binding.have_ref() else binding.OracleLib("ko")
    # utterances closest to the median length until the frame budget is reached (the
    # shortest ones would over-weight the cheap first frames of every utterance)
    lens = np.diff(off)
    order = np.argsort(np.abs(lens - np.median(lens)), kind="stable")
    sample, tot = [], 0
    for u in order:
        T = int(off[u + 1] - off[u])
        if sample and tot + T > budget_frames:
            break
        sample.append(int(u))
        tot += T
    dec = binding.DecoderOracle(g, binding.decoder_config(**DECODE_CFG), "reference")
    t0 = time.perf_counter()
    t_fwd = 0.0
    for u in sample:
        x = feats[off[u]:off[u + 1]]
        a = time.perf_counter()
        ll = fwd.decodable_am_nnet(net, priors, ACWT, x)
        t_fwd += time.perf_counter() - a
        dec.decode(ll)
        dec.best_path()
        dec.raw_lattice()
    el = time.perf_counter() - t0
    desc = ("%d median-length utterances of the rank-0 shard (%d frames): nnet2 forward via %s, "
            "LatticeFasterDecoder restatement in reference iteration order, 1 thread; "
            "forward %.1f s + decode %.1f s" %
            (len(sample), tot, "compiled reference (oracle/_ref, OpenBLAS sgemm)" if fwd.kind == "ref" else "restatement",
             t_fwd, el - t_fwd))
    # (ii) many host cores, one utterance per PROCESS — how the recipes' $nj jobs /
    # nnet-latgen-faster-parallel use a machine.  Forked workers share the graph and
    # the model copy-on-write and run CPU code only (they never touch the GPU and leave through
    # os._exit); bounded: at most 32 workers, 120 s.
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_proc = max(1, min(cores, 32, len(order)))
    mine = [int(u) for u in order[:n_proc]]
    all_cores = None
    if hasattr(os, "fork"):
        import signal
        t1 = time.perf_counter()
        pids = []
        for k in range(n_proc):
            pid = os.fork()
            if pid == 0:
                rc = 1
                try:
                    x = feats[off[mine[k]]:off[mine[k] + 1]]
                    d2 = binding.DecoderOracle(g, binding.decoder_config(**DECODE_CFG), "reference")
                    d2.decode(fwd.decodable_am_nnet(net, priors, ACWT, x))
                    d2.best_path()
                    d2.raw_lattice()
                    rc = 0
                finally:
                    os._exit(rc)
            pids.append(pid)
        ok, deadline = True, time.time() + 120.0
        for pid in pids:
            while True:
                done, status = os.waitpid(pid, os.WNOHANG)
                if done:
                    ok = ok and os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
                    break
                if time.time() > deadline:
                    os.kill(pid, signal.SIGKILL)
                    os.waitpid(pid, 0)
                    ok = False
                    break
                time.sleep(0.01)
        el_all = time.perf_counter() - t1
        tot_all = int(sum(off[u + 1] - off[u] for u in mine))
        if ok:
            all_cores = {"value": tot_all / el_all, "unit": "frames/s", "cores": n_proc,
                         "sample": "%d median-length utterances (%d frames), one per process on %d host cores, %.1f s"
                                   % (n_proc, tot_all, cores, el_all)}
    return tot / el, desc, all_cores


def measured_traffic(args, n_utts, world):
    """HBM bytes per DecodeKernel launch from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, tools/collect_profiles.sh):
    counters cannot be read inside this process.  Only quoted for the workload they
    were measured on (the default one, 1 GPU); FETCH_SIZE + WRITE_SIZE are in KB and
    taken as reported (MI355X guide: FETCH_SIZE is exact for 64-B requests and halves
    wide coalesced reads, so this is a lower bound)."""
    if args.small or args.utts != 2620 or args.graph_states != 10_000_000 or world != 1:
        return None
    tot = 0.0
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", "r01_pmc_%s.txt" % name)
        try:
            with open(path) as f:
                fields = f.readline().strip().split(",")
            tot += float(fields[2]) * 1024.0
        except (OSError, IndexError, ValueError):
            return None
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=2620, help="utterances per GPU (test-clean has 2620)")
    ap.add_argument("--graph-states", type=int, default=10_000_000)
    ap.add_argument("--small", action="store_true", help="tiny model/graph for a quick functional run")
    ap.add_argument("--cpu-frames", type=int, default=1500, help="frame budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    api = importlib.import_module(PKG + ".api")
    api.select_gpu(local_rank)                   # CuDevice::SelectGpuId(ordinal)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.small:
        args.graph_states = min(args.graph_states, 200_000)
    net, priors, g, feats, off = build_workload(3456, rank, args.utts, args.graph_states, args.small)
    n_utts = len(off) - 1
    frames = int(off[-1])
    n_pdf = net[-1]["output_dim"]

    nnet = api.Nnet(net, priors)
    fst = api.Fst(g)
    dec = api.LatticeFasterDecoder(fst, api.decoder_config(**DECODE_CFG), max_batch=n_utts,
                                   max_frames=int(np.diff(off).max()))
    feats_d = torch.from_numpy(feats).cuda()     # inputs resident in HBM before the timed region
    stride = (n_pdf + 3) // 4 * 4
    loglikes = torch.empty((frames, stride), dtype=torch.float32, device="cuda")[:, :n_pdf]

    stats = {}

    verbose = bool(os.environ.get("BENCH_VERBOSE"))

    def step():
        t = [time.perf_counter()]
        forward_all(api, torch, nnet, feats_d, off, loglikes, max_rows=60000)
        if verbose:
            torch.cuda.synchronize(); api.synchronize(); t.append(time.perf_counter())
        dec.decode(loglikes, off)
        t.append(time.perf_counter())
        dec.prepare()                            # raw lattices + best paths, host threads
        t.append(time.perf_counter())
        tot_like, n_ok = 0.0, 0
        arcs = toks = 0
        for u in range(n_utts):
            bp = dec.get_best_path(u)            # GetBestPath + lattice export (host part)
            tot_like += -(bp["graph_cost"] + bp["acoustic_cost"])
            n_ok += 1
            st = dec.counters(u)
            arcs += st["arcs_expanded"]
            toks += st["tokens_created"]
        t.append(time.perf_counter())
        stats.update(tot_like=tot_like, n_ok=n_ok, arcs=arcs, toks=toks, kernel_ms=dec.last_kernel_ms())
        if verbose and rank == 0:
            d = np.diff(t) * 1e3
            print("[bench] forward %.0f ms, decode() %.0f ms (kernel %.0f), prepare %.0f ms, fetch %.0f ms"
                  % (d[0], d[1], stats["kernel_ms"], d[2], d[3]), file=sys.stderr)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(stats["kernel_ms"])
    sync()
    elapsed = time.perf_counter() - t0

    # max over ranks + the final scalar reduction (tot frames / like / utterances)
    red = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(frames), stats["tot_like"], float(stats["n_ok"])], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(red.item())
    total_frames = float(tot[0].item())

    if rank == 0:
        fps = total_frames * args.steps / elapsed
        # roofline of the dominant kernel (DecodeKernel): algorithmic bytes per launch =
        # 60 B per expanded arc + 16 B per created token / measured duration
        alg_bytes = stats["arcs"] * 60.0 + stats["toks"] * 16.0
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            # BASELINE.json's metric; value = frames/s of nnet2 forward + LatticeFasterDecoder, "rtf" = the
            # real-time factor per GPU (10 ms frames: rtf = 100 / frames-per-second-per-GPU)
            "metric": "frames/sec decoded + real-time factor, LibriSpeech nnet2 decode @1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rtf": elapsed * 100.0 / (total_frames * args.steps) * world,
            "config": {"workload": "librispeech_nnet_a_synthetic" + ("_small" if args.small else ""),
                       "utts_per_gpu": n_utts, "frames_per_gpu": frames, "graph_states": int(g["num_states"]),
                       "graph_arcs": int(g["arc_offsets"][-1]), "nnet": "140-700-4x(3500/350)-12000-5800" if not args.small else "small",
                       "decoder": DECODE_CFG, "acwt": ACWT, "parallelism": "utterance-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": measured_traffic(args, n_utts, world),
                         "kernel": "DecodeKernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "arcs_expanded_per_launch": stats["arcs"], "tokens_created_per_launch": stats["toks"]},
            "loglike_per_frame": stats["tot_like"] / frames,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU legs run at N = 1 only
            v, desc, all_cores = cpu_baseline(net, priors, g, feats, off, args.cpu_frames)
            out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": 1, "kind": "port", "sample": desc,
                                   "all_cores": all_cores}
            out["speedup_vs_cpu_1thread_per_gpu"] = fps / world / v
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
