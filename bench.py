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
decoder-wrappers.cc:232-262).

Workload ("librispeech_nnet_a_structured"): no corpus or trained model is
available offline, so everything is seeded synthetic data of the reference
recipe's shape: nnet_a 140 -> 700 -> 4x(3500/350) ->
12000 -> 5800 pdfs with random weights (biases centred so that the outputs depend
on the input, workloads.calibrate_biases), an HCLG-STRUCTURED graph of ~10 M
states (per LM state a prefix tree of pronunciations, 3-state HMM chains with
self-loops, pushed LM costs, back-off epsilon arcs: workloads.make_hclg_structured),
and per GPU a test-clean-sized set of 2620 utterances (~1.94 M frames).  Every
utterance follows a random path through the graph and its features make the
network score that path's pdfs (workloads.make_path_features), so the search
behaves like a real decode: a true hypothesis, competing ones that reconverge,
lattices of tens of arcs per frame.

Launch.  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts
the N ranks itself (a child `python -m torch.distributed.run`, created BEFORE this
process touches a GPU; the parent only relays output and the exit code).  Under
torchrun (WORLD_SIZE set) the process is one rank.  Multi-GPU = utterance
sharding with no data-path collective; the only collective is the final
all-reduce of {frames, utterances, tot_like} (nnet-latgen-faster.cc:100-101,
133-135) over RCCL.  The headline value is WEAK scaling (every rank decodes its own
2620-utterance set); for N > 1 the line also carries a "strong_scaling" object:
ONE 2620-utterance set sharded by sharding.partition_utterances, as the recipes
split one data set over $nj jobs (steps/nnet2/decode.sh:130-136).

RTF = elapsed * 100 / frames (nnet-latgen-faster.cc:179-182).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "old-kaldi-git_amd"

ACWT = 0.1
DECODE_CFG = dict(beam=15.0, max_active=7000, min_active=200, lattice_beam=8.0)
FEATURE_NOISE = 0.20          # N(0, .) on the prototype part of the features: sets the frame accuracy (~0.8) and the lattice density


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=2620, help="utterances per GPU (test-clean has 2620)")
    ap.add_argument("--graph-states", type=int, default=10_000_000)
    ap.add_argument("--small", action="store_true", help="tiny model/graph for a quick functional run")
    ap.add_argument("--cpu-utts", type=int, default=20, help="utterances of the single-thread CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg at N > 1")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (config 2 / 5) legs")
    ap.add_argument("--secondary-timeout", type=int, default=300,
                    help="seconds the secondary legs may take before the line is printed without the one that hangs")
    ap.add_argument("--secondary-only", action="store_true",
                    help="(internal) run the secondary legs alone and print {\"secondary\": ...}: bench.py starts itself with this "
                         "in a CHILD process, so that a leg that takes the GPU down costs its own figures, not the headline line")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the unpipelined and the canonical-rule repeats of the step")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the value_end_to_end leg (the same steps with determinization timed)")
    ap.add_argument("--no-gpu-dryrun", action="store_true",
                    help="launcher / sharding / reduction only (gloo, no GPU, nothing decoded): CPU test of the multi-rank path")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


def launch_ranks(args):
    """`--gpus N` without a torchrun environment: start the N ranks as a child process
    tree and relay its result.  This process never initialises a GPU."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def build_model_and_graph(seed, graph_states, small):
    workloads = importlib.import_module(PKG + ".workloads")
    rng = np.random.default_rng(seed)           # model + graph: identical on every rank
    if small:
        net, _ = workloads.make_pnorm_net(rng, feat_dim=40, splice=2, const_dim=20, pnorm_in=400,
                                          pnorm_out=40, n_hidden=2, n_mix=600, n_pdf=300, final_scale=14.0)
    else:
        net, _ = workloads.librispeech_nnet_a(rng, final_scale=14.0)
    priors = workloads.calibrate_biases(np.random.default_rng(seed + 1), net)
    n_pdf = net[-1]["output_dim"]
    g = workloads.make_hclg_structured(np.random.default_rng(seed + 2), graph_states, n_pdf)
    protos, _ = workloads.make_pdf_prototypes(np.random.default_rng(seed + 3), net, priors,
                                              n_candidates=4096 if small else 16384)
    return net, priors, g, protos


def build_utterances(seed, set_id, n_utts, net, g, protos, small):
    """One test-clean-sized set of utterances: lengths, true paths, features."""
    workloads = importlib.import_module(PKG + ".workloads")
    urng = np.random.default_rng(seed * 1000 + 17 + set_id)
    if small:
        lens = urng.integers(40, 120, n_utts)
    else:
        lens = workloads.utterance_lengths(urng, n_utts)
    lens = np.sort(lens)[::-1].copy()            # longest first
    seqs = workloads.sample_paths(urng, g, lens)
    feats = workloads.make_path_features(urng, net, protos, seqs, noise=FEATURE_NOISE)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return feats, off


def take_utterances(feats, off, idx):
    """Sub-shard (features, offsets) of the utterances `idx` (kept longest first)."""
    idx = sorted((int(i) for i in idx), key=lambda u: -(int(off[u + 1]) - int(off[u])))
    parts = [feats[off[u]:off[u + 1]] for u in idx]
    lens = np.array([len(p) for p in parts], np.int64)
    f = np.concatenate(parts) if parts else np.zeros((0, feats.shape[1]), np.float32)
    return f, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def forward_all(nnet, feats_d, off, loglikes, max_rows):
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


# ---------------------------------------------------------------- CPU baseline
def _cpu_decode_one(binding, fwd, net, priors, g, x):
    dec = binding.DecoderOracle(g, binding.decoder_config(**DECODE_CFG), "reference")
    a = time.perf_counter()
    ll = fwd.decodable_am_nnet(net, priors, ACWT, x)
    t_fwd = time.perf_counter() - a
    dec.decode(ll)
    dec.best_path()
    dec.raw_lattice()
    return t_fwd


def _cpu_secondary(binding, fwd):
    """1-thread CPU samples of BASELINE.json configs 2, 3, 5 (the GPU side: tools/bench_secondary.py),
    a few seconds each: the numbers of BASELINE.md section 2 point 5."""
    workloads = importlib.import_module(PKG + ".workloads")
    out = {}
    ko = binding.OracleLib("ko")
    # config 2: frame x pdf matrix of rm-tri1-sized GMMs (DiagGmm::LogLikelihoods + LogSumExp per pdf)
    rng = np.random.default_rng(1)
    am = workloads.make_am_gmm(rng, 1800, 9000, 39)
    mi, iv = workloads.gmm_inv_params(am)
    gc, _ = ko.gmm_compute_gconsts(am["weights"], mi, iv)
    x = rng.standard_normal((1200, 39)).astype(np.float32)
    t0 = time.perf_counter()
    ko.am_gmm_loglikes(x, gc, mi, iv, am["pdf_offsets"])
    dt = time.perf_counter() - t0
    out["gmm_cfg2"] = {"frames_per_s": len(x) / dt, "kind": "port", "cores": 1,
                       "sample": "%d frames x 39 dims, 1800 pdfs / 9000 Gaussians, %.1f s" % (len(x), dt)}
    # config 3: wsj nnet5d p-norm forward (the reference's compiled code when present)
    rng = np.random.default_rng(2)
    net3, priors3 = workloads.wsj_nnet5d(rng)
    x = rng.standard_normal((1500, 40)).astype(np.float32)
    t0 = time.perf_counter()
    fwd.decodable_am_nnet(net3, priors3, 0.1, x)
    dt = time.perf_counter() - t0
    out["nnet_cfg3"] = {"frames_per_s": len(x) / dt, "kind": "reference" if fwd.kind == "ref" else "port", "cores": 1,
                        "sample": "forward of %d frames, %.1f s" % (len(x), dt)}
    # config 3 end to end: forward + reference-order LatticeFasterDecoder on the wsj-sized structured workload
    bs = importlib.import_module("tools.bench_secondary")
    net3d, pri3d, g3, feats3, off3, scores3 = bs.cfg3_workload(333)
    lens3 = np.diff(off3)
    pick = np.argsort(np.abs(lens3 - np.median(lens3)), kind="stable")[:2]
    t0 = time.perf_counter()
    for u in pick:
        fwd.decodable_am_nnet(net3d, pri3d, ACWT, feats3[off3[u]:off3[u + 1]])
        dec3 = binding.DecoderOracle(g3, binding.decoder_config(**DECODE_CFG), mode="reference")
        dec3.decode(scores3(int(u)))
        dec3.best_path()
        dec3.raw_lattice()
    dt = time.perf_counter() - t0
    n3 = int(lens3[pick].sum())
    out["decode_cfg3"] = {"frames_per_s": n3 / dt, "kind": "port", "cores": 1,
                          "sample": "2 median-length utterances (%d frames), forward + reference-order decode, %.1f s" % (n3, dt)}
    # config 5: lattice forward-backward (denominator lattices: structured graph, lattice-beam 8)
    rng = np.random.default_rng(5)
    P, T, N = 600, 200, 6
    g5 = workloads.make_hclg_structured(rng, 60_000, P)
    seqs = workloads.sample_paths(rng, g5, [T] * N)
    dec = binding.DecoderOracle(g5, binding.decoder_config(beam=13.0, max_active=7000, min_active=200, lattice_beam=8.0),
                                mode="reference")
    csrs, alis = [], []
    for q in seqs:
        ll = (rng.standard_normal((T, P)) * 0.28 - 0.37).astype(np.float32)
        ll[np.arange(T), q] = (0.5 + 0.3 * rng.standard_normal(T)).astype(np.float32)
        dec.decode(ll)
        csrs.append(binding.lattice_csr(dec.raw_lattice()))
        alis.append(np.asarray(dec.best_path()["alignment"], np.int32))
    arcs = sum(len(c["arc_ilabel"]) for c in csrs)
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        for c in csrs:
            binding.lattice_forward_backward(c)
    dt = (time.perf_counter() - t0) / reps
    ntid = len(g5["tid2pdf"]) - 1
    t2ph = np.concatenate([[0], 1 + (np.arange(ntid) // 6) % 40]).astype(np.int32)
    t1 = time.perf_counter()
    for _ in range(reps):
        for c, a in zip(csrs, alis):
            binding.lattice_forward_backward_mpe(c, t2ph, g5["tid2pdf"], [1, 2], a, "smbr")
    dt_mpe = (time.perf_counter() - t1) / reps
    out["lattice_fb_cfg5"] = {"arcs_per_s": arcs / dt, "frames_per_s": N * T / dt, "smbr_arcs_per_s": arcs / dt_mpe,
                              "kind": "port", "cores": 1,
                              "sample": "%d lattices x %d frames, %d arcs, %d repetitions: LatticeForwardBackward %.2f ms, "
                                        "sMBR variant %.2f ms per pass" % (N, T, arcs, reps, dt * 1e3, dt_mpe * 1e3)}
    return out


def cpu_baseline_child(rfd, wfd, net, priors, g, feats, off, n_utts_1t):
    """Runs in a process forked BEFORE the parent initialised the GPU (it never touches
    one): waits for the parent's go (the GPU timing is over), then times the reference
    CPU path on a bounded sample of the same workload and writes one JSON line back.
      (i)  1 thread, `n_utts_1t` median-length utterances: nnet2 forward through the
           reference's own compiled code when oracle/_ref exists (cblas_sgemm as in the
           reference), else the restatement; LatticeFasterDecoder = the restatement in
           reference iteration order (the reference decoder needs OpenFst: not buildable);
      (ii) every host core, one utterance per PROCESS — how the recipes' $nj jobs /
           nnet-latgen-faster-parallel use a machine."""
    import signal
    out = {}
    # (the parent closes the pipe without a "go" when it dies before the GPU timing is over: leave at once
    # instead of loading the host cores of a shared box for minutes)
    if os.read(rfd, 1) != b"g":
        os._exit(1)
    try:
        os.environ["OPENBLAS_NUM_THREADS"] = "1"
        os.environ["OMP_NUM_THREADS"] = "1"
        from oracle import binding
        fwd = binding.OracleLib("ref") if binding.have_ref() else binding.OracleLib("ko")
        lens = np.diff(off)
        order = np.argsort(np.abs(lens - np.median(lens)), kind="stable")
        sample = [int(u) for u in order[:max(1, n_utts_1t)]]
        tot = int(sum(lens[u] for u in sample))
        t0 = time.perf_counter()
        t_fwd = 0.0
        for u in sample:
            t_fwd += _cpu_decode_one(binding, fwd, net, priors, g, feats[off[u]:off[u + 1]])
        el = time.perf_counter() - t0
        out["value"] = tot / el
        out["kind"] = "port"
        out["cores"] = 1
        out["sample"] = ("%d median-length utterances of the rank-0 shard (%d frames): nnet2 forward via %s, "
                         "LatticeFasterDecoder restatement in reference iteration order, 1 thread; "
                         "forward %.1f s + decode %.1f s" %
                         (len(sample), tot,
                          "compiled reference (oracle/_ref, OpenBLAS sgemm)" if fwd.kind == "ref" else "restatement",
                          t_fwd, el - t_fwd))
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = cores
        try:   # the container's CPU time: cgroup v2 cpu.max = "<quota us> <period us>" (the pool's GPU boxes: 16 CPUs of 256 visible)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
        except (OSError, ValueError):
            pass
        # one process per CPU the container may use ($nj jobs of the recipe), a few utterances each (the graph's pages are
        # already resident: the 1-thread leg above ran in this process); more processes than the quota only measure the throttle
        per_proc = 3
        n_proc = max(1, min(cores, quota, len(order) // per_proc))
        mine = [int(u) for u in order[:n_proc * per_proc]]
        t1 = time.perf_counter()
        pids = []
        for k in range(n_proc):
            pid = os.fork()
            if pid == 0:
                rc = 1
                try:
                    for u in mine[k::n_proc]:
                        _cpu_decode_one(binding, fwd, net, priors, g, feats[off[u]:off[u + 1]])
                    rc = 0
                finally:
                    os._exit(rc)
            pids.append(pid)
        ok, deadline = True, time.time() + 180.0
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
        tot_all = int(sum(lens[u] for u in mine))
        out["all_cores"] = ({"value": tot_all / el_all, "unit": "frames/s", "cores": n_proc,
                             "sample": "%d median-length utterances (%d frames), %d per process on %d processes (%d cores visible, "
                                       "CPU quota of the container %d), %.1f s" % (len(mine), tot_all, per_proc, n_proc, cores, quota, el_all)}
                            if ok else None)
        if not os.environ.get("BENCH_NO_CPU_SECONDARY"):
            try:
                out["secondary"] = _cpu_secondary(binding, fwd)
            except Exception as e:  # never fails the headline baseline
                out["secondary"] = {"error": repr(e)}
    except BaseException as e:  # the parent reports the failure
        out = {"error": repr(e)}
    finally:
        try:
            os.write(wfd, (json.dumps(out) + "\n").encode())
        except OSError:   # the parent is gone (EPIPE): still leave through _exit, never unwind into main()
            pass
        os._exit(0)


def start_cpu_baseline(net, priors, g, feats, off, n_utts_1t):
    go_r, go_w = os.pipe()
    res_r, res_w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(go_w)
        os.close(res_r)
        cpu_baseline_child(go_r, res_w, net, priors, g, feats, off, n_utts_1t)
    os.close(go_r)
    os.close(res_w)
    return pid, go_w, res_r


def finish_cpu_baseline(handle):
    pid, go_w, res_r = handle
    os.write(go_w, b"g")
    buf = b""
    while True:
        chunk = os.read(res_r, 65536)
        if not chunk:
            break
        buf += chunk
    os.waitpid(pid, 0)
    return json.loads(buf.decode().strip().splitlines()[-1]) if buf.strip() else {"error": "no result"}


# ---------------------------------------------------------------- profiles
def measured_traffic(args, world, kernel_ms, ref_order=True):
    """HBM bytes per DecodeKernel launch from the PMC record committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, tools/collect_profiles.sh ->
    tools/pmc_record.py): counters cannot be read inside this process, so this is a RECORDED figure,
    quoted only for the workload it was measured on (the default one, 1 GPU), only while the record's
    hash of csrc/kh_decoder.hip equals the running source's and its kernel duration is within 5 % of this
    run's - otherwise `traffic` is null and `traffic_source` says why (a stale profile must not
    be quoted as this build's).  Both counters are in KiB.  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B
    although a read request moves a 128-byte line (MI355X guide, HBM section: "double it"): calibrated for
    THIS kernel's access shapes by tools/pmc_calibrate.hip (profiles/r02_pmc_calibration.txt) and confirmed
    on DecodeKernel itself by the request-size and DRAM-side tallies.  WRITE_SIZE is exact.
    traffic = 2 x FETCH_SIZE + WRITE_SIZE.  The record names the search it was taken for ("search":
    "reference-order" = DecodeKernel<.., true>, the default since round 6); a record of the other kernel is not quoted."""
    if args.small or args.utts != 2620 or args.graph_states != 10_000_000 or world != 1:
        return None, "not the recorded workload"
    import glob
    import hashlib
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not recs:
        return None, "no profiles/r*_pmc_traffic.json"
    path = recs[-1]
    try:
        with open(path) as f:
            rec = json.load(f)
        with open(os.path.join(ROOT, PKG, "csrc", "kh_decoder.hip"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        name = os.path.relpath(path, ROOT)
        if rec.get("kernel_src_sha16") != sha:
            return None, "%s was recorded for another build of the kernel (source hash %s, running %s): not quoted" % (
                name, rec.get("kernel_src_sha16"), sha)
        want = "reference-order" if ref_order else "canonical"
        if rec.get("search", "canonical") != want:     # (records of rounds 2-5 carry no "search": they are the canonical kernel's)
            return None, "%s was recorded for the %s search, this run timed the %s one: not quoted" % (name, rec.get("search", "canonical"), want)
        ref_ms = float(rec["kernel_trace_avg_ms"])
        if not (abs(kernel_ms - ref_ms) <= 0.05 * ref_ms):
            return None, "%s: recorded kernel duration %.1f ms differs from this run's %.1f ms by more than 5 %%: not quoted" % (
                name, ref_ms, kernel_ms)
        tot = (2.0 * float(rec["FETCH_SIZE_KiB"]) + float(rec["WRITE_SIZE_KiB"])) * 1024.0
        return tot, ("2 x FETCH_SIZE + WRITE_SIZE from %s (recorded by tools/collect_profiles.sh for this kernel source, not "
                     "measured in this run; read correction: profiles/r02_pmc_calibration.txt)" % name)
    except (OSError, KeyError, ValueError) as e:
        return None, "unreadable PMC record: %r" % (e,)


def dryrun(args, rank, world):
    """CPU test of the multi-rank path: gloo process group, the strong-scaling partition,
    the scalar reduction and the JSON line - nothing is decoded."""
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module(PKG + ".sharding")
    workloads = importlib.import_module(PKG + ".workloads")
    if os.environ.get("BENCH_FAIL_RANK") == str(rank):      # test hook: a rank that dies
        raise RuntimeError("rank %d: induced failure" % rank)
    if world > 1:
        dist.init_process_group("gloo")
    lens = np.sort(workloads.utterance_lengths(np.random.default_rng(3456017), args.utts))[::-1]
    parts = sharding.partition_utterances(lens, world)
    mine = lens[parts[rank]]
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    red = sharding.reduce_decode_totals(int(mine.sum()), -1.0 * float(mine.sum()), len(mine), 0,
                                        time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "dryrun", "n_gpus": world, "frames": red["frames"], "utterances": red["num_success"],
                          "elapsed": red["elapsed"], "per_rank_frames_rank0": int(mine.sum()),
                          "expected_frames": int(lens.sum())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def secondary_only(args):
    """The secondary legs (BASELINE.json configs 2, 3, 5, iVectors, config 4 as a serving loop: tools/bench_secondary.py) in a
    process of their own; bench.py's main run starts it and takes {"secondary": {...}} from the last line of its stdout.
    Under a watchdog: a leg that does not come back within --secondary-timeout costs its own figures - what is there is
    printed, the leg is named, the process leaves with status 3."""
    import threading
    net, priors, g, protos = build_model_and_graph(3456, args.graph_states, False)
    feats, off = build_utterances(3456, 0, args.utts, net, g, protos, False)
    import torch
    api = importlib.import_module(PKG + ".api")
    api.select_gpu(0)
    secondary = {}
    state = {"leg": None, "printed": False}
    lock = threading.Lock()

    def emit_and_leave():
        with lock:
            if state["printed"]:
                return
            state["printed"] = True
            try:   # where every thread stands, for whoever reads the run's stderr
                import faulthandler
                print("[bench] watchdog: the secondary leg %r has not returned within %d s; Python stacks of all threads:"
                      % (state["leg"], args.secondary_timeout), file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            except Exception:   # noqa: BLE001
                pass
            sec_out = dict(secondary)
            if state["leg"] is not None and state["leg"] not in sec_out:
                sec_out[state["leg"]] = {"error": "no result within %d s (watchdog): the leg was abandoned" % args.secondary_timeout}
            print(json.dumps({"secondary": sec_out}))
            sys.stdout.flush()
        os._exit(3)

    timer = threading.Timer(float(args.secondary_timeout), emit_and_leave)
    timer.daemon = True
    timer.start()
    try:
        sec = importlib.import_module("tools.bench_secondary")
        sec.run_all(api, torch, (net, priors, g, feats, off, DECODE_CFG, ACWT), out=secondary, state=state)
    except Exception as e:  # noqa: BLE001
        secondary["error"] = repr(e)
    timer.cancel()
    with lock:
        if state["printed"]:      # (the watchdog is printing / has printed: it also leaves)
            time.sleep(3600)
        state["printed"] = True
    print(json.dumps({"secondary": secondary}))
    sys.stdout.flush()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.no_gpu_dryrun:
        return dryrun(args, rank, world)
    if args.secondary_only:
        return secondary_only(args)

    # ---- workload (numpy only: no GPU has been touched yet)
    if args.small:
        args.graph_states = min(args.graph_states, 200_000)
    t_build = time.perf_counter()
    net, priors, g, protos = build_model_and_graph(3456, args.graph_states, args.small)
    feats, off = build_utterances(3456, rank, args.utts, net, g, protos, args.small)
    strong = None
    if world > 1 and not args.no_strong:
        sharding = importlib.import_module(PKG + ".sharding")
        f0, o0 = (feats, off) if rank == 0 else build_utterances(3456, 0, args.utts, net, g, protos, args.small)
        parts = sharding.partition_utterances(np.diff(o0), world)
        strong = take_utterances(f0, o0, parts[rank])
        del f0, o0
    t_build = time.perf_counter() - t_build
    cpu_handle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_handle = start_cpu_baseline(net, priors, g, feats, off, args.cpu_utts)   # forked before GPU init

    import torch
    import torch.distributed as dist
    api = importlib.import_module(PKG + ".api")
    api.select_gpu(local_rank)                   # CuDevice::SelectGpuId(ordinal)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_pdf = net[-1]["output_dim"]
    tid_phone = np.zeros(len(g["tid2pdf"]), np.int32)
    tid_phone[1::2] = 1 + g["tid2pdf"][1::2]
    nnet = api.Nnet(net, priors)
    fst = api.Fst(g)
    verbose = bool(os.environ.get("BENCH_VERBOSE"))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(feats_h, off_h, steps, warmup, end_to_end=True):
        """Times `steps` steps over the shard (feats_h, off_h); returns the per-rank record."""
        n_utts = len(off_h) - 1
        frames = int(off_h[-1])
        dec = api.LatticeFasterDecoder(fst, api.decoder_config(**DECODE_CFG), max_batch=max(n_utts, 1),
                                       max_frames=int(np.diff(off_h).max()) if n_utts else 1)
        feats_d = torch.from_numpy(feats_h).cuda()     # inputs resident in HBM before the timed region
        stride = (n_pdf + 3) // 4 * 4
        loglikes = torch.empty((max(frames, 1), stride), dtype=torch.float32, device="cuda")[:, :n_pdf]
        stats = {}

        fe = {}

        def front_end():
            """Config 4's feature pipeline on audio of the shard's duration (online2-wav-nnet2-latgen-faster reads waveforms:
            OnlineMfcc 40 x 40 hires -> OnlineIvectorFeature, online estimation per 10-frame period, 512-Gaussian UBM, 100
            dims -> OnlineAppendFeature).  The synthetic decode workload's features are network-input prototypes, not derived
            from audio, so the front end runs on noise of the same length and shape and its output is not what is decoded:
            the same kernels doing the same amount of work inside the timed region."""
            if not fe:
                workloads = importlib.import_module(PKG + ".workloads")
                fe["mfcc"] = api.Mfcc(num_bins=40, num_ceps=40, low_freq=40.0, high_freq=-200.0)
                fe["wave"] = (1000.0 * torch.randn(frames * 160 + 240, device="cuda")).contiguous()
                fe["ext"] = api.OnlineIvectorExtractor(workloads.make_ivector_extractor(np.random.default_rng(4)))
                fe["iv"] = torch.empty((max(frames, 1), 100), dtype=torch.float32, device="cuda")
            m = fe["mfcc"].compute(fe["wave"])
            fe["ext"].extract(m[:frames], off_h, out=fe["iv"])

        def acoustic(with_front_end):
            """The part of a step in front of the decoder: (front end,) nnet2 forward pass of the whole shard."""
            if with_front_end:
                front_end()
            forward_all(nnet, feats_d, off_h, loglikes, max_rows=int(os.environ.get("KH_BENCH_MAX_ROWS", "60000")))

        lib_stream = []

        def acoustic_under_decode(with_front_end):
            """The NEXT step's acoustic part, called by decode() right after its kernel launch (kh_decoder_set_after_launch):
            enqueued on the library's stream it runs after the decode kernel, while the host threads still determinize
            this step's last lattices.  The wait for the kernel sleeps (the completion threads need the CPUs)."""
            if not lib_stream:
                import ctypes
                get = api.lib().kh_get_stream
                get.restype = ctypes.c_void_p
                lib_stream.append(torch.cuda.ExternalStream(get()))
            while not lib_stream[0].query():
                time.sleep(0.002)
            acoustic(with_front_end)

        background = []   # [(thread, exception box)] of the next step's acoustic part

        def acoustic_in_background(with_front_end):
            """decode()'s after-launch hook (called when the LAST decode kernel of the call has finished and the stream is
            idle): starts a host thread that runs the next step's acoustic part and returns, so that decode() goes on to
            join its completion threads and the step's host work (lattices, best paths, sizes) runs while the GPU is
            already computing the next scores."""
            import threading
            box = {}

            def body():
                try:
                    if os.environ.get("KH_BENCH_FAIL_BACKGROUND"):   # (test hook: the fall-back below)
                        raise RuntimeError("KH_BENCH_FAIL_BACKGROUND")
                    torch.cuda.set_device(local_rank)   # (the current device is per host thread)
                    acoustic_under_decode(with_front_end)
                except BaseException as e:   # noqa: BLE001 - re-raised by join_background()
                    box["exc"] = e
            th = threading.Thread(target=body)
            th.start()
            background.append((th, box))

        pipe = {"ready": False, "disabled": False}   # scores of the coming decode computed? background thread given up?

        def join_background():
            while background:
                th, box = background.pop()
                th.join()
                if "exc" in box:   # never lose the run to the overlap: fall back to one step after the other
                    print("[bench] rank %d: the background forward pass failed (%r); steps run unpipelined from here"
                          % (rank, box["exc"]), file=sys.stderr)
                    pipe["disabled"] = True
                else:
                    pipe["ready"] = True

        def step(determinize=False, with_front_end=False, do_acoustic=True, next_acoustic=False):
            t = [time.perf_counter()]
            # determinize: the timed region is DecodeUtteranceLatticeFaster in full (decoder-wrappers.cc:232-284) - decode,
            # raw lattice, best path, DeterminizeLatticePhonePrunedWrapper -> CompactLattice; the determinization of an
            # utterance starts on a host thread as soon as the kernel has exported it
            # det_opts at the reference's defaults (phone_determinize, word_determinize, no minimize:
            # determinize-lattice-pruned.h:163-167); the synthetic graph's transition model is one one-state phone per pdf
            # (transition-id 2 * pdf + 1 enters the state, 2 * pdf + 2 is its self-loop: workloads.make_hclg_structured)
            dec.set_determinize(determinize, DECODE_CFG["lattice_beam"], tid_phone=tid_phone)
            if do_acoustic:
                acoustic(with_front_end)
            else:
                join_background()   # this step's scores: requested under the previous step's decode
                if not pipe["ready"]:
                    acoustic(with_front_end)
            pipe["ready"] = False   # (consumed by the decode below)
            dec.set_after_launch((lambda: acoustic_in_background(with_front_end))
                                 if next_acoustic and not pipe["disabled"] else None)
            if verbose:
                torch.cuda.synchronize(); api.synchronize(); t.append(time.perf_counter())
            dec.decode(loglikes, off_h)
            t.append(time.perf_counter())
            dec.prepare()                            # raw lattices + best paths, host threads
            t.append(time.perf_counter())
            # GetBestPath (words + alignments of every utterance) and the lattice sizes, one library call each
            # for the whole shard; the sum runs over the utterances in order, in double, as the per-utterance
            # loop it replaces did
            bp = dec.get_best_paths()
            tot_like, n_ok = 0.0, n_utts
            for gc, ac in zip(bp["graph_cost"].tolist(), bp["acoustic_cost"].tolist()):
                tot_like += -(gc + ac)
            cnt, ls = dec.stats_batch()
            arcs, toks = int(cnt["arcs_expanded"].sum()), int(cnt["tokens_created"].sum())
            # (an A/B run against an older build of the library, KH_LIB_OVERRIDE, may predate the counter)
            cand = dec.search_counters(-1)["candidates_materialised"] if hasattr(api.lib(), "kh_decoder_get_search_counters") else 0
            lat_arcs, lat_states = int(ls["num_links"].sum()), int(ls["num_tokens"].sum())
            t.append(time.perf_counter())
            if determinize:
                stats["clat"] = dec.compact_lattice_totals()   # (every lattice is there: the host threads finished inside decode())
                stats["host_tail_ms"] = dec.last_host_tail_ms()
                t.append(time.perf_counter())
            stats.update(tot_like=tot_like, n_ok=n_ok, arcs=arcs, toks=toks, cand=cand, lat_arcs=lat_arcs, lat_states=lat_states,
                         kernel_ms=dec.last_kernel_ms(), ref_order=bool(dec.search_counters(-1)["reference_order"]))
            if verbose and rank == 0:
                d = np.diff(t) * 1e3
                print("[bench] forward %.0f ms, decode() %.0f ms (kernel %.0f), prepare %.0f ms, fetch %.0f ms"
                      % (d[0], d[1], stats["kernel_ms"], d[2], d[3]), file=sys.stderr)

        for _ in range(warmup):
            step()
        sync()
        kernel_ms = []
        # K steps as a binary's main loop runs them: step i + 1's forward pass is started when step i's decode kernel has finished
        # (kh_decoder_set_after_launch) from a second host thread, so the GPU goes on while the host finishes step i
        # (lattice sizes, best paths).  K forward passes and K decodes inside the timed region; KH_BENCH_NO_PIPELINE=1:
        # strictly one after the other.
        pipelined = not os.environ.get("KH_BENCH_NO_PIPELINE")
        t0 = time.perf_counter()
        if pipelined:
            acoustic(False)
            pipe["ready"] = True
        for i in range(steps):
            if pipelined:
                step(do_acoustic=False, next_acoustic=i + 1 < steps)
            else:
                step()
            kernel_ms.append(stats["kernel_ms"])
        join_background()
        sync()
        elapsed = time.perf_counter() - t0
        pipelined = pipelined and not pipe["disabled"]   # (a failed background forward pass: the rest ran one after the other)
        main_stats = dict(stats)
        # ---- the same step strictly one after the other (round-to-round comparison with the unpipelined headline of
        # rounds 1-2), and with the opt-in order-independent acceptance rule (kh_decoder_set_reference_order(dec, 0), the
        # headline of rounds 1-5): min(K, 3) steps each
        extra = {}
        if end_to_end and not args.no_extra_legs:
            k2 = max(1, min(steps, 3))
            t1 = time.perf_counter()
            for _ in range(k2):
                step()
            sync()
            extra["unpipelined"] = dict(elapsed=time.perf_counter() - t1, steps=k2, kernel_ms=stats["kernel_ms"])
            try:
                dec.set_reference_order(False)   # the opt-in order-independent rule E ("canonical"): a comparison leg
                step()        # (the slot arenas are carved again without the order's temporaries)
                sync()
                t1 = time.perf_counter()
                kms_x = []
                for _ in range(k2):
                    step()
                    kms_x.append(stats["kernel_ms"])
                sync()
                extra["canonical"] = dict(elapsed=time.perf_counter() - t1, steps=k2, kernel_ms=float(np.mean(kms_x)),
                                          tot_like=stats["tot_like"], arcs=stats["arcs"], toks=stats["toks"], cand=stats["cand"],
                                          lat_arcs=stats["lat_arcs"], lat_states=stats["lat_states"])
            except Exception as e:   # noqa: BLE001 - an extra leg never costs the headline
                extra["canonical"] = {"error": repr(e)}
            finally:
                dec.set_reference_order(True)
            step()            # back to the default arenas before the end-to-end loops
            sync()
        stats.clear()
        stats.update(main_stats)
        # ---- the same K steps with determinization in the timed region (value_end_to_end)
        e2e = None
        if end_to_end and not args.no_end_to_end:
            step(True)
            sync()
            tail, kms_e = [], []
            # K steps, pipelined as a binary's main loop would: step i + 1's forward pass starts when step i's decode kernel
            # has finished and runs while the host threads finish step i's determinization (K forward passes, K decodes, K sets
            # of CompactLattices, all inside the timed region; `pipelined: false` = KH_BENCH_NO_PIPELINE=1, one after the other)
            t1 = time.perf_counter()
            if pipelined:
                acoustic(False)
                pipe["ready"] = True
            for i in range(steps):
                if pipelined:
                    step(True, False, do_acoustic=False, next_acoustic=i + 1 < steps)
                else:
                    step(True)
                kms_e.append(stats["kernel_ms"])
                tail.append(stats["host_tail_ms"])
            sync()
            e2e = dict(elapsed=time.perf_counter() - t1, kernel_ms=float(np.mean(kms_e)), tail_ms=float(np.mean(tail)), clat=stats["clat"],
                       pipelined=pipelined and not pipe["disabled"])
            # ---- and from the waveform: config 4's binary reads audio (front end + the region above)
            if n_utts > 0 and frames >= 1000:
                step(True, True)
                sync()
                t2 = time.perf_counter()
                if pipelined:
                    acoustic(True)
                    pipe["ready"] = True
                for i in range(steps):
                    if pipelined:
                        step(True, True, do_acoustic=False, next_acoustic=i + 1 < steps)
                    else:
                        step(True, True)
                sync()
                e2e["wave_elapsed"] = time.perf_counter() - t2
                t3 = time.perf_counter()
                front_end()
                api.synchronize()
                torch.cuda.synchronize()
                e2e["front_end_ms"] = (time.perf_counter() - t3) * 1e3
                fe.clear()
            dec.set_determinize(False)
            dec.set_after_launch(None)
        red = torch.tensor([elapsed, e2e["elapsed"] if e2e else 0.0, e2e.get("wave_elapsed", 0.0) if e2e else 0.0,
                            0.0 if pipelined else 1.0, 0.0 if (e2e and e2e["pipelined"]) else 1.0,
                            extra.get("unpipelined", {}).get("elapsed", 0.0), extra.get("canonical", {}).get("elapsed", 0.0)],
                           dtype=torch.float64, device="cuda")
        tot = torch.tensor([float(frames), stats["tot_like"], float(stats["n_ok"])], dtype=torch.float64, device="cuda")
        kms = torch.zeros(world, dtype=torch.float64, device="cuda")
        kms[rank] = float(np.mean(kernel_ms))
        if world > 1:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(kms, op=dist.ReduceOp.SUM)
        del dec, feats_d, loglikes
        if e2e:
            e2e["elapsed"] = float(red[1].item())
            e2e["pipelined"] = float(red[4].item()) == 0.0   # on every rank
            if "wave_elapsed" in e2e:
                e2e["wave_elapsed"] = float(red[2].item())
        if "unpipelined" in extra:
            extra["unpipelined"]["elapsed"] = float(red[5].item())
        if "canonical" in extra and "elapsed" in extra["canonical"]:
            extra["canonical"]["elapsed"] = float(red[6].item())
        return dict(elapsed=float(red[0].item()), e2e=e2e, extra=extra, pipelined=float(red[3].item()) == 0.0,
                    total_frames=float(tot[0].item()), tot_like=float(tot[1].item()),
                    n_ok=int(tot[2].item()), stats=dict(stats), kernel_ms=float(np.mean(kernel_ms)),
                    per_rank_kernel_ms=[float(x) for x in kms.tolist()], frames=frames, n_utts=n_utts,
                    longest=int(np.diff(off_h).max()) if n_utts else 0)

    weak = run(feats, off, args.steps, args.warmup)
    strong_rec = run(strong[0], strong[1], args.steps, args.warmup, end_to_end=False) if strong is not None else None

    if rank == 0:
        st = weak["stats"]
        fps = weak["total_frames"] * args.steps / weak["elapsed"]
        # roofline of the dominant kernel (DecodeKernel): algorithmic bytes per launch =
        # 60 B per expanded arc + 16 B per created token / measured duration
        alg_bytes = st["arcs"] * 60.0 + st["toks"] * 16.0
        tight_bytes = (st["arcs"] - st["cand"]) * 28.0 + st["cand"] * 60.0 + st["toks"] * 16.0
        k_ms = weak["kernel_ms"]
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        ref_order = bool(st.get("ref_order", True))
        search = "reference-order" if ref_order else "canonical-rule-E"
        traffic, traffic_src = measured_traffic(args, world, k_ms, ref_order)
        out = {
            # BASELINE.json's metric; value = frames/s of nnet2 forward + LatticeFasterDecoder, "rtf" = the
            # real-time factor per GPU (10 ms frames: rtf = 100 / frames-per-second-per-GPU).  Since round 6 the decoder
            # of `value` is the library's default: the reference's OWN iteration order (bit-exact against the line-by-line
            # oracle of LatticeFasterDecoder, tests/test_gpu_exact_order.py); `search` says which one was timed
            "metric": "frames/sec decoded + real-time factor, LibriSpeech nnet2 decode @1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": weak["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "search": search,
            "rtf": weak["elapsed"] * 100.0 / (weak["total_frames"] * args.steps) * world,
            "config": {"workload": "librispeech_nnet_a_structured" + ("_small" if args.small else ""),
                       "search": search,
                       "utts_per_gpu": weak["n_utts"], "frames_per_gpu": weak["frames"], "graph_states": int(g["num_states"]),
                       "graph_arcs": int(g["arc_offsets"][-1]), "graph": "HCLG-structured (prefix trees x 3-state HMM chains, "
                       "pushed LM costs, back-off epsilons); %d words, %d LM states" % (g["num_words"], g["num_hubs"]),
                       "nnet": "140-700-4x(3500/350)-12000-5800" if not args.small else "small",
                       "decoder": DECODE_CFG, "acwt": ACWT, "parallelism": "utterance-shard x%d" % world,
                       "rccl_world_size": world, "workload_build_s": t_build,
                       "steps_pipelined": bool(weak["pipelined"]),
                       "step": "forward pass + decode (%s) + best paths and lattice sizes of the whole shard, K steps pipelined" % search,
                       "step_detail": "K steps run as a binary's "
                               "main loop would: when step i's decode kernel has finished, a second host thread starts step "
                               "i + 1's forward pass (kh_decoder_set_after_launch) while the host finishes step i (lattices, best "
                               "paths) - K forward passes and K decodes inside the timed region; steps_pipelined = false "
                               "(KH_BENCH_NO_PIPELINE=1, or a background pass failed on some rank): strictly one after the other; "
                               "value_unpipelined is that figure in every run.  search = reference-order: LatticeFasterDecoder's own "
                               "iteration order (HashList order, running next_cutoff; bit-exact against oracle mode 0), the library's "
                               "default; value_canonical / roofline.canonical_* = the opt-in order-independent rule E (oracle mode 3)"},
            "search_stats": {"arcs_expanded_per_frame": st["arcs"] / weak["frames"], "tokens_per_frame": st["toks"] / weak["frames"],
                             "lattice_arcs_per_frame": st["lat_arcs"] / weak["frames"],
                             "lattice_states_per_frame": st["lat_states"] / weak["frames"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "DecodeKernel<reference order>" if ref_order else "DecodeKernel<canonical>", "kernel_ms": k_ms,
                         "search": search,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "arcs_expanded_per_launch": st["arcs"], "tokens_created_per_launch": st["toks"],
                         # the same kernel against a TIGHT byte count: an arc that is read and rejected costs its 16-byte
                         # record + 4 (score) + 8 (source token) = 28 B, only a materialised candidate the full 60 B
                         "candidates_materialised_per_launch": st["cand"],
                         "tight_bytes_per_launch": tight_bytes,
                         "frac_tight": tight_bytes / (k_ms * 1e-3) / 1e9 / 8000.0,
                         # ... and the bytes the fabric really moved (PMC record) over the peak
                         "frac_traffic": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         "traffic_over_tight": (traffic / tight_bytes) if traffic else None},
            "per_rank_kernel_ms": weak["per_rank_kernel_ms"],
            "loglike_per_frame": st["tot_like"] / weak["frames"],
        }
        ex = weak.get("extra") or {}
        if "unpipelined" in ex:
            u_ = ex["unpipelined"]
            out["value_unpipelined"] = weak["total_frames"] * u_["steps"] / u_["elapsed"]
        if "canonical" in ex:
            x_ = ex["canonical"]
            if "error" in x_:
                out["canonical"] = x_
            else:
                # the same step with the opt-in order-independent acceptance rule E (kh_decoder_set_reference_order(dec, 0);
                # the headline of rounds 1-5; bit-exact against oracle mode 3, NOT the reference's lattices in general);
                # measured one step after the other, to be held against value_unpipelined.  Flat scalars: the driver's
                # record keeps neither nested dictionaries nor long strings
                out["value_canonical"] = weak["total_frames"] * x_["steps"] / x_["elapsed"]
                x_alg = 60.0 * x_["arcs"] + 16.0 * x_["toks"]
                x_tight = 28.0 * (x_["arcs"] - x_["cand"]) + 60.0 * x_["cand"] + 16.0 * x_["toks"]
                out["roofline"]["canonical_kernel_ms"] = x_["kernel_ms"]
                out["roofline"]["canonical_frac"] = x_alg / (x_["kernel_ms"] * 1e-3) / 1e9 / 8000.0
                out["roofline"]["canonical_frac_tight"] = x_tight / (x_["kernel_ms"] * 1e-3) / 1e9 / 8000.0
                out["roofline"]["kernel_ms_over_canonical"] = k_ms / x_["kernel_ms"]
                out["canonical"] = {
                    "unit": "frames/s", "steps": x_["steps"], "ms_per_step": x_["elapsed"] / x_["steps"] * 1e3,
                    "kernel_ms": x_["kernel_ms"],
                    "arcs_expanded_per_frame": x_["arcs"] / weak["frames"], "tokens_per_frame": x_["toks"] / weak["frames"],
                    "candidates_materialised_per_frame": x_["cand"] / weak["frames"],
                    "lattice_arcs_per_frame": x_["lat_arcs"] / weak["frames"],
                    "loglike_per_frame": x_["tot_like"] / weak["frames"],
                    "note": "kh_decoder_set_reference_order(dec, 0): the order-independent rule, an opt-in since round 6"}
        if weak["e2e"] is not None:
            e = weak["e2e"]
            # DecodeUtteranceLatticeFaster in full: what the reference binary's timer brackets per utterance
            # (nnet-latgen-faster.cc:139-160 + decoder-wrappers.cc:232-284), inputs = features resident in HBM
            out["value_end_to_end"] = weak["total_frames"] * args.steps / e["elapsed"]
            out["end_to_end"] = {
                "unit": "frames/s", "ms_per_step": e["elapsed"] / args.steps * 1e3, "kernel_ms": e["kernel_ms"],
                "host_tail_ms": e["tail_ms"],
                "region": "nnet2 forward + LatticeFasterDecoder + GetRawLattice + GetBestPath + DeterminizeLatticePhonePrunedWrapper "
                          "(lattice-beam %g, phone + word passes) -> CompactLattice for every utterance; determinization on host threads "
                          "(as many as the container's CPU quota), started per "
                          "utterance as the decode kernel exports it; host_tail_ms = wall time the host threads still needed "
                          "after the kernel had finished (what the overlap with the kernel does not hide).  pipelined = true: the K "
                          "steps run as a binary's main loop would - step i + 1's forward pass is started when step i's decode kernel has "
                          "finished (kh_decoder_set_after_launch) and runs under step i's host tail; K forward passes, K decodes and "
                          "K sets of CompactLattices inside the timed region (KH_BENCH_NO_PIPELINE=1: one after the other)"
                          % DECODE_CFG["lattice_beam"],
                "pipelined": bool(e.get("pipelined")),
                "compact_lattices": e["clat"]}
            if "wave_elapsed" in e:
                # config 4's own binary starts from audio: MFCC + online iVectors + the region above
                out["value_wave_to_lattice"] = weak["total_frames"] * args.steps / e["wave_elapsed"]
                out["wave_to_lattice"] = {
                    "unit": "frames/s", "ms_per_step": e["wave_elapsed"] / args.steps * 1e3, "front_end_ms": e["front_end_ms"],
                    "region": "online2-wav-nnet2-latgen-faster's work per utterance set: MFCC (40 x 40 hires) and online iVector "
                              "extraction (512-Gaussian UBM, 100 dims, period 10) of audio of the shard's duration, then the "
                              "end_to_end region.  The synthetic decode workload's features are not derived from audio: the front "
                              "end runs on noise of the same length (same kernels, same work), its output is not what is decoded"}
        if strong_rec is not None:
            s = strong_rec
            out["strong_scaling"] = {
                "value": s["total_frames"] * args.steps / s["elapsed"], "unit": "frames/s", "scaling": "strong",
                "utterances_total": s["n_ok"], "frames_total": s["total_frames"], "ms_per_step": s["elapsed"] / args.steps * 1e3,
                "per_rank_kernel_ms": s["per_rank_kernel_ms"], "rank0_utterances": s["n_utts"], "rank0_frames": s["frames"],
                "longest_utterance_frames": s["longest"],
                "note": "one 2620-utterance set sharded longest-first over the ranks (sharding.partition_utterances)"}
        if cpu_handle is not None:                   # the CPU legs run at N = 1 only
            res = finish_cpu_baseline(cpu_handle)
            if "error" in res:
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "sample": res["error"]}
            else:
                out["cpu_baseline"] = {"value": res["value"], "unit": "frames/s", "cores": 1, "kind": res["kind"],
                                       "sample": res["sample"], "all_cores": res.get("all_cores"),
                                       "secondary": res.get("secondary")}
                out["speedup_vs_cpu_1thread_per_gpu"] = fps / world / res["value"]
        # The secondary legs (configs 2, 3, 5, iVectors, config 4 as a serving loop) run LAST, when the headline and its
        # baselines are final, and in a CHILD process (round 6): this process keeps the line, whatever a leg does - a leg that
        # does not come back (the child's own watchdog prints what it has and names the leg) or one that faults on the GPU
        # and takes its process with it (round 6 saw ONE memory-access fault in ~60 runs of the serving legs: the headline of a
        # driver run must not depend on that).
        if world == 1 and not args.no_secondary and not args.small:   # (at N > 1 the other ranks would wait at the final barrier)
            import subprocess
            try:
                sec = importlib.import_module("tools.bench_secondary")
                sec.release_device_memory(api, torch)   # (the child sizes its arenas from what is free)
            except Exception:   # noqa: BLE001
                pass
            cmd = [sys.executable, os.path.abspath(__file__), "--secondary-only", "--utts", str(args.utts),
                   "--graph-states", str(args.graph_states), "--secondary-timeout", str(args.secondary_timeout)]
            secondary = None
            try:
                import tempfile
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
                # The child's stdout goes to a FILE and the child is never waited for beyond its deadline: a process whose
                # kernel hangs on the GPU may not even die when killed (its exit waits for the queue's teardown) - the
                # line it printed before is in the file either way.
                with tempfile.NamedTemporaryFile(prefix="bench_secondary_", suffix=".json", delete=False) as tf:
                    tf_name = tf.name
                with open(tf_name, "wb") as child_out:
                    proc = subprocess.Popen(cmd, stdout=child_out, env=env)
                deadline = time.perf_counter() + args.secondary_timeout + 150
                while proc.poll() is None and time.perf_counter() < deadline:
                    time.sleep(0.5)
                rc = proc.poll()
                if rc is None:
                    proc.kill()
                    for _ in range(20):      # (bounded: 10 s)
                        if proc.poll() is not None:
                            break
                        time.sleep(0.5)
                    rc = proc.poll()
                with open(tf_name, "rb") as f:
                    lines = [l for l in f.read().decode(errors="replace").splitlines() if l.startswith("{")]
                try:
                    os.unlink(tf_name)
                except OSError:
                    pass
                if lines:
                    secondary = json.loads(lines[-1]).get("secondary")
                if secondary is None:
                    secondary = {"error": "the secondary legs' process printed no result (exit status %r)" % (rc,)}
                if rc != 0:
                    secondary["child_exit_status"] = rc if rc is not None else "still running when bench.py left (killed, not reaped)"
                    out["secondary_failed"] = True
            except Exception as e:   # noqa: BLE001 - the secondary legs never fail the headline run
                secondary = {"error": repr(e)}
                out["secondary_failed"] = True
            out["secondary"] = secondary
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
