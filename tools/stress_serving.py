#!/usr/bin/env python3
"""Stress harness for the persistent serving kernel: the serving legs of the benchmark back to back.

    python tools/stress_serving.py [--runs 50] [--streams 256] [--graph-states 2000000] [--idle-ms 2000] [--closure-cap N]

Conditions of the round-4 stalls: the legs ran with the CPU baseline's forked child still alive (a child forked BEFORE the GPU
is touched, blocked on a pipe - reproduced here), right after other legs on the same device.  Every run = the two serving
legs (reference pruning schedule and lazy schedule, 0.05 s chunks, two passes each over the utterance set, slots reused);
a run counts as clean when every leg returns, every utterance is served and no library call reports KH_ETIMEOUT.  With
--idle-ms small the grid's idle decision is exercised all the time; --closure-cap 0 sends every frame through the general
closure routine (its "pending" spin was a suspect).  One line per run, a summary at the end (profiles/r05_serving_stress.txt)."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=50)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--graph-states", type=int, default=2_000_000)
    ap.add_argument("--idle-ms", type=float, default=None)
    ap.add_argument("--closure-cap", type=int, default=None)
    ap.add_argument("--interval-only", action="store_true", help="only the leg with the reference pruning schedule (skip the lazy one)")
    args = ap.parse_args()
    if args.idle_ms is not None:
        os.environ["KH_SERVE_IDLE_MS"] = str(args.idle_ms)
    if args.closure_cap is not None:
        os.environ["KH_DECODER_CLOSURE_CAP"] = str(args.closure_cap)
    os.environ.setdefault("KH_SERVE_TIMEOUT_MS", "20000")
    if args.interval_only:
        os.environ["KH_STRESS_SKIP_LAZY"] = "1"
    # the CPU baseline's stand-in: a child forked before the GPU is initialised, alive and blocked on its pipe throughout
    rfd, wfd = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(wfd)
        os.read(rfd, 1)
        os._exit(0)
    os.close(rfd)
    import numpy as np
    import torch
    api = importlib.import_module("old-kaldi-git_amd.api")
    sec = importlib.import_module("tools.bench_secondary")
    bench = importlib.import_module("bench")
    api.select_gpu(0)
    t0 = time.perf_counter()
    net, priors, g, protos = bench.build_model_and_graph(3456, args.graph_states, False)
    feats, off = bench.build_utterances(3456, 0, 3 * args.streams, net, g, protos, False)
    print("# workload: %d streams, %d utterances, graph %d states, built in %.0f s; idle %s ms, closure cap %s" % (
        args.streams, len(off) - 1, int(g["num_states"]), time.perf_counter() - t0, os.environ.get("KH_SERVE_IDLE_MS", "2000"),
        os.environ.get("KH_DECODER_CLOSURE_CAP", "default")), flush=True)
    wl = (net, priors, g, feats, off, bench.DECODE_CFG, bench.ACWT)
    clean = 0
    worst = 0.0
    for r in range(args.runs):
        t1 = time.perf_counter()
        try:
            res = sec.online2_cfg4(api, torch, workload=wl, streams=args.streams, chunks=(5,), only_persistent=True)
            a = res["chunk_5_frames_persistent"]
            b = res.get("chunk_5_frames_persistent_lazy", a)
            ok = a["utterances_served"] > 0 and b["utterances_served"] > 0
            clean += 1 if ok else 0
            worst = max(worst, a["chunk_latency_ms"]["max"], b["chunk_latency_ms"]["max"])
            print("run %3d: %s  %.1f s  served %d + %d  frames/s %.0f / %.0f  chunk latency ms p50 %.1f / %.1f  p95 %.1f / %.1f  max %.0f / %.0f  "
                  "last chunk + FinalizeDecoding max %.0f / %.0f  host step call ms %.2f / %.2f  streams per call %.0f / %.0f" % (
                r, "clean" if ok else "INCOMPLETE", time.perf_counter() - t1, a["utterances_served"], b["utterances_served"],
                a["frames_per_s"] or 0, b["frames_per_s"] or 0, a["chunk_latency_ms"]["p50"], b["chunk_latency_ms"]["p50"],
                a["chunk_latency_ms"]["p95"], b["chunk_latency_ms"]["p95"], a["chunk_latency_ms"]["max"], b["chunk_latency_ms"]["max"],
                a["last_chunk_and_finalize_ms"]["max"], b["last_chunk_and_finalize_ms"]["max"], a["step_call_ms"]["mean"], b["step_call_ms"]["mean"],
                a["streams_per_step_call"], b["streams_per_step_call"]), flush=True)
        except Exception as e:   # noqa: BLE001 - the harness records, it does not stop
            print("run %3d: FAILED after %.1f s: %r" % (r, time.perf_counter() - t1, e), flush=True)
        sec.release_device_memory(api, torch)
    print("# %d of %d runs clean; worst chunk latency %.0f ms" % (clean, args.runs, worst), flush=True)
    os.write(wfd, b"x")
    os.waitpid(pid, 0)
    return 0 if clean == args.runs else 1


if __name__ == "__main__":
    sys.exit(main())
