"""CPU, world_size 2 over gloo: the N > 1 host path (utterance sharding + the final
scalar reduction) that bench.py / a multi-GPU decode job uses."""
import importlib
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sharding = importlib.import_module("old-kaldi-git_amd.sharding")


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    lens = rng.integers(100, 3500, 2620)
    parts = sharding.partition_utterances(lens, 8)
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(2620))          # every utterance exactly once
    loads = np.array([lens[p].sum() for p in parts])
    assert loads.max() - loads.min() <= lens.max()           # LPT bound
    assert np.array_equal(sharding.partition_utterances(lens, 1)[0], np.argsort(-lens, kind="stable"))


def _worker(rank, world, port, q):
    # file rendezvous: a TCP port picked by bind-and-close can be taken by a parallel test (pytest -n)
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world)
    lens = np.random.default_rng(1).integers(50, 400, 37)
    mine = sharding.partition_utterances(lens, world)[rank]
    frames = int(lens[mine].sum())
    out = sharding.reduce_decode_totals(frames=frames, tot_like=-2.5 * frames, num_success=len(mine), num_fail=rank,
                                        elapsed=1.0 + rank)
    q.put((rank, out, int(lens.sum())))
    dist.destroy_process_group()


def test_two_rank_reduction_over_gloo():
    port = os.path.join(tempfile.mkdtemp(prefix="kh_gloo_"), "rendezvous")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, total in res:
        assert out["frames"] == total                          # sum over shards == all frames
        assert out["num_success"] == 37 and out["num_fail"] == 1
        assert out["elapsed"] == 2.0                           # MAX over ranks
        assert abs(out["loglike_per_frame"] + 2.5) < 1e-12
        assert abs(out["rtf"] - 2.0 * 100.0 / total) < 1e-12


def test_single_process_without_group():
    out = sharding.reduce_decode_totals(1000, -3000.0, 4, 0, 0.5)
    assert out["frames_per_sec"] == 2000.0 and out["rtf"] == 0.05


def _disc_worker(rank, world, port, q):
    # file rendezvous: a TCP port picked by bind-and-close can be taken by a parallel test (pytest -n)
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world)
    stats = dict(tot_t=100.0 * (rank + 1), tot_t_weighted=50.0 * (rank + 1), tot_num_count=3.0 + rank,
                 tot_num_objf=-7.0 * (rank + 1), tot_den_objf=-9.0 * (rank + 1))
    grads = [torch.full((5, 3), float(rank + 1)), torch.full((11,), 10.0 * (rank + 1)), torch.full((2, 2), -1.0)]
    out = sharding.reduce_discriminative(stats, grads, bucket_bytes=64)   # tiny buckets: several all-reduces
    q.put((rank, out, [g.clone() for g in grads]))
    dist.destroy_process_group()


def test_discriminative_stats_and_gradients_over_gloo():
    """Config 5: NnetDiscriminativeStats::Add + the gradient sum across ranks (bucketed all-reduce)."""
    port = os.path.join(tempfile.mkdtemp(prefix="kh_gloo_"), "rendezvous")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_disc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, grads in res:
        assert out["tot_t"] == 300.0 and out["tot_t_weighted"] == 150.0 and out["tot_num_count"] == 7.0
        assert out["tot_num_objf"] == -21.0 and out["tot_den_objf"] == -27.0
        assert abs(out["objf_per_frame"] - (-21.0 + 27.0) / 150.0) < 1e-12
        assert torch.all(grads[0] == 3.0) and torch.all(grads[1] == 30.0) and torch.all(grads[2] == -2.0)
    # single process: identity
    out = sharding.reduce_discriminative(dict(tot_t=1, tot_t_weighted=1, tot_num_count=0, tot_num_objf=0, tot_den_objf=-2.0))
    assert out["tot_den_objf"] == -2.0 and out["objf_per_frame"] == -2.0
