"""Multi-GPU host logic: utterance sharding + the final scalar reduction.

The reference parallelises decoding only across utterances — `split_data.sh` +
`run.pl JOB=1:$nj` (egs/wsj/s5/steps/nnet2/decode.sh:130-136) or
TaskSequencer threads (nnet2bin/nnet-latgen-faster-parallel.cc:91,134-135) — and
combines nothing but scalar totals (nnet-latgen-faster.cc:100-101,133-135,
179-186).  Here: one process per GPU, each decodes its shard with no data-path
collective; one all-reduce (RCCL on GPUs, gloo in the CPU tests) of the totals."""
import numpy as np
import torch
import torch.distributed as dist


def partition_utterances(lengths, world_size):
    """Greedy longest-first assignment of utterances to ranks.
    Returns a list of index arrays, one per rank; deterministic."""
    lengths = np.asarray(lengths, np.int64)
    order = np.argsort(-lengths, kind="stable")
    loads = np.zeros(world_size, np.int64)
    bins = [[] for _ in range(world_size)]
    for u in order:
        r = int(np.argmin(loads))          # ties -> lowest rank
        bins[r].append(int(u))
        loads[r] += lengths[u]
    return [np.asarray(b, np.int64) for b in bins]


def reduce_decode_totals(frames, tot_like, num_success, num_fail, elapsed, device="cpu", group=None):
    """Sum {frame_count, tot_like, num_success, num_fail} over ranks and take the
    MAX of the elapsed time, as the parent process does after `wait`
    (nnet-latgen-faster.cc:179-186 prints them).  Works without an initialised
    process group (single rank)."""
    sums = torch.tensor([float(frames), float(tot_like), float(num_success), float(num_fail)],
                        dtype=torch.float64, device=device)
    mx = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    frames_t, like_t, ok_t, fail_t = [float(x) for x in sums.tolist()]
    el = float(mx.item())
    return dict(frames=frames_t, tot_like=like_t, num_success=int(ok_t), num_fail=int(fail_t), elapsed=el,
                frames_per_sec=frames_t / el if el > 0 else float("inf"),
                # "real-time factor assuming 100 frames/sec" nnet-latgen-faster.cc:179-182
                rtf=el * 100.0 / frames_t if frames_t > 0 else float("nan"),
                loglike_per_frame=like_t / frames_t if frames_t > 0 else float("nan"))
