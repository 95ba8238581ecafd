"""GPU: bench.py's timed loops.  The K steps run pipelined (step i + 1's forward pass is enqueued behind step i's decode
kernel from a second host thread, kh_decoder_set_after_launch); KH_BENCH_NO_PIPELINE=1 runs them one after the other.
Both must produce the same decode (the best paths' log-likelihood sum, the lattice sizes, the CompactLattice totals)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_extra, want_stderr=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", "--utts", "300"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    if want_stderr is not None:
        assert want_stderr in p.stderr.decode()
    return json.loads(lines[0])


def test_pipelined_steps_decode_the_same_as_sequential_ones():
    a, b = _bench({}), _bench({"KH_BENCH_NO_PIPELINE": "1"})
    assert a["config"]["steps_pipelined"] is True and b["config"]["steps_pipelined"] is False
    assert a["end_to_end"]["pipelined"] is True and b["end_to_end"]["pipelined"] is False
    assert a["loglike_per_frame"] == b["loglike_per_frame"]
    assert a["search_stats"] == b["search_stats"]
    # the driver's record must say which search was timed, in short scalars: the library's default = the reference's own order
    assert a["search"] == b["search"] == a["config"]["search"] == a["roofline"]["search"] == "reference-order"
    assert a["value_canonical"] > 0 and a["roofline"]["canonical_kernel_ms"] > 0 and 0 < a["roofline"]["canonical_frac"] < 1
    assert len(a["config"]["step"]) < 120
    assert a["end_to_end"]["compact_lattices"] == b["end_to_end"]["compact_lattices"]
    assert a["end_to_end"]["compact_lattices"]["incomplete"] == 0
    for d in (a, b):
        assert d["value"] > 0 and d["value_end_to_end"] > 0 and d["roofline"]["kernel"] == "DecodeKernel<reference order>"


def test_a_failing_background_forward_pass_falls_back_to_sequential_steps():
    """The overlap must never cost the run: when the second host thread fails, the steps go on one after the other
    (the scores are computed by the main thread) and the decode is the same."""
    a = _bench({"KH_BENCH_FAIL_BACKGROUND": "1"}, want_stderr="steps run unpipelined from here")
    b = _bench({"KH_BENCH_NO_PIPELINE": "1"})
    assert a["loglike_per_frame"] == b["loglike_per_frame"] and a["search_stats"] == b["search_stats"]
    assert a["end_to_end"]["compact_lattices"] == b["end_to_end"]["compact_lattices"]
