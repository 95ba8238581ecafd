"""CPU: bench.py's multi-rank path.  `--gpus 2` without a torchrun environment must start
the two ranks itself; in --no-gpu-dryrun mode they form a gloo group, shard ONE utterance
set with sharding.partition_utterances and reduce the totals (no GPU, nothing decoded)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


def test_launcher_starts_ranks_and_reduces():
    rc, out, err = _run(["--gpus", "2", "--utts", "200", "--no-gpu-dryrun", "--master-port", "29731"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["utterances"] == 200
    assert out["frames"] == out["expected_frames"]            # the two shards cover the set exactly once
    assert 0.4 < out["per_rank_frames_rank0"] / out["frames"] < 0.6   # longest-first greedy balances the load


def test_single_rank_dryrun():
    rc, out, err = _run(["--gpus", "1", "--utts", "50", "--no-gpu-dryrun"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 1 and out["utterances"] == 50


def test_rank_failure_is_reported():
    """A rank that dies makes the launcher exit non-zero (never a silent partial result)."""
    rc, out, err = _run(["--gpus", "2", "--utts", "20", "--no-gpu-dryrun", "--master-port", "29733"],
                        {"BENCH_FAIL_RANK": "1"})
    assert rc != 0 and out is None


def test_secondary_legs_report_progress_for_the_watchdog(monkeypatch):
    """bench.py runs the secondary legs last and under a watchdog that prints the line without a leg that does not come
    back: tools.bench_secondary.run_all fills the caller's dict leg by leg and names the leg that is running."""
    import importlib
    sec = importlib.import_module("tools.bench_secondary")
    seen = []
    state = {"leg": None}

    def leg(name, fail=False):
        def fn(api, torch, *a):
            seen.append((name, state["leg"]))
            if fail:
                raise RuntimeError("boom")
            return {"ok": name}
        return fn

    monkeypatch.setattr(sec, "release_device_memory", lambda api, torch: None)
    for name in ("gmm_cfg2", "nnet_cfg3", "decode_cfg3", "lattice_fb_cfg5", "ivector_f3"):
        monkeypatch.setattr(sec, name, leg(name, fail=(name == "decode_cfg3")))
    monkeypatch.setattr(sec, "online2_cfg4", leg("online2_cfg4"))
    out = {}
    ret = sec.run_all(None, None, main_workload=None, out=out, state=state)
    assert ret is out and state["leg"] is None
    assert [s[0] for s in seen] == ["gmm_cfg2", "nnet_cfg3", "decode_cfg3", "lattice_fb_cfg5", "ivector_f3", "online2_cfg4"]
    assert all(name == running for name, running in seen)          # the watchdog would have named the right leg
    assert out["gmm_cfg2"] == {"ok": "gmm_cfg2"} and "boom" in out["decode_cfg3"]["error"] and out["online2_cfg4"] == {"ok": "online2_cfg4"}
