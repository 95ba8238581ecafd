#!/usr/bin/env python3
"""Config 5's discriminative pipeline leg alone (tools/bench_secondary.py lattice_fb_cfg5), the two-stream schedule against
the one-stream one (KH_LATTICE_ONE_STREAM=1: everything queued behind the forward pass, as until round 5):
python tools/time_cfg5.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "old-kaldi-git_amd"


def main():
    import torch
    api = importlib.import_module(PKG + ".api")
    api.select_gpu(0)
    sec = importlib.import_module("tools.bench_secondary")
    out = {}
    for name, env in (("two_streams", None), ("one_stream", "1"), ("two_streams_again", None)):
        if env is None:
            os.environ.pop("KH_LATTICE_ONE_STREAM", None)
        else:
            os.environ["KH_LATTICE_ONE_STREAM"] = env
        r = sec.lattice_fb_cfg5(api, torch)
        out[name] = {k: r[k] for k in r if k.startswith("pipeline_")}
        print(name, json.dumps(out[name]), flush=True)


if __name__ == "__main__":
    main()
