#!/usr/bin/env python3
"""The online2 (config 4 serving) leg of bench.py alone: python tools/bench_online2.py [streams]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

api = importlib.import_module("old-kaldi-git_amd.api")
sec = importlib.import_module("tools.bench_secondary")
api.select_gpu(0)
print(json.dumps(sec.online2_cfg4(api, torch, None, streams=int(sys.argv[1]) if len(sys.argv) > 1 else 256), indent=1))
