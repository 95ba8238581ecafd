"""iVector extraction rate alone (the ivector_f3 leg of tools/bench_secondary.py), e.g. under
rocprofv3 --kernel-trace --stats -- python3 tools/bench_ivector.py"""
import importlib
import json
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("old-kaldi-git_amd.api")
api.select_gpu(0)
bs = importlib.import_module("tools.bench_secondary")
print(json.dumps(bs.ivector_f3(api, torch)))
