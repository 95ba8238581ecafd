import sys, importlib, json, torch
sys.path.insert(0,'.')
api = importlib.import_module("old-kaldi-git_amd.api")
api.select_gpu(0)
bs = importlib.import_module("tools.bench_secondary")
r = bs.lattice_fb_cfg5(api, torch)
print(json.dumps({k: r[k] for k in r if k.startswith("pipeline") or k.startswith("kh_")}, indent=1))
