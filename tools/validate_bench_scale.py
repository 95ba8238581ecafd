"""One-off check at the benchmark's own scale: the bench.py workload (HCLG-structured 10 M-state
graph, beam 15 / max-active 7000, real forward pass) decoded in one launch with more utterances
than slots, and a sample of utterances — the longest, first and last in their slot's queue,
a median one — compared bit-exactly with the canonical oracle on the SAME log-likelihood rows.
    python tools/validate_bench_scale.py [n_utts=700] [reference]
"reference": the kernel in LatticeFasterDecoder's own iteration order (kh_decoder_set_reference_order) against the line-by-line
restatement (oracle mode 0)."""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bench
from oracle import binding as B
from test_gpu_decoder import assert_same_best_path, assert_same_lattice

api = importlib.import_module("old-kaldi-git_amd.api")
api.select_gpu(0)
n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 700
MODE = "reference" if len(sys.argv) > 2 and sys.argv[2] == "reference" else "canonical"
t0 = time.time()
net, priors, g, protos = bench.build_model_and_graph(3456, 10_000_000, False)
feats, off = bench.build_utterances(3456, 0, n_utts, net, g, protos, False)
print("workload built in %.0f s: %d utterances, %d frames" % (time.time() - t0, n_utts, off[-1]))
nnet = api.Nnet(net, priors)
fst = api.Fst(g)
cfg = api.decoder_config(**bench.DECODE_CFG)
dec = api.LatticeFasterDecoder(fst, cfg, max_batch=n_utts, max_frames=int(np.diff(off).max()), exact_reference_order=MODE == "reference")
print("decoding order:", MODE)
n_pdf = net[-1]["output_dim"]
ll = torch.empty((int(off[-1]), n_pdf), dtype=torch.float32, device="cuda")
bench.forward_all(nnet, torch.from_numpy(feats).cuda(), off, ll, max_rows=60000)
tid_phone = np.zeros(len(g["tid2pdf"]), np.int32)
tid_phone[1::2] = 1 + g["tid2pdf"][1::2]
dec.set_determinize(True, bench.DECODE_CFG["lattice_beam"], tid_phone=tid_phone)    # DeterminizeLatticePhonePrunedWrapper on the completion threads
dec.decode(ll, off)
dec.prepare()
import lattice_equiv as LE
lens = np.diff(off)
sample = [0, 3, n_utts // 2, n_utts - 2, n_utts - 1, int(np.argmin(np.abs(lens - np.median(lens))))]
for u in sample:
    x = ll[off[u]:off[u + 1]].cpu().numpy()
    t1 = time.time()
    oc = B.DecoderOracle(g, cfg, MODE)
    assert oc.decode(x)
    assert_same_lattice(dec.get_raw_lattice(u), oc.raw_lattice())
    assert_same_best_path(dec.get_best_path(u), oc.best_path())
    st = dec.stats(u)
    # the CompactLattice the completion thread determinized under the kernel against the restated reference determinizer
    want_c = B.determinize_lattice_phone_pruned(oc.raw_lattice(), bench.DECODE_CFG["lattice_beam"], tid_phone)
    got_c = dec.get_compact_lattice(u)
    res = LE.compare_deterministic(got_c, want_c, delta=1e-3)
    assert got_c["n_states"] == want_c["n_states"] and len(got_c["arc_src"]) == len(want_c["arc_src"]) and LE.deterministic_equal(res), res
    sc = dec.schedule_counters(u)
    print("utterance %d (%d frames, %d lattice states, %d arcs, max %d tokens/frame; %d garbage collections, final pass: %d frames dense in LDS, "
          "%d through the general routines, %d hand-offs through memory): raw lattice and best path bit-exact, determinized lattice "
          "identical (%d states, %d arcs) (oracle %.0f s)" %
          (u, lens[u], st["num_tokens"], st["num_links"], st["max_tokens_frame"], sc["garbage_collections"], sc["dense_final_visits"], sc["general_final_visits"], sc["handoffs_through_memory"], got_c["n_states"],
           len(got_c["arc_src"]), time.time() - t1))
print("OK")
