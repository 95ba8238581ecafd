"""GPU parity for the online call sequence (LatticeFasterOnlineDecoder,
decoder/lattice-faster-online-decoder.{h,cc}) through the C-ABI: concurrent streams
advanced in chunks against the oracle's restatement — bit-exact lattices after
FinalizeDecoding and at intermediate points (un-finalized GetRawLattice with and
without final probs)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import binding as B
from test_gpu_decoder import assert_same_best_path, assert_same_lattice

pytestmark = pytest.mark.gpu
workloads = importlib.import_module("old-kaldi-git_amd.workloads")


def test_streams_in_chunks_match_one_shot_decoding(api):
    rng = np.random.default_rng(41)
    g = workloads.make_hclg_like(rng, 30000, 300)
    Ts = [97, 160, 33, 211]
    lls = [workloads.make_loglikes(rng, T, 300) for T in Ts]
    cfg = api.decoder_config(beam=12.0, max_active=1500, min_active=100, lattice_beam=6.0)
    fst = api.Fst(g)
    dev = [torch.from_numpy(x).cuda() for x in lls]
    dec = api.LatticeFasterOnlineDecoder(fst, cfg, num_streams=len(Ts), max_frames=max(Ts))
    dec.init_decoding(list(range(len(Ts))))
    pos = [0] * len(Ts)
    snapshots = {}
    step = 0
    while any(p < T for p, T in zip(pos, Ts)):
        # a random subset of the unfinished streams advances by a random chunk (0 allowed)
        act = [s for s in range(len(Ts)) if pos[s] < Ts[s] and rng.random() < 0.8]
        if not act:
            continue
        n = [int(min(Ts[s] - pos[s], rng.integers(0, 40))) for s in act]
        dec.advance_decoding(act, [dev[s][pos[s]:pos[s] + k] for s, k in zip(act, n)])
        for s, k in zip(act, n):
            pos[s] += k
            assert dec.num_frames_decoded(s) == pos[s]
        step += 1
        if step in (2, 5):  # look at a stream mid-utterance
            s = act[0]
            if pos[s] > 0:
                snapshots[(s, pos[s])] = (dec.get_raw_lattice(s, use_final_probs=False),
                                          dec.get_raw_lattice(s, use_final_probs=True),
                                          dec.get_best_path(s, use_final_probs=True))
    dec.finalize_decoding(list(range(len(Ts))))
    # offline decoder on the same data
    off = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int32)
    offline = api.LatticeFasterDecoder(fst, cfg, max_batch=len(Ts), max_frames=max(Ts))
    offline.decode(torch.from_numpy(np.concatenate(lls, 0)).cuda(), off)
    for s, x in enumerate(lls):
        oc = B.DecoderOracle(g, cfg, "canonical")
        assert oc.decode(x)
        got = dec.get_raw_lattice(s)
        assert_same_lattice(got, oc.raw_lattice())
        assert_same_lattice(got, offline.get_raw_lattice(s))
        assert_same_best_path(dec.get_best_path(s), oc.best_path())
        so, sg = oc.stats(), dec.stats(s)
        for k in ("num_frames", "reached_final", "num_tokens", "num_links", "tokens_created", "max_tokens_frame"):
            assert so[k] == sg[k], (k, so[k], sg[k])
        assert np.float32(so["final_relative_cost"]).tobytes() == np.float32(sg["final_relative_cost"]).tobytes()
    assert snapshots
    for (s, t), (l_nofinal, l_final, bp) in snapshots.items():
        oc = B.DecoderOracle(g, cfg, "canonical")
        oc.begin(lls[s])
        assert oc.advance(t) == t
        oc.snapshot(False)
        assert_same_lattice(l_nofinal, oc.raw_lattice())
        oc.snapshot(True)
        assert_same_lattice(l_final, oc.raw_lattice())
        assert_same_best_path(bp, oc.best_path())


def test_call_sequence_errors_and_stream_reuse(api):
    rng = np.random.default_rng(42)
    g = workloads.make_hclg_like(rng, 2000, 40)
    cfg = api.decoder_config(beam=9.0, lattice_beam=5.0)
    fst = api.Fst(g)
    dec = api.LatticeFasterOnlineDecoder(fst, cfg, num_streams=2, max_frames=64)
    x = torch.from_numpy(workloads.make_loglikes(rng, 30, 40)).cuda()
    with pytest.raises(api.KhError):  # AdvanceDecoding before InitDecoding (.cc:749-750)
        dec.advance_decoding([0], [x[:5]])
    with pytest.raises(api.KhError):  # duplicate stream in one call
        dec.init_decoding([1, 1])
    dec.init_decoding([0])
    with pytest.raises(api.KhError):  # no frame decoded yet (.cc:171)
        dec.get_raw_lattice(0)
    dec.advance_decoding([0], [x[:30]])
    with pytest.raises(api.KhError):  # beyond max_frames
        dec.advance_decoding([0], [torch.cat([x, x, x])[:40]])
    dec.finalize_decoding([0])
    with pytest.raises(api.KhError):  # GetRawLattice(use_final_probs=false) after FinalizeDecoding (.cc:156-158)
        dec.get_raw_lattice(0, use_final_probs=False)
    with pytest.raises(api.KhError):  # AdvanceDecoding after FinalizeDecoding
        dec.advance_decoding([0], [x[:5]])
    first = dec.get_raw_lattice(0)
    # the stream is reused for the next utterance; same input -> same lattice
    dec.init_decoding([0])
    dec.advance_decoding([0], [x[:11]])
    dec.advance_decoding([0], [x[11:30]])
    dec.finalize_decoding([0])
    assert_same_lattice(dec.get_raw_lattice(0), first)
    oc = B.DecoderOracle(g, cfg, "canonical")
    assert oc.decode(x.cpu().numpy())
    assert_same_lattice(first, oc.raw_lattice())
