"""Generate golden input/output vectors from the REFERENCE's own CPU code.

Runs only in the build container (needs /root/reference): it loads
oracle/_ref/libkaldi_ref.so — the reference's base/matrix/cudamatrix(CPU
branch)/gmm/nnet2 sources compiled where they lie by oracle/Makefile — feeds it
seeded inputs, and stores inputs + the reference's outputs as small .npz
fixtures next to this script.  The fixtures are data only; nothing from
/root/reference travels.

    python tests/golden/make_golden.py

Cases follow SURVEY.md §8c and the reference's own unit tests
(cudamatrix/cu-matrix-test.cc: AddMatMat :1038-1066, Softmax :1559-1587,
CopyRows :379-402 with -1 indices, GroupPnorm :246, SumColumnRanges :442,
Lookup :2011; cu-math-test.cc Splice; gmm/diag-gmm-test.cc:93-188;
nnet2/nnet-compute-test.cc:28-80).
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import binding  # noqa: E402

workloads = importlib.import_module("old-kaldi-git_amd.workloads")


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def main():
    binding.build(ref=True)
    ref = binding.OracleLib("ref")
    rng = np.random.default_rng(20151001)
    f32 = np.float32

    # ---- a1 AddMatMat: NN/NT/TN/TT, alpha/beta in {0, 0.5, 1}, odd sizes
    cases = {}
    i = 0
    for (m, n, k) in [(37, 23, 50), (70, 40, 90), (1, 7, 3), (65, 129, 17)]:
        for tA in (0, 1):
            for tB in (0, 1):
                alpha = [1.0, 0.5, 1.0, 0.5][i % 4]
                beta = [0.0, 1.0, 0.5, 0.0][(i // 2) % 4]
                A = rng.standard_normal((k, m) if tA else (m, k)).astype(f32)
                B = rng.standard_normal((n, k) if tB else (k, n)).astype(f32)
                Cm = rng.standard_normal((m, n)).astype(f32)
                out = ref.add_mat_mat(alpha, A, tA, B, tB, beta, Cm)
                cases["c%d_A" % i], cases["c%d_B" % i], cases["c%d_C" % i] = A, B, Cm
                cases["c%d_out" % i] = out
                cases["c%d_par" % i] = np.array([alpha, beta, tA, tB], f32)
                i += 1
    cases["n"] = np.array(i)
    save("add_mat_mat", **cases)

    # ---- a2 softmax / log-softmax: cols 1, 10, 256, 257, 3000; randn*5 and +-80 extremes
    cases = {}
    for j, cols in enumerate([1, 10, 59, 256, 257, 3000]):
        X = (rng.standard_normal((7, cols)) * 5).astype(f32)
        if cols > 1:
            X[0, 0], X[0, 1] = 80.0, -80.0
            X[1, :] = -80.0
        cases["x%d" % j] = X
        cases["soft%d" % j] = ref.softmax_per_row(X)
        cases["logsoft%d" % j] = ref.log_softmax_per_row(X)
    cases["n"] = np.array(6)
    save("softmax", **cases)

    # ---- a3 CopyRows with -1; a4 Splice with clamping
    src = rng.standard_normal((40, 33)).astype(f32)
    idx = rng.integers(-1, 40, 55).astype(np.int32)
    idx[:3] = [-1, 0, 39]
    save("copy_rows", src=src, idx=idx, out=ref.copy_rows(src, idx))
    offs = np.array([-5, -2, 0, 1, 5], np.int32)
    save("splice", src=src, offsets=offs, out=ref.splice(src, offs))

    # ---- a5 GroupPnorm p in {1, 2, 3, 0.5}, and overflow rescue for p=3
    X = rng.standard_normal((19, 60)).astype(f32)
    Xbig = X.copy()
    Xbig[0, :10] *= 1e20  # x^2 overflows float -> +inf out of the p == 2 branch
    cases = dict(x=X, xbig=Xbig)
    for p in (1.0, 2.0, 3.0, 0.5):
        cases["p%g" % p] = ref.group_pnorm(X, 10, p)
    # (p = 3 on such input trips KALDI_ASSERT(tmp != HUGE_VAL), kaldi-vector.cc:533,
    # before the rescue branch :535-543 can run, so no fixture exists for it.)
    cases["big_p2"] = ref.group_pnorm(Xbig, 10, 2.0)
    save("group_pnorm", **cases)

    # ---- a6 Normalize incl. all-zero row; AddDiagMat2; MulRowsVec / MulColsVec
    X = rng.standard_normal((23, 35)).astype(f32)
    X[4, :] = 0.0
    X[5, :] *= 1e-12
    v = rng.standard_normal(23).astype(f32)
    s = rng.standard_normal(35).astype(f32)
    save("normalize", x=X, out=ref.normalize(X), v=v,
         diag2=ref.add_diag_mat2(0.7, X, 0.3, v),
         mul_rows=ref.mul_rows_vec(X, v), mul_cols=ref.mul_cols_vec(X, s), s=s)

    # ---- a7 element-wise + SumColumnRanges + Lookup
    X = (rng.standard_normal((17, 40)) * 3).astype(f32)
    P = np.abs(X) + f32(1e-3)
    b = rng.standard_normal(40).astype(f32)
    sizes = np.array([3, 1, 7, 2, 10, 5, 4, 8], np.int32)
    ends = np.cumsum(sizes)
    ranges = np.stack([ends - sizes, ends], 1).astype(np.int32).ravel()
    pairs = np.stack([rng.integers(0, 17, 29), rng.integers(0, 40, 29)], 1).astype(np.int32).ravel()
    save("elementwise", x=X, p=P, b=b,
         copy_rows_from_vec=ref.copy_rows_from_vec(5, b),
         add_vec_to_rows=ref.add_vec_to_rows(-1.0, b, 1.0, X),
         add_vec_to_rows_beta=ref.add_vec_to_rows(0.5, b, 0.25, X),
         floor=ref.apply_floor(X, 0.5), log=ref.apply_log(P), exp=ref.apply_exp(X),
         pow_half=ref.apply_pow(P, 0.5), pow_2=ref.apply_pow(X, 2.0), pow_m05=ref.apply_pow(P, -0.5),
         scale=ref.scale(X, 0.1), ranges=ranges, sum_ranges=ref.sum_column_ranges(X, ranges),
         pairs=pairs, lookup=ref.matrix_lookup(X, pairs))

    # ---- a8 nnet2 forward: random p-norm net (with mix-up SumGroup and const part)
    net, priors = workloads.tiny_net(np.random.default_rng(7), n_pdf=40)
    feats = rng.standard_normal((37, 13)).astype(f32)
    flat = {}
    for ci, comp in enumerate(net):
        for key, val in comp.items():
            if isinstance(val, np.ndarray):
                flat["c%d_%s" % (ci, key)] = val
    save("nnet_tiny", feats=feats, priors=priors,
         out_pad=ref.nnet_forward(net, feats, True), out_nopad=ref.nnet_forward(net, feats, False),
         logprobs=ref.decodable_am_nnet(net, priors, 0.1, feats),
         context=np.array(ref.nnet_context(net), np.int32), **flat)

    # ---- a9 DiagGmm: M in {1, 7, 32}, D in {13, 39, 40}; zero-weight component
    cases = {}
    for gi, (M, D) in enumerate([(1, 13), (7, 39), (32, 40)]):
        w = rng.dirichlet(np.ones(M)).astype(f32)
        if M == 7:
            w[3] = 0.0
            w /= w.sum()
        m = rng.standard_normal((M, D)).astype(f32)
        v = np.exp(rng.standard_normal((M, D)) * 0.5).astype(f32)
        data = rng.standard_normal((11, D)).astype(f32)
        g, mi, iv, bad = ref.ref_diag_gmm_build(w, m, v)
        ll = ref.ref_diag_gmm_loglikes(w, m, v, data)
        per_frame = ref.ref_diag_gmm_loglike_per_frame(w, m, v, data)
        lse_prune5 = np.array([ref.log_sum_exp(ll[t], 5.0) for t in range(len(data))], f32)
        lse = np.array([ref.log_sum_exp(ll[t], -1.0) for t in range(len(data))], f32)
        for k, a in dict(w=w, m=m, v=v, data=data, g=g, mi=mi, iv=iv, bad=np.array(bad), ll=ll,
                         per_frame=per_frame, lse=lse, lse_prune5=lse_prune5).items():
            cases["g%d_%s" % (gi, k)] = a
    cases["n"] = np.array(3)
    save("diag_gmm", **cases)


def make_features():
    """Feature front-end vectors from the reference's own feat/ + transform/cmvn.cc
    (oracle/_ref): MFCC of a synthetic 2 s waveform with the two recipe configurations,
    deltas, CMVN statistics and normalised features."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from test_feature_oracle import MFCC_CONFIGS, MFCC_OPTION_CASES, wave
    ref = binding.OracleLib("ref")
    w = wave(7)
    out = dict(wave=w)
    for name, cfg in MFCC_CONFIGS.items():
        out["mfcc_" + name] = ref.mfcc_compute(w, **cfg)
    for name, kw in MFCC_OPTION_CASES.items():
        out["mfcc_opt_" + name] = ref.mfcc_compute(w, **kw)
    x = out["mfcc_mfcc13"]
    out["deltas"] = ref.compute_deltas(x, 2, 2)
    out["cmvn_stats"] = ref.acc_cmvn_stats(x)
    out["cmn"] = ref.apply_cmvn(out["cmvn_stats"], False, x)
    out["cmvn"] = ref.apply_cmvn(out["cmvn_stats"], True, x)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "features.npz"), **out)
    print("features.npz:", {k: v.shape for k, v in out.items()})


TOPO_TEXT = """<Topology>
<TopologyEntry>
<ForPhones> 2 3 5 </ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.75 <Transition> 1 0.25 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
<State> 2 <PdfClass> 2 <Transition> 2 0.6 <Transition> 3 0.4 </State>
<State> 3 </State>
</TopologyEntry>
<TopologyEntry>
<ForPhones> 1 </ForPhones>
<State> 0 <PdfClass> 0 <Transition> 0 0.25 <Transition> 1 0.25 <Transition> 2 0.5 </State>
<State> 1 <PdfClass> 1 <Transition> 1 0.5 <Transition> 2 0.5 </State>
<State> 2 </State>
</TopologyEntry>
</Topology>
"""


def kaldi_io_net(rng):
    """A small net with every component type the forward path reads."""
    net, priors = workloads.make_pnorm_net(rng, feat_dim=6, splice=1, const_dim=2, pnorm_in=20, pnorm_out=4,
                                           n_hidden=1, n_mix=12, n_pdf=5)
    k = next(i for i, c in enumerate(net) if c["type"] == "normalize") + 1
    net.insert(k, dict(type="fixed_scale", input_dim=4, output_dim=4, bias=rng.uniform(0.5, 2, 4).astype(np.float32)))
    net.insert(k + 1, dict(type="fixed_bias", input_dim=4, output_dim=4, bias=rng.standard_normal(4).astype(np.float32)))
    return net, priors


def make_kaldi_io():
    """Files written by the REFERENCE's own Write functions (oracle/_ref: Matrix / Vector /
    CompressedMatrix / WriteIntegerVector / TableWriter / AmNnet / HmmTopology) + the arrays
    that went in, for tests/test_kaldi_io.py.  Data only."""
    import ctypes as C
    out_dir = os.path.join(HERE, "kaldi_io")
    os.makedirs(out_dir, exist_ok=True)
    ref = binding.OracleLib("ref").lib
    fp, ip = binding._fp, binding._ip
    rng = np.random.default_rng(11)
    M = (rng.standard_normal((5, 4)) * 10).astype(np.float32)
    M[0, 0], M[1, 1], M[2, 2] = np.inf, -np.inf, 0.0
    v = rng.standard_normal(7).astype(np.float32)
    big = (rng.standard_normal((20, 7)) * 3).astype(np.float32)      # format 1 ("CM")
    small = (rng.standard_normal((5, 7)) * 3).astype(np.float32)     # <= 8 rows: format 2 ("CM2")
    iv = rng.integers(-5, 1000, 9).astype(np.int32)
    cwd = os.getcwd()
    os.chdir(out_dir)   # the script files then hold relative paths
    try:
        def b(x):
            return x.encode()
        assert ref.ref_write_matrix(b("mat_f.bin"), fp(M), 5, 4, 1, 0) == 0
        assert ref.ref_write_matrix(b("mat_f.txt"), fp(M), 5, 4, 0, 0) == 0
        assert ref.ref_write_matrix(b("mat_d.bin"), fp(M), 5, 4, 1, 1) == 0
        assert ref.ref_write_matrix(b("mat_empty.txt"), fp(M), 0, 0, 0, 0) == 0
        assert ref.ref_write_vector(b("vec.bin"), fp(v), 7, 1) == 0
        assert ref.ref_write_vector(b("vec.txt"), fp(v), 7, 0) == 0
        assert ref.ref_write_compressed_matrix(b("cm.bin"), fp(big), 20, 7) == 0
        assert ref.ref_write_compressed_matrix(b("cm2.bin"), fp(small), 5, 7) == 0
        big_rt, small_rt = np.zeros_like(big), np.zeros_like(small)
        ref.ref_compress_roundtrip(fp(big), 20, 7, fp(big_rt))
        ref.ref_compress_roundtrip(fp(small), 5, 7, fp(small_rt))
        assert ref.ref_write_int_vector(b("ivec.bin"), ip(iv), 9, 1) == 0
        assert ref.ref_write_int_vector(b("ivec.txt"), ip(iv), 9, 0) == 0
        feats = rng.standard_normal((30, 6)).astype(np.float32)
        off = np.asarray([0, 9, 10, 30], np.int32)
        assert ref.ref_write_matrix_table(b("ark,scp:feats.ark,feats.scp"), fp(feats), 6, ip(off), 3, 0) == 0
        assert ref.ref_write_matrix_table(b("ark,t:feats_t.ark"), fp(feats), 6, ip(off), 3, 0) == 0
        assert ref.ref_write_matrix_table(b("ark:feats_cm.ark"), fp(feats), 6, ip(off), 3, 1) == 0
        ali = rng.integers(1, 500, 14).astype(np.int32)
        ali_off = np.asarray([0, 6, 6, 14], np.int32)     # the middle one is empty
        assert ref.ref_write_int_vector_table(b("ark:ali.ark"), ip(ali), ip(ali_off), 3) == 0
        assert ref.ref_write_int_vector_table(b("ark,t:ali_t.ark"), ip(ali), ip(ali_off), 3) == 0
        feats_cm = np.zeros_like(feats)
        for u in range(3):
            a, e = off[u], off[u + 1]
            blk = np.ascontiguousarray(feats[a:e])
            rt = np.zeros_like(blk)
            ref.ref_compress_roundtrip(fp(blk), int(e - a), 6, fp(rt))
            feats_cm[a:e] = rt
        net, priors = kaldi_io_net(np.random.default_rng(12))
        arr, keep = binding.pack_components(net)
        for kind in (0, 1, 2):
            for binary in (1, 0):
                name = "am_nnet_%s_%d" % ("bin" if binary else "txt", kind)
                rc = ref.ref_write_am_nnet(b(name), arr, len(net), fp(priors), len(priors), binary, kind, 1)
                assert rc == 0, (name, rc)
        assert ref.ref_write_am_nnet(b("am_nnet_body_bin"), arr, len(net), fp(priors), len(priors), 1, 2, 0) == 0
        am = workloads.make_am_gmm(np.random.default_rng(13), 5, 17, 6)
        for binary, name in ((1, "am_gmm_body_bin"), (0, "am_gmm_body_txt")):
            assert ref.ref_write_am_diag_gmm(b(name), fp(am["weights"]), fp(am["means"]), fp(am["vars"]),
                                             ip(am["pdf_offsets"]), 5, 6, binary) == 0
        assert ref.ref_write_topology(b("topo.bin"), b(TOPO_TEXT), 1) == 0
        assert ref.ref_write_topology(b("topo.txt"), b(TOPO_TEXT), 0) == 0
        # the online2 front-end's files: final.dubm, final.ie, a wave file
        ie = workloads.make_ivector_extractor(np.random.default_rng(14), base_dim=5, splice=1, feat_dim=4, num_gauss=3,
                                              ivector_dim=6, prior_offset=7.5)
        dp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
        for binary, name in ((1, "final.dubm"), (0, "final_dubm.txt")):
            assert ref.ref_write_diag_gmm_file(b(name), fp(ie["ubm_weights"]), fp(ie["ubm_means"]), fp(ie["ubm_vars"]), 3, 4, binary) == 0
        w_vec = np.log(ie["ubm_weights"].astype(np.float64))
        for binary, name in ((1, "final.ie"), (0, "final_ie.txt")):
            assert ref.ref_write_ivector_extractor(b(name), dp(np.zeros(0)), 0, 0, dp(w_vec), 3, 4, 6, dp(ie["M"]), dp(ie["Sigma_inv"]),
                                                   C.c_double(ie["prior_offset"]), binary) == 0
        wav = (np.random.default_rng(15).standard_normal(500) * 3000).astype(np.float32)
        assert ref.ref_write_wave(b("utt.wav"), fp(wav), 500, C.c_float(16000.0)) == 0
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(out_dir, "expected.npz"), M=M, v=v, big=big, small=small, big_rt=big_rt,
                        small_rt=small_rt, iv=iv, feats=feats, off=off, feats_cm=feats_cm, ali=ali, ali_off=ali_off, wav=wav,
                        ie_w_vec=w_vec)
    print("kaldi_io:", sorted(os.listdir(out_dir)))


def make_ivector():
    """Reference outputs for the iVector front-end pieces that compile from the reference's own
    sources (feat/online-feature.cc: OnlineCmvn, OnlineSpliceFrames, OnlineTransform;
    matrix/optimization.cc: LinearCgd) on the cases of tests/test_ivector_oracle.py."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_ivector_oracle as T
    m, m2, X = T.cases()
    lda_a, lda_norm_a, cmvn_a = T.ref_chain(m, X)
    lda_b, lda_norm_b, cmvn_b = T.ref_chain(m2, X)
    rng = np.random.default_rng(3)
    S = 12
    Q = rng.standard_normal((S, S))
    A = Q @ Q.T + np.eye(S)
    b = rng.standard_normal(S)
    x0 = np.zeros(S)
    x0[0] = 2.0
    x, k = T.ref_cgd(A, b, x0, 15)
    np.savez_compressed(os.path.join(HERE, "ivector.npz"), lda_a=lda_a, lda_norm_a=lda_norm_a, cmvn_a=cmvn_a, lda_b=lda_b,
                        lda_norm_b=lda_norm_b, cmvn_b=cmvn_b, cg_A=A, cg_b=b, cg_x0=x0, cg_x=x, cg_k=k)
    print("ivector: written")


if __name__ == "__main__":
    if "--ivector" in sys.argv:
        make_ivector()
    elif "--features" in sys.argv:
        make_features()
    elif "--kaldi-io" in sys.argv:
        make_kaldi_io()
    else:
        main()
        make_features()
        make_kaldi_io()
        make_ivector()
