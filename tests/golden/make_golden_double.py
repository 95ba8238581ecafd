"""Golden vectors of the <double> instantiation from the REFERENCE's own CPU code (oracle/_ref, ref_driver.cc:ref_op_d
over CuMatrix<double> / CuVector<double>; cu-matrix.cc:2415-2418).  Runs only in the build container (needs
/root/reference).  Output: tests/golden/double_ops.npz (inputs + the reference's outputs: data only).

    python tests/golden/make_golden_double.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


def ref_op(lib, op, alpha=0.0, beta=0.0, A=None, tA=0, B=None, tB=0, idx=None, Cm=None):
    def d(a):
        return (None, 0, 0) if a is None else (a.ctypes.data_as(dp), a.shape[0] if a.ndim == 2 else 1, a.shape[-1])
    A = None if A is None else np.ascontiguousarray(A, np.float64)
    B = None if B is None else np.ascontiguousarray(B, np.float64)
    Cm = np.ascontiguousarray(Cm, np.float64).copy()
    idx = None if idx is None else np.ascontiguousarray(idx, np.int32)
    (ap, ar, ac), (bp, br, bc), (cp, cr, cc) = d(A), d(B), d(Cm)
    if Cm.ndim == 1:
        cr, cc = Cm.shape[0], 1
    rc = lib.ref_op_d(int(op), C.c_double(alpha), C.c_double(beta), ap, ar, ac, int(tA), bp, br, bc, int(tB),
                      None if idx is None else idx.ctypes.data_as(ip), 0 if idx is None else idx.size, cp, cr, cc)
    assert rc == 0, (op, rc)
    return Cm


def main():
    binding.build(ref=True)
    lib = C.CDLL(binding.REF_SO)
    rng = np.random.default_rng(20261003)
    out = {}
    # 0 AddMatMat: NN/NT/TN/TT, alpha/beta, odd sizes
    i = 0
    for (m, n, k) in [(37, 23, 50), (1, 7, 3), (65, 129, 17), (70, 40, 33)]:
        for tA in (0, 1):
            for tB in (0, 1):
                alpha = [1.0, 0.5, -2.0, 0.25][i % 4]
                beta = [0.0, 1.0, 0.5, 0.0][(i // 2) % 4]
                A = rng.standard_normal((k, m) if tA else (m, k))
                B = rng.standard_normal((n, k) if tB else (k, n))
                Cm = rng.standard_normal((m, n))
                out["gemm%d_A" % i], out["gemm%d_B" % i], out["gemm%d_C" % i] = A, B, Cm
                out["gemm%d_par" % i] = np.array([alpha, beta, tA, tB])
                out["gemm%d_out" % i] = ref_op(lib, 0, alpha, beta, A, tA, B, tB, None, Cm)
                i += 1
    out["n_gemm"] = np.array([i])
    # softmax / log-softmax: cols 1, 10, 257, 3000; values randn * 5 and extremes
    for j, cols in enumerate((1, 10, 257, 1200)):
        x = rng.standard_normal((5, cols)) * 5.0
        x[0, 0] = 700.0
        x[1, -1] = -700.0
        out["sm%d_x" % j] = x
        out["sm%d_y" % j] = ref_op(lib, 1, A=x, Cm=np.zeros_like(x))
        out["sm%d_ly" % j] = ref_op(lib, 2, A=x, Cm=np.zeros_like(x))
    # CopyRows with -1, Splice with clamping
    src = rng.standard_normal((19, 13))
    idx = rng.integers(-1, 19, 31).astype(np.int32)
    out["cr_src"], out["cr_idx"] = src, idx
    out["cr_out"] = ref_op(lib, 3, A=src, idx=idx, Cm=rng.standard_normal((31, 13)))
    off = np.array([-5, -1, 0, 2, 5], np.int32)
    out["sp_src"], out["sp_off"] = src, off
    out["sp_out"] = ref_op(lib, 4, A=src, idx=off, Cm=np.zeros((19, 13 * 5)))
    # GroupPnorm p in {0.5, 1, 2, 3, 0} + the overflow rescue
    g = rng.standard_normal((7, 40))
    g[0, :5] = 0.0
    for j, p in enumerate((0.5, 1.0, 2.0, 3.0, 0.0)):
        out["gp%d_out" % j] = ref_op(lib, 5, alpha=p, A=g, Cm=np.zeros((7, 8)))
    out["gp_src"] = g
    # (an overflowing group is an assertion failure in the reference's <double> Norm, kaldi-vector.cc:533: no vector)
    # vector / element-wise
    M = rng.standard_normal((11, 29))
    v_r, v_c = rng.standard_normal(11), rng.standard_normal(29)
    out["ew_M"], out["ew_vr"], out["ew_vc"] = M, v_r, v_c
    out["ew_diag"] = ref_op(lib, 6, alpha=0.5, beta=2.0, A=M, Cm=v_r)
    out["ew_diag0"] = ref_op(lib, 6, alpha=1.0 / 29, beta=0.0, A=M, Cm=np.zeros(11))
    out["ew_mulrows"] = ref_op(lib, 7, B=v_r, Cm=M)
    out["ew_mulcols"] = ref_op(lib, 8, B=v_c, Cm=M)
    out["ew_rowsfromvec"] = ref_op(lib, 9, B=v_c, Cm=M)
    out["ew_addvec"] = ref_op(lib, 10, alpha=0.7, beta=0.3, B=v_c, Cm=M)
    out["ew_floor"] = ref_op(lib, 11, alpha=-0.2, Cm=M)
    P = np.abs(M) + 0.01
    out["ew_P"] = P
    out["ew_log"] = ref_op(lib, 12, Cm=P)
    out["ew_exp"] = ref_op(lib, 13, Cm=M)
    for j, p in enumerate((1.0, 2.0, 0.5, -0.5, 3.3)):
        out["ew_pow%d" % j] = ref_op(lib, 14, alpha=p, Cm=P)
    out["ew_scale"] = ref_op(lib, 15, alpha=-1.5, Cm=M)
    ranges = np.array([0, 3, 3, 3, 5, 29, 10, 11], np.int32)
    out["ew_ranges"] = ranges
    out["ew_sumranges"] = ref_op(lib, 16, A=M, idx=ranges, Cm=np.zeros((11, 4)))
    pairs = np.stack([rng.integers(0, 11, 17), rng.integers(0, 29, 17)], axis=1).astype(np.int32).ravel()
    out["ew_pairs"] = pairs
    out["ew_lookup"] = ref_op(lib, 17, A=M, idx=pairs, Cm=np.zeros(17))
    path = os.path.join(HERE, "double_ops.npz")
    np.savez_compressed(path, **out)
    print("double_ops.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024.0, len(out)))


if __name__ == "__main__":
    main()
