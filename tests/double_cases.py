"""The <double> cases of tests/golden/double_ops.npz as (name, run(ops) -> result, expected, tolerance) tuples, shared by
the CPU test of the numpy restatement (oracle/double_oracle.py) and the GPU test of csrc/kh_double.hip.
`ops` is a namespace with add_mat_mat(alpha, A, tA, B, tB, beta, C), softmax_per_row(x), log_softmax_per_row(x),
copy_rows(src, idx, rows), splice(src, offsets), group_pnorm(src, group, p), add_diag_mat2(alpha, M, beta, v),
mul_rows_vec(M, v), mul_cols_vec(M, v), copy_rows_from_vec(M, v), add_vec_to_rows(M, alpha, v, beta), apply_floor(M, f),
apply_log(M), apply_exp(M), apply_pow(M, p), scale(M, a), sum_column_ranges(src, ranges, cols), lookup(M, pairs) —
all taking and returning numpy float64 arrays (inputs are never modified)."""
import numpy as np

# dgemm / exp / log / pow of two correct implementations differ in the last bits (summation order, libm vs OCML)
RTOL = 1e-12


def cases(G):
    out = []
    for i in range(int(G["n_gemm"][0])):
        alpha, beta, tA, tB = G["gemm%d_par" % i]
        out.append(("AddMatMat %d" % i, (lambda o, i=i, alpha=alpha, beta=beta, tA=tA, tB=tB: o.add_mat_mat(
            float(alpha), G["gemm%d_A" % i], int(tA), G["gemm%d_B" % i], int(tB), float(beta), G["gemm%d_C" % i])),
            G["gemm%d_out" % i], 1e-11))
    for j in range(4):
        out.append(("softmax %d" % j, (lambda o, j=j: o.softmax_per_row(G["sm%d_x" % j])), G["sm%d_y" % j], RTOL))
        out.append(("log-softmax %d" % j, (lambda o, j=j: o.log_softmax_per_row(G["sm%d_x" % j])), G["sm%d_ly" % j], RTOL))
    out.append(("CopyRows", lambda o: o.copy_rows(G["cr_src"], G["cr_idx"], 31), G["cr_out"], 0.0))
    out.append(("Splice", lambda o: o.splice(G["sp_src"], G["sp_off"]), G["sp_out"], 0.0))
    for j, p in enumerate((0.5, 1.0, 2.0, 3.0, 0.0)):
        out.append(("GroupPnorm p=%g" % p, (lambda o, p=p: o.group_pnorm(G["gp_src"], 5, p)), G["gp%d_out" % j], RTOL))
    M, vr, vc, P = G["ew_M"], G["ew_vr"], G["ew_vc"], G["ew_P"]
    out += [
        ("AddDiagMat2", lambda o: o.add_diag_mat2(0.5, M, 2.0, vr), G["ew_diag"], RTOL),
        ("AddDiagMat2 beta=0", lambda o: o.add_diag_mat2(1.0 / 29, M, 0.0, np.zeros(11)), G["ew_diag0"], RTOL),
        ("MulRowsVec", lambda o: o.mul_rows_vec(M, vr), G["ew_mulrows"], 0.0),
        ("MulColsVec", lambda o: o.mul_cols_vec(M, vc), G["ew_mulcols"], 0.0),
        ("CopyRowsFromVec", lambda o: o.copy_rows_from_vec(M, vc), G["ew_rowsfromvec"], 0.0),
        ("AddVecToRows", lambda o: o.add_vec_to_rows(M, 0.7, vc, 0.3), G["ew_addvec"], 1e-15),
        ("ApplyFloor", lambda o: o.apply_floor(M, -0.2), G["ew_floor"], 0.0),
        ("ApplyLog", lambda o: o.apply_log(P), G["ew_log"], RTOL),
        ("ApplyExp", lambda o: o.apply_exp(M), G["ew_exp"], RTOL),
        ("Scale", lambda o: o.scale(M, -1.5), G["ew_scale"], 0.0),
        ("SumColumnRanges", lambda o: o.sum_column_ranges(M, G["ew_ranges"], 4), G["ew_sumranges"], RTOL),
        ("Lookup", lambda o: o.lookup(M, G["ew_pairs"]), G["ew_lookup"], 0.0),
    ]
    for j, p in enumerate((1.0, 2.0, 0.5, -0.5, 3.3)):
        out.append(("ApplyPow %g" % p, (lambda o, p=p: o.apply_pow(P, p)), G["ew_pow%d" % j], RTOL))
    return out


def check(name, got, want, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if tol == 0.0:
        assert np.array_equal(got, want), name
    else:
        scale = np.maximum(np.abs(want), 1e-300)
        err = np.abs(got - want) / np.maximum(scale, np.abs(want).max() * 1e-3 if want.size else 1.0)
        assert err.max() <= tol, (name, float(err.max()))
