"""The numpy float64 restatement of the <double> CuMatrix primitives (oracle/double_oracle.py) against the golden
vectors the reference's own CuMatrix<double> produced (tests/golden/double_ops.npz, make_golden_double.py)."""
import types

import numpy as np

from conftest import load_golden
from oracle import double_oracle as D
import double_cases as DC


def numpy_ops():
    o = types.SimpleNamespace()
    o.add_mat_mat = D.add_mat_mat
    o.softmax_per_row, o.log_softmax_per_row = D.softmax_per_row, D.log_softmax_per_row
    o.copy_rows, o.splice, o.group_pnorm = D.copy_rows, D.splice, D.group_pnorm
    o.add_diag_mat2 = D.add_diag_mat2
    o.mul_rows_vec = lambda M, v: M * v[:, None]
    o.mul_cols_vec = lambda M, v: M * v[None, :]
    o.copy_rows_from_vec = lambda M, v: np.broadcast_to(v, M.shape).copy()
    o.add_vec_to_rows = lambda M, alpha, v, beta: alpha * v[None, :] + beta * M
    o.apply_floor = lambda M, f: np.maximum(M, f)
    o.apply_log, o.apply_exp, o.apply_pow = np.log, np.exp, D.apply_pow
    o.scale = lambda M, a: M * a
    o.sum_column_ranges, o.lookup = D.sum_column_ranges, D.lookup
    return o


def test_numpy_restatement_matches_the_reference_golden_vectors():
    G = load_golden("double_ops")
    ops = numpy_ops()
    n = 0
    for name, run, want, tol in DC.cases(G):
        DC.check(name, run(ops), want, tol)
        n += 1
    assert n >= 45
