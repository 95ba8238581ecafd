"""GPU: the reference's lower C-ABI seam (include/cu_kernels_ansi_hip.h) — the
cu-kernels-ansi.h launcher names called the way cu-matrix.cc calls them (geometry
arguments by value, MatrixDim by value), checked against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class MatrixDim(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("stride", C.c_int32)]


G, B = Dim3(1, 1, 1), Dim3(16, 16, 1)   # ignored by the library
vp, f32, i32 = C.c_void_p, C.c_float, C.c_int


@pytest.fixture(scope="module")
def seam(api):
    import torch
    lib = C.CDLL(api.capi.LIB_PATH)
    sig = {
        "cudaF_softmax_reduce": [C.c_size_t, C.c_size_t, vp, vp, MatrixDim, i32],
        "cudaF_log_softmax_reduce": [C.c_size_t, C.c_size_t, vp, vp, MatrixDim, i32],
        "cudaF_copy_rows": [Dim3, Dim3, vp, vp, vp, MatrixDim, i32],
        "cudaF_splice": [Dim3, Dim3, vp, vp, vp, MatrixDim, MatrixDim],
        "cudaF_group_pnorm": [Dim3, Dim3, vp, vp, MatrixDim, i32, i32, f32],
        "cudaF_add_diag_mat_mat": [i32, i32, f32, vp, i32, vp, i32, i32, i32, vp, i32, i32, i32, f32],
        "cudaF_mul_cols_vec": [Dim3, Dim3, vp, vp, MatrixDim],
        "cudaF_mul_rows_vec": [Dim3, Dim3, vp, vp, MatrixDim],
        "cudaF_copy_rows_from_vec": [Dim3, Dim3, vp, MatrixDim, vp],
        "cudaF_add_vec_to_rows": [Dim3, Dim3, f32, vp, f32, vp, MatrixDim],
        "cudaF_apply_exp": [Dim3, Dim3, vp, MatrixDim],
        "cudaF_apply_pow": [Dim3, Dim3, vp, f32, MatrixDim],
        "cudaF_apply_floor": [Dim3, Dim3, vp, f32, MatrixDim],
        "cudaF_scale": [Dim3, Dim3, vp, f32, MatrixDim],
        "cudaF_apply_log": [Dim3, Dim3, vp, MatrixDim],
        "cudaF_sum_column_ranges": [Dim3, Dim3, vp, MatrixDim, vp, MatrixDim, vp],
        "cudaF_matrix_lookup": [Dim3, Dim3, vp, MatrixDim, vp, i32, vp],
        "cudaF_comp_obj_deriv": [Dim3, Dim3, vp, i32, vp, MatrixDim, vp, MatrixDim, vp],
        "cublasSgemm": [C.c_char, C.c_char, i32, i32, i32, f32, vp, i32, vp, i32, f32, vp, i32],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, None
    lib.kh_cuda_seam_status.restype = C.c_int

    class S:
        pass
    s = S()
    s.lib, s.torch, s.api = lib, torch, api

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s.dev = dev
    s.ptr = lambda t: vp(t.data_ptr())
    s.dim = lambda t: MatrixDim(t.shape[0], t.shape[1], t.stride(0))

    def host(t):
        api.synchronize()
        assert lib.kh_cuda_seam_status() == 0, api.lib().kh_last_error()
        return t.cpu().numpy()
    s.host = host
    return s


def test_softmax_pnorm_copy_rows_splice(seam, oracle, rng):
    s = seam
    X = (rng.standard_normal((37, 300)) * 3).astype(np.float32)
    x = s.dev(X)
    y = s.torch.empty_like(x)
    s.lib.cudaF_softmax_reduce(1, 256, s.ptr(y), s.ptr(x), s.dim(y), x.stride(0))
    cases.close(s.host(y), oracle.softmax_per_row(X), rtol=1e-5, atol=1e-30)
    s.lib.cudaF_log_softmax_reduce(1, 256, s.ptr(y), s.ptr(x), s.dim(y), x.stride(0))
    cases.close(s.host(y), oracle.log_softmax_per_row(X), rtol=1e-5, atol=5e-5)
    yp = s.torch.empty((37, 30), device="cuda")
    s.lib.cudaF_group_pnorm(G, B, s.ptr(yp), s.ptr(x), s.dim(yp), x.stride(0), 10, 2.0)
    cases.close(s.host(yp), oracle.group_pnorm(X, 10, 2.0))
    idx = rng.integers(-1, 37, 50).astype(np.int32)
    yc = s.torch.empty((50, 300), device="cuda")
    s.lib.cudaF_copy_rows(G, B, s.ptr(yc), s.ptr(x), s.ptr(s.dev(idx)), s.dim(yc), x.stride(0))
    cases.exact(s.host(yc), oracle.copy_rows(X, idx))
    off = np.asarray([-2, 0, 3], np.int32)
    ys = s.torch.empty((37, 900), device="cuda")
    s.lib.cudaF_splice(G, B, s.ptr(ys), s.ptr(x), s.ptr(s.dev(off)), s.dim(ys), s.dim(x))
    cases.exact(s.host(ys), oracle.splice(X, off))


def test_elementwise_and_broadcasts(seam, rng):
    s = seam
    X = (np.abs(rng.standard_normal((23, 70))) + 0.1).astype(np.float32)
    v_r = rng.standard_normal(23).astype(np.float32)
    v_c = rng.standard_normal(70).astype(np.float32)
    m = s.dev(X); s.lib.cudaF_apply_log(G, B, s.ptr(m), s.dim(m))
    cases.close(s.host(m), np.log(X), rtol=2e-6, atol=1e-6)
    m = s.dev(X); s.lib.cudaF_apply_exp(G, B, s.ptr(m), s.dim(m))
    cases.close(s.host(m), np.exp(X), rtol=2e-6)
    m = s.dev(X); s.lib.cudaF_apply_pow(G, B, s.ptr(m), 0.5, s.dim(m))
    cases.close(s.host(m), np.sqrt(X), rtol=2e-6)
    m = s.dev(X); s.lib.cudaF_apply_floor(G, B, s.ptr(m), 0.7, s.dim(m))
    cases.exact(s.host(m), np.maximum(X, np.float32(0.7)))
    m = s.dev(X); s.lib.cudaF_scale(G, B, s.ptr(m), 0.25, s.dim(m))
    cases.exact(s.host(m), X * np.float32(0.25))
    m = s.dev(X); s.lib.cudaF_mul_rows_vec(G, B, s.ptr(m), s.ptr(s.dev(v_r)), s.dim(m))
    cases.exact(s.host(m), X * v_r[:, None])
    m = s.dev(X); s.lib.cudaF_mul_cols_vec(G, B, s.ptr(m), s.ptr(s.dev(v_c)), s.dim(m))
    cases.exact(s.host(m), X * v_c[None, :])
    m = s.dev(X); s.lib.cudaF_copy_rows_from_vec(G, B, s.ptr(m), s.dim(m), s.ptr(s.dev(v_c)))
    cases.exact(s.host(m), np.broadcast_to(v_c, X.shape))
    m = s.dev(X); s.lib.cudaF_add_vec_to_rows(G, B, 2.0, s.ptr(s.dev(v_c)), 1.0, s.ptr(m), s.dim(m))
    cases.exact(s.host(m), np.float32(2.0) * v_c[None, :] + X)


def test_ranges_lookup_objf_diag(seam, oracle, rng):
    s = seam
    X = rng.standard_normal((19, 120)).astype(np.float32)
    x = s.dev(X)
    sizes = 1 + rng.multinomial(120 - 40, np.full(40, 1 / 40))
    ends = np.cumsum(sizes)
    ranges = np.stack([ends - sizes, ends], 1).astype(np.int32)
    y = s.torch.empty((19, 40), device="cuda")
    s.lib.cudaF_sum_column_ranges(G, B, s.ptr(y), s.dim(y), s.ptr(x), s.dim(x), s.ptr(s.dev(ranges)))
    cases.close(s.host(y), oracle.sum_column_ranges(X, ranges.ravel()), atol=1e-5)
    pairs = np.stack([rng.integers(0, 19, 33), rng.integers(0, 120, 33)], 1).astype(np.int32)
    out = s.torch.empty(33, device="cuda")
    s.lib.cudaF_matrix_lookup(G, B, s.ptr(x), s.dim(x), s.ptr(s.dev(pairs)), 33, s.ptr(out))
    cases.exact(s.host(out), X[pairs[:, 0], pairs[:, 1]])
    # CompObjfAndDeriv: MatrixElement<float> array on the device, t = 2 floats on the device
    P = (np.abs(rng.standard_normal((19, 120))) + 0.05).astype(np.float32)
    el = np.zeros(25, dtype=[("row", np.int32), ("column", np.int32), ("weight", np.float32)])
    el["row"], el["column"] = rng.permutation(19 * 120)[:25] // 120, rng.permutation(120)[:25]
    el["weight"] = rng.uniform(0.2, 1.5, 25).astype(np.float32)
    d_el = s.torch.from_numpy(el.view(np.uint8).copy()).cuda()
    p, deriv, t = s.dev(P), s.torch.zeros((19, 120), device="cuda"), s.torch.empty(2, device="cuda")
    s.lib.cudaF_comp_obj_deriv(G, B, s.ptr(d_el), 25, s.ptr(p), s.dim(p), s.ptr(deriv), s.dim(deriv), s.ptr(t))
    want_d = np.zeros_like(P)
    np.add.at(want_d, (el["row"], el["column"]), el["weight"] / P[el["row"], el["column"]])
    cases.close(s.host(deriv), want_d, rtol=1e-6)
    tt = s.host(t)
    assert abs(tt[0] - float(np.sum(el["weight"].astype(np.float64) * np.log(P[el["row"], el["column"]].astype(np.float64))))) < 1e-4
    assert abs(tt[1] - float(el["weight"].astype(np.float64).sum())) < 1e-5
    # AddDiagMat2 (N = M^T) and the general strided AddDiagMatMat
    M = rng.standard_normal((19, 120)).astype(np.float32)
    N = rng.standard_normal((120, 19)).astype(np.float32)
    m, n = s.dev(M), s.dev(N)
    v = s.dev(np.ones(19, np.float32))
    s.lib.cudaF_add_diag_mat_mat(1, 256, 0.5, s.ptr(v), 19, s.ptr(m), 120, 120, 1, s.ptr(m), 1, 120, 1, 2.0)
    cases.close(s.host(v), 2.0 + 0.5 * (M.astype(np.float64) ** 2).sum(1), rtol=1e-5)
    v = s.dev(np.ones(19, np.float32))
    s.lib.cudaF_add_diag_mat_mat(1, 256, 1.0, s.ptr(v), 19, s.ptr(m), 120, 120, 1, s.ptr(n), 19, 1, 1, 0.0)
    cases.close(s.host(v), np.einsum("ij,ji->i", M.astype(np.float64), N.astype(np.float64)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_cublas_sgemm_as_cu_matrix_calls_it(seam, oracle, rng, tA, tB):
    """CuMatrixBase::AddMatMat (cu-matrix.cc:947-982) hands the ROW-major operands to the
    column-major BLAS swapped: cublas_gemm(transB, transA, m = C.cols, n = C.rows, k, alpha,
    B, B.stride, A, A.stride, beta, C, C.stride)."""
    s = seam
    mm, nn, kk = 45, 70, 33
    A = rng.standard_normal((kk, mm) if tA else (mm, kk)).astype(np.float32)
    Bm = rng.standard_normal((nn, kk) if tB else (kk, nn)).astype(np.float32)
    C0 = rng.standard_normal((mm, nn)).astype(np.float32)
    a, b, c = s.dev(A), s.dev(Bm), s.dev(C0)
    s.lib.cublasSgemm(b"T" if tB else b"N", b"T" if tA else b"N", nn, mm, kk, 0.5, s.ptr(b), b.stride(0),
                      s.ptr(a), a.stride(0), 0.25, s.ptr(c), c.stride(0))
    cases.exact(s.host(c), oracle.add_mat_mat(0.5, A, tA, Bm, tB, 0.25, C0))


def test_errors_are_recorded(seam):
    s = seam
    s.lib.cudaF_apply_log(G, B, None, MatrixDim(3, 3, 2))   # stride < cols
    assert s.lib.kh_cuda_seam_status() != 0
    assert s.lib.kh_cuda_seam_status() == 0
