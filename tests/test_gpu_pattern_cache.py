"""What the library remembers about a pattern between calls (32-bit index copy, tile tables, the bin plan of the
scatter): products with a remembered pattern equal the first product (rows that straddle two tiles are combined
with floating-point atomics, so two runs of the SAME kernel already differ in the last bits: the comparison allows
for that and no more), the memory is dropped when the arrays are freed, rewritten (gba_matrix_forget) or belong to
another matrix, and a row-slice view has its own."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, gen

pytestmark = pytest.mark.gpu


def _csr(A):
    return sp.csr_matrix((A.x, A.i, A.p), shape=(A.nrows, A.ncols))


def _mxv(gba, dA, x, n_out):
    y = Matrix.full_vector(np.zeros(n_out, np.float32)).to_device()
    dx = Matrix.full_vector(x).to_device()
    rep = gba.mxv(y, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, dx)
    out = y.to_host().x
    y.free(); dx.free()
    return out, rep


def _vxm(gba, dA, x, n_out):
    y = Matrix.full_vector(np.zeros(n_out, np.float32)).to_device()
    dx = Matrix.full_vector(x).to_device()
    rep = gba.vxm(y, None, "PLUS_FP32", "PLUS_TIMES_FP32", dx, dA)
    out = y.to_host().x
    y.free(); dx.free()
    return out, rep


def test_remembered_pattern_gives_the_same_product(gba):
    A = gen.rmat(17, 16, seed=5, type=Type.FP32)
    n = A.nrows
    x = (np.random.default_rng(0).random(n) + 0.5).astype(np.float32)
    dA = A.to_device()
    first, r1 = _mxv(gba, dA, x, n)
    assert gba.Method.NAMES[r1.method] == "dot4" and r1.index_bytes == 8
    for _ in range(3):
        again, r = _mxv(gba, dA, x, n)
        assert np.allclose(first, again, rtol=2e-6, atol=0)
    assert r.index_bytes == 4                             # the 32-bit copy is in use by now
    want = _csr(A).astype(np.float64) @ x.astype(np.float64)
    assert np.allclose(first, want, rtol=1e-4)
    s_first, q1 = _vxm(gba, dA, x, n)
    assert gba.Method.NAMES[q1.method] == "saxpy4"
    for _ in range(3):
        s_again, q = _vxm(gba, dA, x, n)
        assert np.allclose(s_first, s_again, rtol=2e-6, atol=0)
    assert q.index_bytes == 4
    assert np.allclose(s_first, _csr(A).T.astype(np.float64) @ x.astype(np.float64), rtol=1e-4)
    gba.matrix_forget(dA)
    fresh, r = _mxv(gba, dA, x, n)
    assert r.index_bytes == 8 and np.allclose(first, fresh, rtol=2e-6, atol=0)
    dA.free()


def test_freed_arrays_take_their_memory_with_them(gba):
    """Another matrix of the same size uploaded right after a free is likely to land on the same addresses."""
    A = gen.rmat(16, 16, seed=1, type=Type.FP32)
    B = gen.rmat(16, 16, seed=2, type=Type.FP32)
    n = A.nrows
    x = (np.random.default_rng(3).random(n) + 0.5).astype(np.float32)
    dA = A.to_device()
    for _ in range(3):
        _mxv(gba, dA, x, n); _vxm(gba, dA, x, n)
    dA.free()
    M = B
    if B.nvals != A.nvals:                               # same entry count: cut both to the shorter
        k = min(A.nvals, B.nvals)
        rows = int(np.searchsorted(B.p, k, side="right")) - 1
        M = Matrix(B.vlen, B.vdim, B.type, B.sparsity, B.is_csc, p=np.concatenate([B.p[:rows + 1], np.full(n - rows, B.p[rows])]),
                   i=B.i[:B.p[rows]], x=B.x[:B.p[rows]])
    dB = M.to_device()
    S = _csr(M).astype(np.float64)
    for _ in range(3):
        got, _ = _mxv(gba, dB, x, n)
        assert np.allclose(got, S @ x.astype(np.float64), rtol=1e-4)
        got, _ = _vxm(gba, dB, x, n)
        assert np.allclose(got, S.T @ x.astype(np.float64), rtol=1e-4)
    dB.free()


def test_row_slice_views_have_their_own_memory(gba):
    A = gen.rmat(17, 16, seed=9, type=Type.FP32)
    n = A.nrows
    S = _csr(A).astype(np.float64)
    x = (np.random.default_rng(4).random(n) + 0.5).astype(np.float32)
    dA = A.to_device()
    cuts = [0, n // 8, n // 2, n]
    for _ in range(3):
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            got, _ = _mxv(gba, dA.row_slice(lo, hi), x, hi - lo)
            assert np.allclose(got, S[lo:hi] @ x.astype(np.float64), rtol=1e-4), (lo, hi)
        whole, _ = _mxv(gba, dA, x, n)
        assert np.allclose(whole, S @ x.astype(np.float64), rtol=1e-4)
    dA.free()
