"""Shared helpers of the parity tests."""
import numpy as np

from graphblas_amd.binding import Type, NP_TYPES

# The reference's own comparison rule (Test/GB_spec_compare.m:24-58): pattern and type
# identical; values exact for integer/bool; within 64*eps(type) relative for floating point.
FP_RTOL = {Type.FP32: 64 * np.finfo(np.float32).eps, Type.FP64: 64 * np.finfo(np.float64).eps}


def assert_same(got, want, exact=None, check_values=True, rtol_scale=1.0, what=""):
    """Compare two host Matrix objects as sets of (row, col, value)."""
    assert got.type == want.type, f"{what}: type {got.type} != {want.type}"
    assert (got.nrows, got.ncols) == (want.nrows, want.ncols), f"{what}: shape"
    gr, gc, gv = got.coo()
    wr, wc, wv = want.coo()
    assert len(gr) == len(wr), f"{what}: nvals {len(gr)} != {len(wr)}"
    assert np.array_equal(gr, wr) and np.array_equal(gc, wc), f"{what}: pattern differs"
    if not check_values:
        return
    if exact is None:
        exact = got.type not in FP_RTOL
    if exact:
        if got.type in FP_RTOL:
            same = (gv == wv) | (np.isnan(gv) & np.isnan(wv))
        else:
            same = gv == wv
        bad = np.nonzero(~same)[0]
        assert len(bad) == 0, f"{what}: {len(bad)} values differ, first at ({gr[bad[0]]},{gc[bad[0]]}): {gv[bad[0]]} vs {wv[bad[0]]}"
    else:
        rtol = FP_RTOL[got.type] * rtol_scale
        np.testing.assert_allclose(gv.astype(np.float64), wv.astype(np.float64), rtol=rtol, atol=0, err_msg=what)


def dense_of(M):
    """(present, value) nrows-by-ncols arrays of a host Matrix."""
    r, c, v = M.coo()
    present = np.zeros((M.nrows, M.ncols), bool)
    value = np.zeros((M.nrows, M.ncols), NP_TYPES[M.type])
    present[r, c] = True
    value[r, c] = v
    return present, value


def assert_dense_same(got, present, value, exact=None, rtol_scale=1.0, what=""):
    gp, gv = dense_of(got)
    assert np.array_equal(gp, present), f"{what}: pattern differs ({gp.sum()} vs {present.sum()} entries)"
    if exact is None:
        exact = got.type not in FP_RTOL
    a, b = gv[present], value[present]
    if exact:
        assert np.array_equal(a, b), f"{what}: values differ"
    else:
        np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=FP_RTOL[got.type] * rtol_scale, atol=0, err_msg=what)
