"""Shared helpers of the parity tests."""
import numpy as np

from graphblas_amd.binding import Type, NP_TYPES

# The reference's own comparison rule (Test/GB_spec_compare.m:24-58): pattern and type
# identical; values exact for integer/bool; within 64*eps(type) relative for floating point.
FP_RTOL = {Type.FP32: 64 * np.finfo(np.float32).eps, Type.FP64: 64 * np.finfo(np.float64).eps}


def assert_same(got, want, exact=None, check_values=True, what=""):
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
        rtol = FP_RTOL[got.type]
        np.testing.assert_allclose(gv.astype(np.float64), wv.astype(np.float64), rtol=rtol, atol=0, err_msg=what)


def assert_sums_within_bound(got, want, terms, what=""):
    """Two floating-point evaluations, in different orders, of sums of POSITIVE products: same pattern, and
    every value pair within the a-priori bound.  A sum of m products evaluated in any order is within
    (m + 1) * u * exact of the exact sum (u the unit roundoff: one rounding per product, at most m - 1 roundings
    on the path of any term through the additions, first-order), so two such evaluations differ by at most
    2 * (m + 1) * u * exact; `terms` gives m per entry in the (row, col) order of Matrix.coo()."""
    assert got.type == want.type and got.type in FP_RTOL, what
    gr, gc, gv = got.coo()
    wr, wc, wv = want.coo()
    assert np.array_equal(gr, wr) and np.array_equal(gc, wc), f"{what}: pattern differs"
    terms = np.asarray(terms, np.float64)
    assert len(terms) == len(gv), f"{what}: {len(terms)} term counts for {len(gv)} entries"
    u = np.finfo(NP_TYPES[got.type]).eps / 2
    a, b = gv.astype(np.float64), wv.astype(np.float64)
    assert np.all(b > 0), f"{what}: the bound is stated for positive sums"
    ratio = np.abs(a - b) / (2 * (terms + 1) * u * b)
    worst = int(np.argmax(ratio)) if len(ratio) else 0
    assert len(ratio) == 0 or ratio[worst] <= 1.0, \
        f"{what}: ({gr[worst]},{gc[worst]}) {a[worst]!r} vs {b[worst]!r}: {ratio[worst]:.2f} x the bound for {int(terms[worst])} terms"


def dense_of(M):
    """(present, value) nrows-by-ncols arrays of a host Matrix."""
    r, c, v = M.coo()
    present = np.zeros((M.nrows, M.ncols), bool)
    value = np.zeros((M.nrows, M.ncols), NP_TYPES[M.type])
    present[r, c] = True
    value[r, c] = v
    return present, value


def assert_dense_same(got, present, value, exact=None, what=""):
    gp, gv = dense_of(got)
    assert np.array_equal(gp, present), f"{what}: pattern differs ({gp.sum()} vs {present.sum()} entries)"
    if exact is None:
        exact = got.type not in FP_RTOL
    a, b = gv[present], value[present]
    if exact:
        assert np.array_equal(a, b), f"{what}: values differ"
    else:
        np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=FP_RTOL[got.type], atol=0, err_msg=what)
