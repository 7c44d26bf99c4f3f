"""Edge cases of the two places where the front end works on ranks instead of indices: vectors of 2^31 indices or more
(saxpy3 over the ranks of A's distinct indices) and a hypersparse inner dimension (renamed to the operand's list of
vectors, Source/Shared/GB_hyper_hash_lookup.h:16 in the reference).  Values are small integers; expected results are
written out by hand."""
import numpy as np
import pytest

from graphblas_amd import Matrix, Type, Sparsity, GrbError, Info

pytestmark = pytest.mark.gpu
BIG = 2**40


def _sparse(vlen, vdim, cols, rows, vals):
    cols, rows, vals = np.asarray(cols, np.int64), np.asarray(rows, np.int64), np.asarray(vals, np.float64)
    order = np.lexsort((rows, cols))
    p = np.zeros(vdim + 1, np.int64)
    np.add.at(p, cols + 1, 1)
    return Matrix(vlen, vdim, Type.FP64, Sparsity.SPARSE, True, p=np.cumsum(p), i=rows[order], x=vals[order])


def _run(gba, A, B, C0, M=None, accum=None, desc=None, sr="PLUS_TIMES_FP64"):
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    dM = M.to_device() if M is not None else None
    try:
        gba.mxm(dC, dM, accum, sr, dA, dB, desc)
        gba.matrix_wait(dC)
        return dC.to_host()
    finally:
        for d in (dA, dB, dC, dM):
            if d is not None:
                d.free()


def _entries(out):
    x = [out.x[0]] * out.nvals if out.iso else np.asarray(out.x).tolist()
    return sorted(zip(np.repeat(np.arange(out.nvec), np.diff(out.p)).tolist(), out.i.tolist(), x))


def _hyper_X():
    """6 x 2^40 by column, columns 10 and 2^40 - 3 listed."""
    return Matrix(6, BIG, Type.FP64, Sparsity.HYPERSPARSE, True, p=np.array([0, 2, 3]), i=np.array([1, 4, 2]),
                  x=np.array([1.0, 2.0, 3.0]), h=np.array([10, BIG - 3]))


def test_long_vectors_of_an_operand_without_entries(gba):
    A = _sparse(BIG, 5, [], [], [])
    B = _sparse(5, 3, [0, 1], [1, 2], [1, 2])
    out = _run(gba, A, B, Matrix.sparse_empty(BIG, 3, Type.FP64, by_row=False))
    assert out.nvals == 0 and out.vlen == BIG


@pytest.mark.parametrize("comp", [False, True])
def test_long_vectors_under_a_mask_without_entries(gba, comp):
    A = _sparse(BIG, 4, [0, 0, 2], [5, BIG - 1, 7], [1, 2, 3])
    B = _sparse(4, 2, [0, 1], [0, 2], [2, 5])
    M0 = Matrix(BIG, 2, Type.BOOL, Sparsity.SPARSE, True, p=np.zeros(3, np.int64), i=np.zeros(0, np.int64), x=np.zeros(0, np.bool_))
    out = _run(gba, A, B, Matrix.sparse_empty(BIG, 2, Type.FP64, by_row=False), M0, desc="C" if comp else None)
    assert _entries(out) == ([(0, 5, 2.0), (0, BIG - 1, 4.0), (1, 7, 15.0)] if comp else [])


def test_no_index_of_the_other_operand_is_listed(gba):
    Y = _sparse(BIG, 2, [0, 1], [11, 12], [1, 1])
    assert _run(gba, _hyper_X(), Y, Matrix.sparse_empty(6, 2, Type.FP64, by_row=False)).nvals == 0


def test_an_iso_valued_operand_over_the_renamed_dimension(gba):
    Y = Matrix(BIG, 2, Type.FP64, Sparsity.SPARSE, True, p=np.array([0, 2, 3]), i=np.array([10, 77, BIG - 3]), x=np.array([5.0]), iso=True)
    out = _run(gba, _hyper_X(), Y, Matrix.sparse_empty(6, 2, Type.FP64, by_row=False))
    assert _entries(out) == [(0, 1, 5.0), (0, 4, 10.0), (1, 2, 15.0)]


def test_inner_dimensions_that_differ_are_reported_as_such(gba):
    Y = _sparse(BIG - 1, 2, [0], [10], [1])
    with pytest.raises(GrbError) as err:
        _run(gba, _hyper_X(), Y, Matrix.sparse_empty(6, 2, Type.FP64, by_row=False))
    assert err.value.info == Info.DIMENSION_MISMATCH


def test_a_multiplier_that_yields_the_inner_index_sees_the_true_index(gba):
    """MIN_FIRSTJ: the value is the inner index itself, so that dimension is not renamed."""
    X = Matrix(6, 1000, Type.FP64, Sparsity.HYPERSPARSE, True, p=np.array([0, 1, 2]), i=np.array([1, 2]), x=np.array([1.0, 1.0]),
               h=np.array([10, 990]))
    Y = _sparse(1000, 1, [0, 0], [10, 990], [1, 1])
    out = _run(gba, X, Y, Matrix.sparse_empty(6, 1, Type.INT64, by_row=False), sr="MIN_FIRSTJ_INT64")
    assert _entries(out) == [(0, 1, 10), (0, 2, 990)]


def test_a_jumbled_operand_with_a_zombie_over_the_renamed_dimension(gba):
    Y = Matrix(BIG, 1, Type.FP64, Sparsity.SPARSE, True, p=np.array([0, 3]), i=np.array([BIG - 3, -12, 10]), x=np.array([2.0, 9.0, 3.0]),
               jumbled=True, nzombies=1)
    out = _run(gba, _hyper_X(), Y, Matrix.sparse_empty(6, 1, Type.FP64, by_row=False))
    assert _entries(out) == [(0, 1, 3.0), (0, 2, 6.0), (0, 4, 6.0)]


@pytest.mark.parametrize("ckind", ["hypersparse", "iso"])
def test_an_accumulator_of_another_type_into_a_hypersparse_or_iso_C(gba, ckind):
    """PLUS_INT32 into an FP64 C (Test/GB_spec_accum.m:45-63): where both have an entry the sum is made in int32 (both
    truncated); entries of C alone keep their fractions, entries of the product alone are taken as they are."""
    A = _sparse(4, 3, [0, 0, 2], [0, 3, 1], [1.5, 2.5, 4.5])
    B = _sparse(3, 5, [1, 1, 4], [0, 2, 2], [2.0, 1.0, 1.0])          # T(:,1) = [3, 4.5, ., 5], T(:,4) = [., 4.5, ., .]
    if ckind == "hypersparse":
        C0 = Matrix(4, 5, Type.FP64, Sparsity.HYPERSPARSE, True, p=np.array([0, 2, 3]), i=np.array([0, 2, 1]),
                    x=np.array([10.75, 20.25, 7.5]), h=np.array([1, 3]))
        want = [(1, 0, 13.0), (1, 1, 4.5), (1, 2, 20.25), (1, 3, 5.0), (3, 1, 7.5), (4, 1, 4.5)]
    else:
        C0 = Matrix(4, 5, Type.FP64, Sparsity.SPARSE, True, p=np.array([0, 0, 2, 2, 3, 3]), i=np.array([0, 2, 1]), x=np.array([10.75]), iso=True)
        want = [(1, 0, 13.0), (1, 1, 4.5), (1, 2, 10.75), (1, 3, 5.0), (3, 1, 10.75), (4, 1, 4.5)]
    out = _run(gba, A, B, C0, accum="PLUS_INT32")
    got = sorted((int(out.h[c]) if out.sparsity == Sparsity.HYPERSPARSE else c, i, x) for c, i, x in _entries(out))
    assert got == want and out.type == Type.FP64


def _hyper_cols(S, cols, vdim, csc, t=Type.FP64):
    """The CSC matrix S (few columns) as a hypersparse matrix of vdim vectors: column q of S is vector cols[q]."""
    S = S.tocsc(); S.sort_indices()
    keep = np.diff(S.indptr) > 0
    S = S[:, keep].tocsc()
    return Matrix(S.shape[0], vdim, t, Sparsity.HYPERSPARSE, csc, p=S.indptr.astype(np.int64), i=S.indices.astype(np.int64),
                  x=S.data.astype(np.float64 if t == Type.FP64 else np.int8), h=np.asarray(cols, np.int64)[keep])


@pytest.mark.parametrize("by_row", [False, True])
@pytest.mark.parametrize("desc,accum,masked", [("", None, False), ("", "PLUS_FP64", False), ("", None, True), ("C", "PLUS_FP64", True),
                                               ("RS", None, True), ("RC", "MIN_FP64", True)])
def test_an_outer_dimension_of_2_40_vectors(gba, by_row, desc, accum, masked):
    """C<M> = accum (C, A*B) where C, the mask and B have 2^40 columns and list a few hundred of them each (or, stored by
    row, C, the mask and A have 2^40 rows): the dimension is renamed to the union of the three lists.  Expected: the
    same call on the host over the union (dense arrays, Test/GB_spec_accum_mask.m), the other vectors staying empty."""
    import scipy.sparse as sp
    rng = np.random.default_rng(321)
    m, kk, nu = 40, 30, 400
    union = np.sort(rng.choice(BIG, nu, replace=False).astype(np.int64))
    def some(density, seed, frac):
        S = sp.random(m, nu, density=density, random_state=np.random.default_rng(seed), format="csc")
        S.data = np.random.default_rng(seed + 1).integers(1, 6, len(S.data)).astype(np.float64)
        drop = np.random.default_rng(seed + 2).random(nu) > frac                       # whole vectors left out of the list
        S = S.tolil(); S[:, drop] = 0; S = S.tocsc(); S.eliminate_zeros()
        return S
    As = sp.random(m, kk, density=0.2, random_state=rng, format="csc"); As.data = rng.integers(1, 4, len(As.data)).astype(np.float64)
    Bs = sp.random(kk, nu, density=0.1, random_state=rng, format="csc"); Bs.data = rng.integers(1, 4, len(Bs.data)).astype(np.float64)
    Bl = Bs.tolil(); Bl[:, rng.random(nu) > 0.5] = 0; Bs = Bl.tocsc(); Bs.eliminate_zeros()
    Cs, Ms = some(0.1, 11, 0.4), some(0.3, 21, 0.6)
    Ms.data[:] = rng.integers(0, 2, len(Ms.data))                                      # a valued mask with zeros in it
    # host: dense over the union
    T = (As @ Bs).toarray(); tin = ((As != 0).astype(int) @ (Bs != 0).astype(int)).toarray() > 0
    c, cin = Cs.toarray(), (Cs != 0).toarray()
    if accum == "PLUS_FP64":
        z, zin = np.where(cin & tin, c + T, np.where(cin, c, T)), cin | tin
    elif accum == "MIN_FP64":
        z, zin = np.where(cin & tin, np.minimum(c, T), np.where(cin, c, T)), cin | tin
    else:
        z, zin = T, tin
    if masked:
        mp = sp.csc_matrix((np.ones(len(Ms.data)), Ms.indices, Ms.indptr), shape=Ms.shape).toarray() > 0
        mk = mp if "S" in desc else (Ms.toarray() != 0)
        if "C" in desc:
            mk = ~mk
    else:
        mk = np.ones((m, nu), bool)
    replace = "R" in desc
    win = np.where(mk, zin, False if replace else cin)
    wv = np.where(mk, z, c)
    # device
    C0 = _hyper_cols(Cs, union, BIG, True); M = _hyper_cols(Ms, union, BIG, True, Type.INT8) if masked else None
    B = _hyper_cols(Bs, union, BIG, True); A = Matrix.from_scipy(As, type=Type.FP64, by_row=False)
    if by_row:
        # the same arrays read by row are the transposes: C' = B' A', with C', M' and B' of 2^40 rows
        for X in (C0, M, B, A):
            if X is not None:
                X.is_csc = False
        A, B = B, A
    out = _run(gba, A, B, C0, M, accum, desc or None)
    assert out.sparsity == Sparsity.HYPERSPARSE and out.vdim == BIG and out.vlen == m
    got = sorted((int(out.h[q]), i, x) for q, i, x in _entries(out))
    r, q = np.nonzero(win)
    want = sorted((int(union[qq]), int(rr), float(wv[rr, qq])) for rr, qq in zip(r, q))
    assert got == want
