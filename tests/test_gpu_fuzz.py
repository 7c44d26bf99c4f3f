"""Randomised differential test of the front end: gba_mxm / gba_mxv against the definition
(oracle.mxm_spec, the restatement of Test/GB_spec_mxm.m) over random shapes, storage formats
(sparse / bitmap / full, by row / by column) of C, Mask, A and B, descriptors, accumulators
and semirings.  Values are small integers, so every comparison is exact.  A call the device
path declines must say so with GrB_NOT_IMPLEMENTED -- any other difference is a bug."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, Sparsity, GrbError, Info
from graphblas_amd.binding import NP_TYPES
from oracle import oracle
from util import assert_dense_same

pytestmark = pytest.mark.gpu

SEMIRINGS = [
    ("PLUS_TIMES_FP64", Type.FP64, ["PLUS_FP64", "MIN_FP64", "SECOND_FP64", "TIMES_FP64"]),
    ("PLUS_TIMES_INT32", Type.INT32, ["PLUS_INT32", "MAX_INT32", "FIRST_INT32", "MINUS_INT32"]),
    ("MIN_PLUS_FP32", Type.FP32, ["MIN_FP32", "PLUS_FP32"]),
    ("MAX_MIN_INT64", Type.INT64, ["MAX_INT64", "PLUS_INT64"]),
    ("LOR_LAND_BOOL", Type.BOOL, ["LOR_BOOL", "LXOR_BOOL"]),
    ("PLUS_PAIR_INT64", Type.INT64, ["PLUS_INT64"]),
    ("ANY_PAIR_BOOL", Type.BOOL, ["LOR_BOOL"]),
    ("MIN_FIRSTJ_INT64", Type.INT64, ["MIN_INT64"]),
    ("PLUS_SECOND_FP32", Type.FP32, ["PLUS_FP32"]),
    ("TIMES_PLUS_UINT8", Type.UINT8, ["PLUS_UINT8"]),
]


def _random_matrix(rng, nrows, ncols, t, fmt, by_row, density):
    dt = NP_TYPES[t]
    vals = rng.integers(1, 4, (nrows, ncols))
    pres = rng.random((nrows, ncols)) < density
    if fmt == "full":
        return Matrix.dense(vals.astype(dt), by_row=by_row)
    if fmt == "bitmap":
        return Matrix.dense(vals.astype(dt), present=pres, by_row=by_row)
    S = sp.csr_matrix(np.where(pres, vals, 0).astype(np.float64))
    S.eliminate_zeros()
    M = Matrix.from_scipy(S, type=t, by_row=by_row)
    return M.to_hyper() if fmt == "hyper" else M


def _case(seed, lo=1, hi=40):
    rng = np.random.default_rng(1000 + seed)
    sr, t, accums = SEMIRINGS[rng.integers(len(SEMIRINGS))]
    m, k, n = (int(x) for x in rng.integers(lo, hi, 3))
    vector = rng.random() < 0.3
    if vector:
        n = 1
    a_tran, b_tran = bool(rng.random() < 0.35), bool(rng.random() < 0.35) and not vector
    fmts = ["sparse", "sparse", "bitmap", "full"]
    A = _random_matrix(rng, *( (k, m) if a_tran else (m, k) ), t, fmts[rng.integers(4)], bool(rng.random() < 0.5), rng.choice([0.05, 0.3, 0.8]))
    B = _random_matrix(rng, *( (n, k) if b_tran else (k, n) ), t, fmts[rng.integers(4)] if not vector else ["sparse", "bitmap", "full"][rng.integers(3)],
                       False if vector else bool(rng.random() < 0.5), rng.choice([0.05, 0.3, 0.8]))
    zt = Type.BOOL if sr.split("_")[1] in ("EQ", "NE", "GT", "LT", "GE", "LE") else t
    c_fmt = ["empty", "sparse", "bitmap", "full"][rng.integers(4)]
    c_by_row = False if vector else bool(rng.random() < 0.5)
    if c_fmt == "empty":
        C0 = Matrix.sparse_empty(m, n, zt, by_row=c_by_row)
    else:
        C0 = _random_matrix(rng, m, n, zt, c_fmt, c_by_row, 0.3)
    M = None
    if rng.random() < 0.6:
        M = _random_matrix(rng, m, n, Type.INT8, ["sparse", "bitmap", "full"][rng.integers(3)],
                           False if vector else bool(rng.random() < 0.5), 0.4)
        M.x = (M.x % 2).astype(np.int8) if M.x is not None else M.x          # valued masks carry zeros too
    accum = accums[rng.integers(len(accums))] if rng.random() < 0.5 else None
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "") \
        + ("T0" if a_tran else "") + ("T1" if b_tran else "")
    return sr, A, B, C0, M, accum, desc, vector


@pytest.mark.parametrize("seed", range(160))
def test_random_call_matches_the_definition(gba, seed):
    _check(gba, seed, *_case(seed))


def _hyper_case(seed):
    """Like _case, with hypersparse operands, mask and C mixed in (very sparse matrices: most vectors empty)."""
    rng = np.random.default_rng(9000 + seed)
    sr, t, accums = SEMIRINGS[rng.integers(len(SEMIRINGS))]
    m, k, n = (int(x) for x in rng.integers(2, 60, 3))
    a_tran, b_tran = bool(rng.random() < 0.35), bool(rng.random() < 0.35)
    pick = lambda: ["hyper", "hyper", "sparse", "bitmap"][rng.integers(4)]
    A = _random_matrix(rng, *((k, m) if a_tran else (m, k)), t, pick(), bool(rng.random() < 0.5), rng.choice([0.02, 0.1]))
    B = _random_matrix(rng, *((n, k) if b_tran else (k, n)), t, pick(), bool(rng.random() < 0.5), rng.choice([0.02, 0.1]))
    c_fmt = ["empty", "hyper", "sparse"][rng.integers(3)]
    c_by_row = bool(rng.random() < 0.5)
    C0 = Matrix.sparse_empty(m, n, t, by_row=c_by_row) if c_fmt == "empty" else _random_matrix(rng, m, n, t, c_fmt, c_by_row, 0.05)
    M = None
    if rng.random() < 0.6:
        M = _random_matrix(rng, m, n, Type.INT8, ["hyper", "sparse"][rng.integers(2)], bool(rng.random() < 0.5), 0.2)
        M.x = (M.x % 2).astype(np.int8)
    accum = accums[rng.integers(len(accums))] if rng.random() < 0.5 else None
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "") \
        + ("T0" if a_tran else "") + ("T1" if b_tran else "")
    return sr, A, B, C0, M, accum, desc, False


@pytest.mark.parametrize("seed", range(40))
def test_random_call_with_hypersparse_matrices(gba, seed):
    _check(gba, seed, *_hyper_case(seed))


CAST_TYPES = [Type.BOOL, Type.INT8, Type.UINT8, Type.INT16, Type.UINT16, Type.INT32, Type.UINT32, Type.INT64, Type.UINT64,
              Type.FP32, Type.FP64]


def _typed_matrix(rng, nrows, ncols, t, fmt, by_row, density):
    """Entries that do not survive a typecast unchanged: halves, negatives, values past 8 and 16 bits."""
    dt = np.dtype(NP_TYPES[t])
    pool = np.array([-300.5, -2.5, -1, 0.5, 1, 2, 3.5, 70000, 129, 255.5])
    vals = pool[rng.integers(len(pool), size=(nrows, ncols))]
    if dt.kind in "iu":                       # wrapped into the type's range; magnitudes stay small enough that
        v = np.trunc(np.abs(vals) if dt.kind == "u" else vals).astype(np.int64)      # fp32 sums are exact in any order
        vals = v.astype(dt)                   # int64 -> narrower integer wraps
    else:
        vals = vals.astype(dt) if dt.kind != "b" else (vals > 1)
    pres = rng.random((nrows, ncols)) < density
    if fmt == "full":
        return Matrix.dense(vals, by_row=by_row, type=t)
    if fmt == "bitmap":
        return Matrix.dense(vals, present=pres, by_row=by_row, type=t)
    r, c = np.nonzero(pres)
    if not by_row:
        o = np.lexsort((r, c)); r, c = r[o], c[o]
    vec, idx, nv = (r, c, nrows) if by_row else (c, r, ncols)
    p = np.zeros(nv + 1, np.int64)
    np.add.at(p, vec + 1, 1)
    return Matrix(ncols if by_row else nrows, nv, t, 2, not by_row, p=np.cumsum(p), i=idx, x=vals[r, c])


def _cast_case(seed):
    rng = np.random.default_rng(7000 + seed)
    sr, t, _ = SEMIRINGS[rng.integers(len(SEMIRINGS))]
    m, k, n = (int(x) for x in rng.integers(1, 30, 3))
    ta, tb, tc = (CAST_TYPES[rng.integers(len(CAST_TYPES))] for _ in range(3))
    fm = lambda: ["sparse", "sparse", "bitmap", "full"][rng.integers(4)]
    A = _typed_matrix(rng, m, k, ta, fm(), bool(rng.random() < 0.5), 0.4)
    B = _typed_matrix(rng, k, n, tb, fm(), bool(rng.random() < 0.5), 0.4)
    c_by_row = bool(rng.random() < 0.5)
    C0 = Matrix.sparse_empty(m, n, tc, by_row=c_by_row) if rng.random() < 0.3 else _typed_matrix(rng, m, n, tc, fm(), c_by_row, 0.3)
    M = None
    if rng.random() < 0.5:
        M = _random_matrix(rng, m, n, Type.INT8, ["sparse", "bitmap", "full"][rng.integers(3)], bool(rng.random() < 0.5), 0.4)
        M.x = (M.x % 2).astype(np.int8)
    tname = {v: k_ for k_, v in vars(Type).items() if isinstance(v, int)}[tc]
    ops = ["LOR", "LXOR"] if tc == Type.BOOL else ["PLUS", "MIN", "MAX", "SECOND", "FIRST"]
    accum = f"{ops[rng.integers(len(ops))]}_{tname}" if rng.random() < 0.5 else None
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "")
    return sr, A, B, C0, M, accum, desc, False


@pytest.mark.parametrize("seed", range(80))
def test_random_call_with_typecasts(gba, seed):
    """Operands and C of types other than the semiring's: cast while loading and on the write-back."""
    _check(gba, seed, *_cast_case(seed))


def _mixed_accum_case(seed):
    """As _cast_case, with an accumulator of a type that is not C's."""
    rng = np.random.default_rng(9000 + seed)
    sr, t, _ = SEMIRINGS[rng.integers(len(SEMIRINGS))]
    m, k, n = (int(x) for x in rng.integers(1, 30, 3))
    ta, tb, tc = (CAST_TYPES[rng.integers(len(CAST_TYPES))] for _ in range(3))
    tu = [u for u in CAST_TYPES if u != tc][rng.integers(len(CAST_TYPES) - 1)]
    fm = lambda: ["sparse", "sparse", "bitmap", "full"][rng.integers(4)]
    A = _typed_matrix(rng, m, k, ta, fm(), bool(rng.random() < 0.5), 0.4)
    B = _typed_matrix(rng, k, n, tb, fm(), bool(rng.random() < 0.5), 0.4)
    c_by_row = bool(rng.random() < 0.5)
    C0 = Matrix.sparse_empty(m, n, tc, by_row=c_by_row) if rng.random() < 0.15 else _typed_matrix(rng, m, n, tc, fm(), c_by_row, 0.4)
    M = None
    if rng.random() < 0.5:
        M = _random_matrix(rng, m, n, Type.INT8, ["sparse", "bitmap", "full"][rng.integers(3)], bool(rng.random() < 0.5), 0.4)
        M.x = (M.x % 2).astype(np.int8)
    uname = {v: k_ for k_, v in vars(Type).items() if isinstance(v, int)}[tu]
    ops = ["LOR", "LXOR", "LAND"] if tu == Type.BOOL else ["PLUS", "MIN", "MAX", "SECOND", "FIRST", "TIMES", "MINUS"]
    accum = f"{ops[rng.integers(len(ops))]}_{uname}"
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "")
    return sr, A, B, C0, M, accum, desc, False


@pytest.mark.parametrize("seed", range(80))
def test_random_call_with_an_accumulator_of_another_type(gba, seed):
    """Z = accum (C,T) in the accumulator's own type (Test/GB_spec_accum.m:45-63): both operands cast to it, the
    result cast to C's type, entries of C alone kept bit for bit, entries of T alone cast to C's type."""
    _check(gba, seed, *_mixed_accum_case(seed))


@pytest.mark.parametrize("seed", range(500, 548))
def test_random_call_matches_the_definition_larger(gba, seed):
    """Dimensions of 120..320: vectors reach the block-level hash kernels and the long-row dot paths."""
    _check(gba, seed, *_case(seed, 120, 320))


def _check(gba, seed, sr, A, B, C0, M, accum, desc, vector):
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    dM = M.to_device() if M is not None else None
    what = f"seed {seed}: {sr} accum={accum} desc='{desc}' A={A.sparsity}/{'c' if A.is_csc else 'r'} B={B.sparsity}/" \
           f"{'c' if B.is_csc else 'r'} C={C0.sparsity}/{C0.nvals} M={None if M is None else M.sparsity}"
    try:
        if vector:
            gba.mxv(dC, dM, accum, sr, dA, dB, desc)
        else:
            gba.mxm(dC, dM, accum, sr, dA, dB, desc)
    except GrbError as e:
        assert e.info == Info.NOT_IMPLEMENTED, what + f" -> {e}"
        pytest.skip(f"declined: {e}")
    p, v = oracle.mxm_spec(C0, M, accum, sr, A, B, desc)
    got = dC.to_host()
    assert_dense_same(got, p, v, exact=True, what=what)          # ANY_PAIR yields 1 everywhere: values are pinned too
    if C0.sparsity == Sparsity.HYPERSPARSE and got.sparsity != Sparsity.BITMAP:
        # a hypersparse C stays hypersparse: h lists exactly the non-empty vectors, in order
        assert got.sparsity == Sparsity.HYPERSPARSE, what
        assert np.all(np.diff(got.p) > 0) and np.all(np.diff(got.h) > 0), what


COMPLEX_SEMIRINGS = [
    ("PLUS_TIMES_FC64", Type.FC64, ["PLUS_FC64", "TIMES_FC64", "SECOND_FC64", "MINUS_FC64"]),
    ("PLUS_TIMES_FC32", Type.FC32, ["PLUS_FC32", "FIRST_FC32"]),
    ("TIMES_PLUS_FC64", Type.FC64, ["PLUS_FC64"]),
    ("ANY_FIRST_FC32", Type.FC32, ["PLUS_FC32"]),
    ("PLUS_PAIR_FC64", Type.FC64, ["PLUS_FC64", "RMINUS_FC64"]),
    ("PLUS_RMINUS_FC32", Type.FC32, ["TIMES_FC32"]),
    ("PLUS_SECOND_FC64", Type.FC64, ["PLUS_FC64"]),
]


def _random_complex_matrix(rng, nrows, ncols, t, fmt, by_row, density):
    """Gaussian integers: sums and products stay exact in both precisions, whatever the order."""
    dt = NP_TYPES[t]
    vals = (rng.integers(1, 3, (nrows, ncols)) + 1j * rng.integers(-1, 2, (nrows, ncols))).astype(dt)
    pres = rng.random((nrows, ncols)) < density
    if fmt == "full":
        return Matrix.dense(vals, by_row=by_row, type=t)
    if fmt == "bitmap":
        return Matrix.dense(vals, present=pres, by_row=by_row, type=t)
    r, c = np.nonzero(pres)
    if not by_row:
        o = np.lexsort((r, c)); r, c = r[o], c[o]
    vec, idx, nv = (r, c, nrows) if by_row else (c, r, ncols)
    p = np.zeros(nv + 1, np.int64)
    np.add.at(p, vec + 1, 1)
    M = Matrix(ncols if by_row else nrows, nv, t, 2, not by_row, p=np.cumsum(p), i=idx, x=vals[r, c])
    return M.to_hyper() if fmt == "hyper" else M


def _complex_case(seed):
    rng = np.random.default_rng(4000 + seed)
    sr, t, accums = COMPLEX_SEMIRINGS[rng.integers(len(COMPLEX_SEMIRINGS))]
    m, k, n = (int(x) for x in rng.integers(1, 24, 3))          # short sums: TIMES stays inside 2^24 in fp32
    vector = rng.random() < 0.3
    if vector:
        n = 1
    a_tran, b_tran = bool(rng.random() < 0.35), bool(rng.random() < 0.35) and not vector
    fm = lambda: ["sparse", "sparse", "hyper", "bitmap", "full"][rng.integers(5)]
    dens = 0.25 if sr.startswith("TIMES") else rng.choice([0.05, 0.3, 0.8])
    A = _random_complex_matrix(rng, *((k, m) if a_tran else (m, k)), t, fm(), bool(rng.random() < 0.5), dens)
    B = _random_complex_matrix(rng, *((n, k) if b_tran else (k, n)), t, fm() if not vector else ["sparse", "bitmap", "full"][rng.integers(3)],
                               False if vector else bool(rng.random() < 0.5), dens)
    c_fmt = ["empty", "sparse", "hyper", "bitmap", "full"][rng.integers(5)]
    if vector and c_fmt == "hyper":
        c_fmt = "sparse"
    c_by_row = False if vector else bool(rng.random() < 0.5)
    C0 = Matrix.sparse_empty(m, n, t, by_row=c_by_row) if c_fmt == "empty" else _random_complex_matrix(rng, m, n, t, c_fmt, c_by_row, 0.3)
    M = None
    if rng.random() < 0.6:
        M = _random_matrix(rng, m, n, Type.INT8, ["sparse", "bitmap", "full"][rng.integers(3)],
                           False if vector else bool(rng.random() < 0.5), 0.4)
        M.x = (M.x % 2).astype(np.int8) if M.x is not None else M.x
    accum = accums[rng.integers(len(accums))] if rng.random() < 0.5 else None
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "") \
        + ("T0" if a_tran else "") + ("T1" if b_tran else "")
    return sr, A, B, C0, M, accum, desc, vector


@pytest.mark.parametrize("seed", range(120))
def test_random_call_with_complex_semirings(gba, seed):
    _check(gba, seed, *_complex_case(seed))


MIXED_TYPES = [Type.BOOL, Type.INT8, Type.UINT16, Type.INT32, Type.INT64, Type.FP32, Type.FP64, Type.FC32, Type.FC64]


def _mixed_matrix(rng, nrows, ncols, t, fmt, by_row, density):
    if t in (Type.FC32, Type.FC64):
        return _random_complex_matrix(rng, nrows, ncols, t, fmt, by_row, density)
    return _random_matrix(rng, nrows, ncols, t, fmt, by_row, density)


def _mixed_case(seed):
    """A, B and C of unrelated built-in types, real or complex, under a real or a complex semiring."""
    rng = np.random.default_rng(6000 + seed)
    pool = COMPLEX_SEMIRINGS[:2] + COMPLEX_SEMIRINGS[4:] + SEMIRINGS[:2] + SEMIRINGS[4:6]
    sr, t, _ = pool[rng.integers(len(pool))]
    ta, tb, tc = (MIXED_TYPES[rng.integers(len(MIXED_TYPES))] for _ in range(3))
    if not any(x in (Type.FC32, Type.FC64) for x in (t, tc)):
        tc = Type.FC64                               # every case crosses the boundary somewhere
    m, k, n = (int(x) for x in rng.integers(1, 20, 3))
    fm = lambda: ["sparse", "sparse", "hyper", "bitmap", "full"][rng.integers(5)]
    a_tran, b_tran = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    A = _mixed_matrix(rng, *((k, m) if a_tran else (m, k)), ta, fm(), bool(rng.random() < 0.5), 0.3)
    B = _mixed_matrix(rng, *((n, k) if b_tran else (k, n)), tb, fm(), bool(rng.random() < 0.5), 0.3)
    c_by_row = bool(rng.random() < 0.5)
    c_fmt = ["empty", "sparse", "bitmap", "full"][rng.integers(4)]
    C0 = Matrix.sparse_empty(m, n, tc, by_row=c_by_row) if c_fmt == "empty" else _mixed_matrix(rng, m, n, tc, c_fmt, c_by_row, 0.3)
    M = None
    if rng.random() < 0.5:
        M = _random_matrix(rng, m, n, Type.INT8, ["sparse", "bitmap", "full"][rng.integers(3)], bool(rng.random() < 0.5), 0.4)
        M.x = (M.x % 2).astype(np.int8)
    tname = {v: k_ for k_, v in vars(Type).items() if isinstance(v, int)}[tc]
    ops = ["LOR", "LXOR"] if tc == Type.BOOL else ["PLUS", "SECOND", "FIRST"]
    accum = f"{ops[rng.integers(len(ops))]}_{tname}" if rng.random() < 0.5 else None
    desc = ("R" if rng.random() < 0.3 else "") + (("S" if rng.random() < 0.5 else "") + ("C" if rng.random() < 0.4 else "") if M is not None else "") \
        + ("T0" if a_tran else "") + ("T1" if b_tran else "")
    return sr, A, B, C0, M, accum, desc, False


@pytest.mark.parametrize("seed", range(100))
def test_random_call_across_real_and_complex_types(gba, seed):
    _check(gba, seed, *_mixed_case(seed))
