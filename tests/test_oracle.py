"""CPU-only checks of the oracle: pinned against the reference's own known-answer
demo output, then cross-checked against independent numpy/scipy computations."""
import json
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, Sparsity
from graphblas_amd import gen
from oracle import oracle
from util import assert_same, dense_of

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_gauss_demo_known_answers():
    """The oracle's saxbit, dot3, saxpy3 (colscale/rowscale), dot4, saxpy4 and saxpy5
    reproduce Demo/Output/gauss_demo.out (fixture tests/golden/gauss_demo_mxm.json)."""
    want = json.load(open(os.path.join(GOLDEN, "gauss_demo_mxm.json")))
    text = oracle.gauss_replay()
    blocks = re.split(r"^(?=[AC] )", text, flags=re.M)
    blocks = [b for b in blocks if b.strip()]
    assert len(blocks) == len(want) == 7
    entry = re.compile(r"\(\s*(-?\d+),\s*(-?\d+)\)|\s{6}\.\s{5}")
    for blk, w in zip(blocks, want):
        lines = blk.strip().splitlines()
        assert lines[0] == w["name"]
        for r in range(4):
            row = [None if m.group(1) is None else [int(m.group(1)), int(m.group(2))]
                   for m in entry.finditer(lines[1 + r].split(":", 1)[1])]
            assert row == w["matrix"][r], (w["name"], r)
    methods = ["bitmap saxpy", "dot3", "colscale", "rowscale", "dot4", "saxpy4", "saxpy5"]
    for w, m in zip(want, methods):
        assert m in w["burble"]


def test_wildtype_demo_known_answers():
    """The oracle's saxpy3 and dot3 on the 192-byte user-defined type of Demo/Program/wildtype_demo.c reproduce
    Demo/Output/wildtype_demo.out (fixture tests/golden/wildtype_demo_mxm.json): positions, the text each
    operator leaves behind, and every number at the precision the reference prints ("%10.1f")."""
    want = json.load(open(os.path.join(GOLDEN, "wildtype_demo_mxm.json")))
    blocks = [b for b in re.split(r"^== ", oracle.wildtype_replay(), flags=re.M) if b.strip()]
    assert len(blocks) == len(want) == 4
    line = re.compile(r"^\((\d+),(\d+)\) \[(.*?)\] (.*)$")
    for blk, w in zip(blocks, want):
        lines = blk.strip().splitlines()
        assert lines[0] == w["name"]
        assert len(lines) - 1 == len(w["entries"])
        for got, e in zip(lines[1:], w["entries"]):
            m = line.match(got)
            assert (int(m.group(1)), int(m.group(2)), m.group(3)) == (e["i"], e["j"], e["text"]), w["name"]
            vals = [float(v) for v in m.group(4).split()]
            flat = [v for row in e["stuff"] for v in row]
            assert ["%.1f" % v for v in vals] == ["%.1f" % v for v in flat], (w["name"], e["i"], e["j"])
    assert "saxpy" in want[2]["burble"] and "dot3" in want[3]["burble"]


def _rand_sparse(n, m, density, seed, ints=True):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csr")
    S.data = rng.integers(1, 9, len(S.data)).astype(np.float64) if ints else 1.0 - rng.random(len(S.data))
    return S


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_saxpy3_matches_scipy(seed):
    A = _rand_sparse(60, 50, 0.08, seed)
    B = _rand_sparse(50, 70, 0.08, seed + 10)
    # internal CSC vectors: use by_row=False so vectors are columns
    Am, Bm = Matrix.from_scipy(A, by_row=False), Matrix.from_scipy(B, by_row=False)
    C = oracle.saxpy3(None, False, False, Am, Bm, "PLUS_TIMES_FP64")
    want = Matrix.from_scipy((A @ B).tocsc(), by_row=False)
    assert_same(C, want, exact=True)


def test_saxpy3_masks():
    A = _rand_sparse(40, 40, 0.15, 5)
    B = _rand_sparse(40, 40, 0.15, 6)
    Mk = _rand_sparse(40, 40, 0.3, 7)
    Am, Bm, Mm = (Matrix.from_scipy(X, by_row=False) for X in (A, B, Mk))
    full = (A @ B).toarray()
    pat = ((A != 0).astype(int) @ (B != 0).astype(int)).toarray() > 0
    mpat = Mk.toarray() != 0
    for comp in (False, True):
        C = oracle.saxpy3(Mm, comp, True, Am, Bm, "PLUS_TIMES_FP64")
        p, v = dense_of(C)
        allow = mpat != comp
        assert np.array_equal(p, pat & allow)
        assert np.array_equal(v[p], full[p])


def test_dot3_triangles():
    G = gen.rmat(8, 8, seed=3, symmetric=True, drop_diagonal=True, values="ones")
    L = gen.tril_strict(G)
    C = oracle.dot3(L, True, L, L, "PLUS_PAIR_INT64")
    S = L.to_scipy()
    S.data[:] = 1
    want = (S @ S.T).multiply(S)
    r, c, v = C.coo()
    assert v.sum() == int(want.sum())
    assert C.nzombies == L.nvals - len(v)
    W = sp.csr_matrix(want)
    W.eliminate_zeros()
    assert len(v) == W.nnz


def test_flopcount_formula():
    A = Matrix.from_scipy(_rand_sparse(50, 50, 0.1, 11), by_row=False)
    B = Matrix.from_scipy(_rand_sparse(50, 30, 0.1, 12), by_row=False)
    mw, fl = oracle.saxpy3_flopcount(None, False, A, B)
    alen = np.diff(A.p)
    per = np.array([(B.p[j + 1] - B.p[j]) + alen[B.i[B.p[j]:B.p[j + 1]]].sum() if B.p[j + 1] > B.p[j] else 0
                    for j in range(B.vdim)])
    assert mw == 0
    assert np.array_equal(fl, np.concatenate([[0], np.cumsum(per)]))


@pytest.mark.parametrize("desc", ["", "T0", "T1", "T0T1"])
def test_mxm_spec_transposes(desc):
    rng = np.random.default_rng(4)
    A = _rand_sparse(12, 12, 0.3, 21)
    B = _rand_sparse(12, 12, 0.3, 22)
    Am, Bm = Matrix.from_scipy(A), Matrix.from_scipy(B)
    Cm = Matrix.sparse_empty(12, 12, Type.FP64)
    p, v = oracle.mxm_spec(Cm, None, None, "PLUS_TIMES_FP64", Am, Bm, desc)
    X = A.T if "T0" in desc else A
    Y = B.T if "T1" in desc else B
    want = (X @ Y).toarray()
    wp = ((X != 0).astype(int) @ (Y != 0).astype(int)).toarray() > 0
    assert np.array_equal(p, wp)
    assert np.array_equal(v[p], want[wp])


def test_mxm_spec_accum_mask_replace():
    A = _rand_sparse(10, 10, 0.3, 31)
    B = _rand_sparse(10, 10, 0.3, 32)
    C0 = _rand_sparse(10, 10, 0.4, 33)
    Mk = _rand_sparse(10, 10, 0.5, 34)
    Am, Bm, Cm, Mm = (Matrix.from_scipy(X) for X in (A, B, C0, Mk))
    T = (A @ B).toarray()
    tp = ((A != 0).astype(int) @ (B != 0).astype(int)).toarray() > 0
    cp, cv = C0.toarray() != 0, C0.toarray()
    mp = Mk.toarray() != 0
    p, v = oracle.mxm_spec(Cm, Mm, "PLUS_FP64", "PLUS_TIMES_FP64", Am, Bm, "RSC")
    zp = cp | tp
    zv = np.where(cp & tp, cv + T, np.where(cp, cv, T))
    allow = ~mp
    assert np.array_equal(p, zp & allow)
    assert np.array_equal(v[p], zv[p])
    p2, v2 = oracle.mxm_spec(Cm, Mm, None, "PLUS_TIMES_FP64", Am, Bm, "S")
    want_p = np.where(mp, tp, cp)
    assert np.array_equal(p2, want_p)
    assert np.array_equal(v2[p2], np.where(mp, T, cv)[p2])


def test_semiring_arithmetic_corner_cases():
    """Integer wrap, GB_idiv rules (Source/GB_math.h:27-260) and omit-NaN min."""
    one = lambda t, a: Matrix(1, 1, t, Sparsity.SPARSE, True, p=[0, 1], i=[0], x=np.array([a]))
    def prod(sr, t, a, b):
        return oracle.saxpy3(None, False, False, one(t, a), one(t, b), sr).x[0]
    assert prod("PLUS_TIMES_INT8", Type.INT8, 100, 3) == np.int8(300 - 256)
    assert prod("PLUS_DIV_INT32", Type.INT32, 7, 0) == np.iinfo(np.int32).max
    assert prod("PLUS_DIV_INT32", Type.INT32, -7, 0) == np.iinfo(np.int32).min
    assert prod("PLUS_DIV_INT32", Type.INT32, 0, 0) == 0
    assert prod("PLUS_DIV_UINT8", Type.UINT8, 9, 0) == 255
    assert prod("PLUS_RDIV_INT16", Type.INT16, 2, 9) == 4
    assert prod("PLUS_MIN_FP64", Type.FP64, np.nan, 2.0) == 2.0
    assert prod("LOR_GT_FP32", Type.FP32, 2.0, 1.0) == True  # noqa: E712


def test_mxm_spec_reads_hypersparse_like_sparse():
    """A hypersparse matrix is the same matrix: the definition gives the same C either way."""
    A = Matrix.from_scipy(_rand_sparse(30, 40, 0.04, 21), by_row=True)
    B = Matrix.from_scipy(_rand_sparse(40, 25, 0.04, 22), by_row=False)
    Mk = Matrix.from_scipy(_rand_sparse(30, 25, 0.2, 23), by_row=True)
    C0 = Matrix.from_scipy(_rand_sparse(30, 25, 0.05, 24), by_row=False)
    Ah, Bh, Mh, Ch = A.to_hyper(), B.to_hyper(), Mk.to_hyper(), C0.to_hyper()
    assert Ah.nvec < A.vdim and len(Ah.p) == Ah.nvec + 1
    r0, c0, v0 = A.coo()
    r1, c1, v1 = Ah.coo()
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
    for desc in ("", "C", "RS"):
        p0, x0 = oracle.mxm_spec(C0, Mk, "PLUS_FP64", "PLUS_TIMES_FP64", A, B, desc)
        p1, x1 = oracle.mxm_spec(Ch, Mh, "PLUS_FP64", "PLUS_TIMES_FP64", Ah, Bh, desc)
        assert np.array_equal(p0, p1) and np.array_equal(x0[p0], x1[p1])


def test_typecast_rules():
    """Source/GB_casting.h: floating point to integer saturates and sends NaN to 0; anything to bool is x != 0."""
    x = np.array([np.nan, -np.inf, np.inf, -1.5, 2.9, 300.0, -200.0, 3e9, -3e9, 1e20, 0.0], np.float64)
    assert oracle.cast(x, Type.FP64, Type.INT8).tolist() == [0, -128, 127, -1, 2, 127, -128, 127, -128, 127, 0]
    assert oracle.cast(x, Type.FP64, Type.UINT8).tolist() == [0, 0, 255, 0, 2, 255, 0, 255, 0, 255, 0]
    assert oracle.cast(x, Type.FP64, Type.INT32).tolist() == [0, -2**31, 2**31 - 1, -1, 2, 300, -200, 2**31 - 1, -2**31, 2**31 - 1, 0]
    assert oracle.cast(x, Type.FP64, Type.UINT64).tolist() == [0, 0, 2**64 - 1, 0, 2, 300, 0, 3000000000, 0, 2**64 - 1, 0]
    assert oracle.cast(x, Type.FP64, Type.BOOL).tolist() == [True, True, True, True, True, True, True, True, True, True, False]
    i = np.array([-1, 300, -129, 65536 + 7], np.int32)
    assert oracle.cast(i, Type.INT32, Type.UINT8).tolist() == [255, 44, 127, 7]          # integer to integer: C's modular cast
    assert oracle.cast(i, Type.INT32, Type.INT8).tolist() == [-1, 44, 127, 7]
    assert oracle.cast(i, Type.INT32, Type.FP32).tolist() == [-1.0, 300.0, -129.0, 65543.0]
    assert oracle.cast(np.array([2.5], np.float32), Type.FP32, Type.INT64).tolist() == [2]


def test_mxm_spec_typecasts_operands_and_result():
    """A int8 and B fp64 under PLUS_TIMES_INT32 into an fp32 C with a PLUS_FP32 accumulator."""
    A = Matrix.dense(np.array([[1, 2], [3, 4]], np.int8))
    B = Matrix.dense(np.array([[0.5, 1.9], [2.2, -1.7]], np.float64))
    C0 = Matrix.dense(np.array([[10, 20], [30, 40]], np.float32))
    p, v = oracle.mxm_spec(C0, None, "PLUS_FP32", "PLUS_TIMES_INT32", A, B)
    Bi = np.array([[0, 1], [2, -1]])                # B cast to int32 truncates toward zero
    want = np.array([[10, 20], [30, 40]]) + np.array([[1, 2], [3, 4]]) @ Bi
    assert p.all() and v.dtype == np.float32 and np.array_equal(v, want.astype(np.float32))


def test_accumulator_of_another_type_works_in_its_own_type():
    """Test/GB_spec_accum.m:45-63: where C and T both have an entry, both are cast to the accumulator's type and the
    result is cast to C's; an entry of C alone is kept as it is, an entry of T alone is cast to C's type."""
    A = Matrix.dense(np.array([[300.5, 0.0], [2.5, 1.0]], np.float64))
    B = Matrix.dense(np.array([[1.0, 0.0], [0.0, 1.0]], np.float64), present=np.array([[True, False], [False, True]]))
    C0 = Matrix.dense(np.array([[10.7, 20.9], [0.0, -3.25]], np.float64), present=np.array([[True, True], [False, True]]))
    p, v = oracle.mxm_spec(C0, None, "PLUS_INT8", "PLUS_TIMES_FP64", A, B)
    # T = A*B = [[300.5, 0], [2.5, 1]], every entry present (A is full; B's pattern picks one term each)
    # (0,0): int8 (10.7) + int8 (300.5) = 10 + 127 = 137 -> wraps to -119;  (0,1): 20 + 0 = 20;  (1,0): T alone -> 2.5;
    # (1,1): -3 + 1 = -2
    assert p.all() and v.dtype == np.float64
    assert v.tolist() == [[-119.0, 20.0], [2.5, -2.0]]
    # under a mask that hides (0,1), that entry of C stays 20.9
    M = Matrix.dense(np.array([[1, 0], [1, 1]], np.int8))
    p, v = oracle.mxm_spec(C0, M, "PLUS_INT8", "PLUS_TIMES_FP64", A, B)
    assert v.tolist() == [[-119.0, 20.9], [2.5, -2.0]]
    # T is cast to the accumulator's type from the monoid's type, not through C's: int8 C, fp64 accumulator
    C1 = Matrix.dense(np.array([[1, 1], [1, 1]], np.int8))
    p, v = oracle.mxm_spec(C1, None, "PLUS_FP64", "PLUS_TIMES_FP64", A, B)
    assert v.dtype == np.int8 and v.tolist() == [[127, 1], [3, 2]]     # 1 + 300.5 = 301.5 saturates; 1 + 2.5 = 3.5 -> 3


def _rand_complex(n, m, density, seed, t=Type.FC64, by_row=True, gaussian_integers=True):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csr").astype(np.complex128)
    k = len(S.data)
    if gaussian_integers:
        S.data = rng.integers(1, 5, k) + 1j * rng.integers(-4, 5, k)
    else:
        S.data = rng.uniform(0.5, 2.0, k) + 1j * rng.uniform(-2.0, 2.0, k)
    return Matrix.from_scipy(S, type=t, by_row=by_row)


@pytest.mark.parametrize("t", [Type.FC32, Type.FC64])
def test_complex_semirings_against_numpy(t):
    """The built-in complex semirings (Source/Factories/GB_AxB_factory.c:46-243 with GB_COMPLEX): Gaussian-integer
    values make every sum and product exact, so numpy's complex matmul is the answer bit for bit."""
    name = "FC32" if t == Type.FC32 else "FC64"
    A, B = _rand_complex(14, 11, 0.3, 51, t), _rand_complex(11, 9, 0.3, 52, t)
    C0 = Matrix.sparse_empty(14, 9, t)
    a, b = A.to_scipy().toarray(), B.to_scipy().toarray()
    ap, bp = (a != 0), (b != 0)
    tp = (ap.astype(int) @ bp.astype(int)) > 0
    p, v = oracle.mxm_spec(C0, None, None, f"PLUS_TIMES_{name}", A, B)
    assert np.array_equal(p, tp) and np.array_equal(v[p], (a @ b)[tp])
    p, v = oracle.mxm_spec(C0, None, None, f"PLUS_FIRST_{name}", A, B)
    assert np.array_equal(v[p], (a @ bp.astype(a.dtype))[tp])
    p, v = oracle.mxm_spec(C0, None, None, f"PLUS_SECOND_{name}", A, B)
    assert np.array_equal(v[p], (ap.astype(a.dtype) @ b)[tp])
    p, v = oracle.mxm_spec(C0, None, None, f"PLUS_PAIR_{name}", A, B)
    assert np.array_equal(v[p], (ap.astype(int) @ bp.astype(int))[tp].astype(a.dtype))
    p, v = oracle.mxm_spec(C0, None, None, f"PLUS_MINUS_{name}", A, B)
    assert np.array_equal(v[p], (a @ bp.astype(a.dtype) - ap.astype(a.dtype) @ b)[tp])
    # the saxpy3 method on the same operands (by column), against the definition
    Ac, Bc = _rand_complex(14, 11, 0.3, 51, t, by_row=False), _rand_complex(11, 9, 0.3, 52, t, by_row=False)
    T = oracle.saxpy3(None, False, False, Ac, Bc, f"PLUS_TIMES_{name}", wait=True)
    assert np.array_equal(T.to_scipy().toarray(), a @ b)


def test_complex_division_is_the_librarys_own():
    """GB_FC64_div (Source/GB_math.h:301-340): real and imaginary divisors divide componentwise, the rest goes
    through Smith's two cases; single complex divides in double and rounds (GB_FC32_div, :379-389)."""
    one = lambda t, a: Matrix(1, 1, t, Sparsity.SPARSE, True, p=[0, 1], i=[0], x=np.array([a]))
    def quot(t, x, y, sr="PLUS_DIV"):
        name = "FC32" if t == Type.FC32 else "FC64"
        return oracle.saxpy3(None, False, False, one(t, x), one(t, y), f"{sr}_{name}").x[0]
    assert quot(Type.FC64, 6 + 8j, 2 + 0j) == 3 + 4j
    assert quot(Type.FC64, 6 + 8j, 0 + 2j) == 4 - 3j
    assert quot(Type.FC64, 1 + 0j, 0j).real == np.inf                    # x / (0,0) divides by the real part
    # |yr| >= |yi|: r = yi/yr, d = yr + r*yi
    x, y = 1.25 + 2.5j, 4 + 2j
    r = y.imag / y.real; d = y.real + r * y.imag
    assert quot(Type.FC64, x, y) == complex((x.real + x.imag * r) / d, (x.imag - x.real * r) / d)
    # |yr| < |yi|: r = yr/yi, d = r*yr + yi
    y = 1 + 3j
    r = y.real / y.imag; d = r * y.real + y.imag
    assert quot(Type.FC64, x, y) == complex((x.real * r + x.imag) / d, (x.imag * r - x.real) / d)
    assert quot(Type.FC64, x, y, "PLUS_RDIV") == quot(Type.FC64, y, x)
    got = quot(Type.FC32, np.complex64(x), np.complex64(y))
    assert got == np.complex64(complex((x.real * r + x.imag) / d, (x.imag * r - x.real) / d))
    assert got.dtype == np.complex64


def test_complex_accum_and_mask_in_the_definition():
    t = Type.FC64
    A, B = _rand_complex(10, 10, 0.3, 61, t), _rand_complex(10, 10, 0.3, 62, t)
    C0 = Matrix.dense((np.arange(100).reshape(10, 10) * (1 - 1j)).astype(np.complex128))
    Mk = Matrix.from_scipy(_rand_sparse(10, 10, 0.5, 63))
    a, b = A.to_scipy().toarray(), B.to_scipy().toarray()
    tp = ((a != 0).astype(int) @ (b != 0).astype(int)) > 0
    p, v = oracle.mxm_spec(C0, None, "PLUS_FC64", "PLUS_TIMES_FC64", A, B)
    assert p.all() and np.array_equal(v, C0.x.reshape(10, 10) + a @ b)
    p, v = oracle.mxm_spec(Matrix.sparse_empty(10, 10, t), Mk, None, "PLUS_TIMES_FC64", A, B, "S")
    mp = Mk.to_scipy().toarray() != 0
    assert np.array_equal(p, tp & mp) and np.array_equal(v[p], (a @ b)[p])


def test_typecast_rules_across_real_and_complex():
    """Source/GB_casting.h:211-446: a real value becomes the real part; a complex value gives its real part to every
    real type (through the saturating rule for integers) and is true as a bool when either part is non-zero."""
    z = np.array([1.5 - 2j, 0 + 3j, 0j, -300.7 + 1j, np.nan + 0j, 1e20 - 1j], np.complex128)
    assert oracle.cast(z, Type.FC64, Type.FP64).tolist()[:4] == [1.5, 0.0, 0.0, -300.7]
    assert oracle.cast(z, Type.FC64, Type.BOOL).tolist() == [True, True, False, True, True, True]
    assert oracle.cast(z, Type.FC64, Type.INT8).tolist() == [1, 0, 0, -128, 0, 127]
    assert oracle.cast(z, Type.FC64, Type.UINT8).tolist() == [1, 0, 0, 0, 0, 255]
    got = oracle.cast(z[:4], Type.FC64, Type.FC32)
    assert got.dtype == np.complex64 and np.array_equal(got, z[:4].astype(np.complex64))
    r = np.array([2.5, -1.0, 0.0], np.float32)
    assert np.array_equal(oracle.cast(r, Type.FP32, Type.FC64), np.array([2.5 + 0j, -1 + 0j, 0j]))
    i = np.array([-3, 7], np.int16)
    assert np.array_equal(oracle.cast(i, Type.INT16, Type.FC32), np.array([-3 + 0j, 7 + 0j], np.complex64))
