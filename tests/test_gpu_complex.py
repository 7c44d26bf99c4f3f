"""GPU parity for the built-in complex semirings through gba_mxm / gba_mxv / gba_vxm:
(PLUS, TIMES, ANY) x (FIRST, SECOND, PAIR, PLUS, MINUS, RMINUS, TIMES, DIV, RDIV) over GxB_FC32 and GxB_FC64
(Source/Factories/GB_AxB_factory.c:46-243 with GB_COMPLEX, GB_AxB_type_factory.c:100-154).

The device evaluates every entry as one dot product in increasing k, which is the order of the definition
restated in oracle.mxm_spec, with the same unfused multiply and add: values are compared bit for bit.
The reference holds no runnable known answers for its complex semirings (its test suite is MATLAB-only):
parity of these operators is pinned by the oracle's restatement of GB_complex.h / GB_math.h alone."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, GrbError, Info
from oracle import oracle
from util import assert_dense_same

pytestmark = pytest.mark.gpu
NAME = {Type.FC32: "FC32", Type.FC64: "FC64"}


def _rand(n, m, density, seed, t=Type.FC64, by_row=True):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csr").astype(np.complex128)
    k = len(S.data)
    S.data = rng.uniform(0.5, 2.0, k) + 1j * rng.uniform(-2.0, 2.0, k)
    return Matrix.from_scipy(S, type=t, by_row=by_row)


def _mask(n, m, density, seed):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csr")
    S.data = rng.integers(0, 2, len(S.data)).astype(np.float64)
    return Matrix.from_scipy(S, type=Type.INT8)


@pytest.mark.parametrize("t", [Type.FC32, Type.FC64])
@pytest.mark.parametrize("mult", ["FIRST", "SECOND", "PAIR", "PLUS", "MINUS", "RMINUS", "TIMES", "DIV", "RDIV"])
@pytest.mark.parametrize("add", ["PLUS", "TIMES", "ANY"])
def test_every_complex_semiring(gba, add, mult, t):
    n = 48
    A, B = _rand(n, 40, 0.12, 1, t), _rand(40, n, 0.12, 2, t)
    C0 = Matrix.sparse_empty(n, n, t)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    sr = f"{add}_{mult}_{NAME[t]}"
    gba.mxm(dC, None, None, sr, dA, dB, None)
    p, v = oracle.mxm_spec(C0, None, None, sr, A, B)
    got = dC.to_host()
    assert got.type == t
    # ANY may keep any of the terms: the device and the definition both keep the first in k
    assert_dense_same(got, p, v, exact=True, what=sr)


@pytest.mark.parametrize("desc", ["", "T0", "T1", "T0T1"])
@pytest.mark.parametrize("fmt", ["rrr", "ccc", "rcr", "crc"])
def test_complex_transposes_and_formats(gba, desc, fmt):
    n = 40
    A = _rand(n, n, 0.1, 3, by_row=fmt[1] == "r")
    B = _rand(n, n, 0.1, 4, by_row=fmt[2] == "r")
    C0 = Matrix.sparse_empty(n, n, Type.FC64, by_row=fmt[0] == "r")
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, None, "PLUS_TIMES_FC64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, None, None, "PLUS_TIMES_FC64", A, B, desc)
    got = dC.to_host()
    assert got.is_csc == C0.is_csc
    assert_dense_same(got, p, v, exact=True, what=f"{desc}/{fmt}")


@pytest.mark.parametrize("desc", ["S", "SC", "", "C", "RS", "RSC"])
def test_complex_masked(gba, desc):
    n = 50
    A, B = _rand(n, n, 0.1, 5), _rand(n, n, 0.1, 6)
    M = _mask(n, n, 0.3, 7)
    C0 = Matrix.sparse_empty(n, n, Type.FC64)
    dA, dB, dM, dC = A.to_device(), B.to_device(), M.to_device(), C0.to_device()
    gba.mxm(dC, dM, None, "PLUS_TIMES_FC64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, M, None, "PLUS_TIMES_FC64", A, B, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=desc)


@pytest.mark.parametrize("accum", ["PLUS_FC64", "TIMES_FC64", "SECOND_FC64", "MINUS_FC64", "DIV_FC64"])
@pytest.mark.parametrize("cfmt", ["full", "bitmap"])
def test_complex_accumulated_into_a_dense_C(gba, accum, cfmt):
    n = 30
    rng = np.random.default_rng(8)
    A, B = _rand(n, n, 0.15, 9), _rand(n, n, 0.15, 10)
    vals = rng.uniform(0.5, 2, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))
    C0 = Matrix.dense(vals, present=None if cfmt == "full" else rng.random((n, n)) < 0.5)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, accum, "PLUS_TIMES_FC64", dA, dB, None)
    p, v = oracle.mxm_spec(C0, None, accum, "PLUS_TIMES_FC64", A, B)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{accum}/{cfmt}")


@pytest.mark.parametrize("t", [Type.FC32, Type.FC64])
def test_complex_mxv_and_vxm(gba, t):
    """w = A*u and w' = u'*A; vxm hands the multiplier its operands flipped (Source/GrB_vxm.c:63-72), which
    MINUS and DIV show."""
    n = 64
    dt = np.complex64 if t == Type.FC32 else np.complex128
    rng = np.random.default_rng(11)
    A = _rand(n, n, 0.1, 12, t)
    u = Matrix.full_vector((rng.uniform(0.5, 2, n) + 1j * rng.uniform(-1, 1, n)).astype(dt))
    idx = np.sort(rng.choice(n, 20, replace=False))
    us = Matrix.sparse_vector(n, idx, (rng.uniform(0.5, 2, 20) + 1j * rng.uniform(-1, 1, 20)).astype(dt))
    w0 = Matrix.sparse_vector(n, [], np.zeros(0, dt))
    dA = A.to_device()
    for vec in (u, us):
        for mult in ("TIMES", "MINUS", "DIV", "FIRST"):
            sr = f"PLUS_{mult}_{NAME[t]}"
            dw, dv = w0.to_device(), vec.to_device()
            gba.mxv(dw, None, None, sr, dA, dv, None)
            p, v = oracle.mxm_spec(w0, None, None, sr, A, vec, "")
            assert_dense_same(dw.to_host(), p, v, exact=True, what=f"mxv {sr}")
            dw = w0.to_device()
            gba.vxm(dw, None, None, sr, dv, dA, None)
            # w' = u'*A  <=>  w = A'*u with mult (x = u(k), y = A(k,j))
            p, v = oracle.mxm_spec(w0, None, None, sr, A, vec, "T0", flipxy=True)
            assert_dense_same(dw.to_host(), p, v, exact=True, what=f"vxm {sr}")


def test_complex_operands_staged_from_the_host(gba):
    n = 32
    A, B = _rand(n, n, 0.2, 13), _rand(n, n, 0.2, 14)
    C0 = Matrix.sparse_empty(n, n, Type.FC64)
    dC = C0.to_device()
    keep = []

    class HostView:
        def __init__(self, M):
            self.view = M.view(keep)

        def ref(self):
            import ctypes
            return ctypes.byref(self.view)

    gba.mxm(dC, None, None, "PLUS_TIMES_FC64", HostView(A), HostView(B), None)
    p, v = oracle.mxm_spec(C0, None, None, "PLUS_TIMES_FC64", A, B)
    assert_dense_same(dC.to_host(), p, v, exact=True)


def test_complex_infinities_follow_annex_g(gba):
    """(inf + i nan) * (1 + i) is an infinity, not nan + i nan: the C99 operator the CPU library compiles
    (Source/Shared/GB_complex.h:237-238) recovers it, and so does the device."""
    one = lambda a: Matrix(1, 1, Type.FC64, 2, True, p=[0, 1], i=[0], x=np.array([a], np.complex128))
    for x, y in ((complex(np.inf, np.nan), 1 + 1j), (complex(np.inf, 1), complex(0, np.inf)), (complex(1e308, 1e308), complex(1e308, 1e308))):
        A, B = one(x), one(y)
        C0 = Matrix.sparse_empty(1, 1, Type.FC64, by_row=False)
        dC = C0.to_device()
        gba.mxm(dC, None, None, "PLUS_TIMES_FC64", A.to_device(), B.to_device(), None)
        p, v = oracle.mxm_spec(C0, None, None, "PLUS_TIMES_FC64", A, B)
        got = dC.to_host().x[0]
        assert p.all()
        assert np.array_equal(np.array([got]).view(np.float64), v.reshape(-1).view(np.float64), equal_nan=True), (x, y, got, v)


def test_complex_calls_that_are_declined(gba):
    n = 16
    A = _rand(n, n, 0.2, 15)
    rng = np.random.default_rng(16)
    R = Matrix.from_scipy(sp.random(n, n, density=0.2, random_state=rng, format="csr"), type=Type.FP64)
    dA, dR = A.to_device(), R.to_device()
    dC = Matrix.sparse_empty(n, n, Type.FC64).to_device()
    with pytest.raises(GrbError) as e:                            # GrB_MIN_FC64 does not exist
        gba.mxm(dC, None, None, "MIN_TIMES_FC64", dA, dA, None)
    assert e.value.info == Info.DOMAIN_MISMATCH
    with pytest.raises(GrbError) as e:                            # the accumulator would have to be cast
        gba.mxm(dC, None, "PLUS_FP64", "PLUS_TIMES_FC64", dA, dA, None)
    assert e.value.info == Info.NOT_IMPLEMENTED


@pytest.mark.parametrize("desc", ["", "R", "S", "RS", "SC", "RSC", "C", "T0", "ST1"])
@pytest.mark.parametrize("accum", [None, "PLUS_FC64", "SECOND_FC64", "MINUS_FC64"])
@pytest.mark.parametrize("masked", [False, True])
def test_complex_into_a_sparse_C_that_holds_entries(gba, desc, accum, masked):
    """C<M> = accum (C, A*B) with a sparse, non-empty C (Source/GB_accum_mask.c): every combination of accum /
    mask / complement / replace; the values of a 16-byte type ride through the write-back as positions."""
    if not masked and any(ch in desc for ch in "SC"):
        pytest.skip("mask modifiers without a mask")
    n = 45
    A, B = _rand(n, n, 0.1, 21), _rand(n, n, 0.1, 22)
    C0 = _rand(n, n, 0.2, 23, by_row=(len(desc) % 2 == 0))
    M = _mask(n, n, 0.35, 24) if masked else None
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    dM = M.to_device() if masked else None
    gba.mxm(dC, dM, accum, "PLUS_TIMES_FC64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, M, accum, "PLUS_TIMES_FC64", A, B, desc)
    got = dC.to_host()
    assert got.is_csc == C0.is_csc and got.type == Type.FC64
    assert_dense_same(got, p, v, exact=True, what=f"{desc}/{accum}/{masked}")


@pytest.mark.parametrize("desc", ["", "S", "RSC", "R"])
@pytest.mark.parametrize("accum", [None, "TIMES_FC32"])
@pytest.mark.parametrize("cfmt", ["bitmap", "full"])
def test_complex_masked_into_a_dense_C(gba, desc, accum, cfmt):
    n = 24
    t = Type.FC32
    rng = np.random.default_rng(25)
    A, B = _rand(n, n, 0.2, 26, t), _rand(n, n, 0.2, 27, t)
    vals = (rng.uniform(0.5, 2, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))).astype(np.complex64)
    C0 = Matrix.dense(vals, present=None if cfmt == "full" else rng.random((n, n)) < 0.5, by_row=False)
    M = _mask(n, n, 0.4, 28)
    dA, dB, dC, dM = A.to_device(), B.to_device(), C0.to_device(), M.to_device()
    gba.mxm(dC, dM, accum, "PLUS_TIMES_FC32", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, M, accum, "PLUS_TIMES_FC32", A, B, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{desc}/{accum}/{cfmt}")


def test_complex_accumulating_over_iterations(gba):
    """C += A*B three times: the result of one call (jumbled) is the C of the next."""
    n = 60
    rng = np.random.default_rng(29)
    def gi(seed):
        S = sp.random(n, n, density=0.06, random_state=np.random.default_rng(seed), format="csr").astype(np.complex128)
        S.data = rng.integers(1, 4, len(S.data)) + 1j * rng.integers(-3, 4, len(S.data))
        return Matrix.from_scipy(S, type=Type.FC64)
    A, B, C0 = gi(30), gi(31), gi(32)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    S = C0.to_scipy().toarray()
    P = A.to_scipy().toarray() @ B.to_scipy().toarray()
    for _ in range(3):
        gba.mxm(dC, None, "PLUS_FC64", "PLUS_TIMES_FC64", dA, dB, None)
        S = S + P
    assert np.array_equal(dC.to_host().to_scipy().toarray(), S)


@pytest.mark.parametrize("desc,accum", [("", None), ("RSC", None), ("S", "PLUS_FC64"), ("T0", "MINUS_FC64"), ("C", None)])
def test_complex_hypersparse_operands(gba, desc, accum):
    """Hypersparse A, B, mask and C are the same matrices: the result equals the sparse call's, and a C that
    came in hypersparse goes back hypersparse."""
    from graphblas_amd import Sparsity
    n = 64
    A, B = _rand(n, n, 0.02, 41), _rand(n, n, 0.02, 42, by_row=False)
    C0, M = _rand(n, n, 0.02, 43), _mask(n, n, 0.05, 44)
    Ah, Bh, Ch, Mh = A.to_hyper(), B.to_hyper(), C0.to_hyper(), M.to_hyper()
    assert Ah.nvec < n and Bh.nvec < n and Ch.nvec < n
    dC = Ch.to_device()
    gba.mxm(dC, Mh.to_device(), accum, "PLUS_TIMES_FC64", Ah.to_device(), Bh.to_device(), desc)
    p, v = oracle.mxm_spec(C0, M, accum, "PLUS_TIMES_FC64", A, B, desc)
    got = dC.to_host()
    assert got.sparsity == Sparsity.HYPERSPARSE
    assert_dense_same(got, p, v, exact=True, what=f"{desc}/{accum}")


@pytest.mark.parametrize("ta,tb", [(Type.FP64, Type.FC64), (Type.FC64, Type.INT32), (Type.FC32, Type.FC64), (Type.FP32, Type.BOOL),
                                   (Type.UINT8, Type.FC32)])
@pytest.mark.parametrize("mult", ["TIMES", "FIRST", "SECOND", "DIV"])
def test_operands_of_other_types_are_cast_to_complex(gba, ta, tb, mult):
    """A and B of another built-in type under a complex semiring are cast to its type first (Source/GB_mxm.c:84-99,
    Source/GB_casting.h): the real part takes the value, a complex value converts part by part."""
    n = 36
    zt = Type.FC32 if Type.FC32 in (ta, tb) and Type.FC64 not in (ta, tb) else Type.FC64

    def make(t, seed, by_row):
        rng = np.random.default_rng(seed)
        S = sp.random(n, n, density=0.15, random_state=rng, format="csr").astype(np.complex128)
        k = len(S.data)
        if t in (Type.FC32, Type.FC64):
            S.data = rng.uniform(0.5, 2.0, k) + 1j * rng.uniform(-2.0, 2.0, k)
        elif t == Type.BOOL:
            S.data = np.ones(k) + 0j
        elif t in (Type.FP32, Type.FP64):
            S.data = rng.uniform(0.5, 2.0, k) + 0j
        else:
            S.data = rng.integers(1, 9, k) + 0j
        if t not in (Type.FC32, Type.FC64):
            S = S.real
        return Matrix.from_scipy(S, type=t, by_row=by_row)

    A, B = make(ta, 51, True), make(tb, 52, False)
    C0 = Matrix.sparse_empty(n, n, zt)
    dC = C0.to_device()
    sr = f"PLUS_{mult}_{NAME[zt]}"
    rep = gba.mxm(dC, None, None, sr, A.to_device(), B.to_device(), None)
    assert rep.typecast == (1 if ta != zt else 0) | (2 if tb != zt else 0)
    p, v = oracle.mxm_spec(C0, None, None, sr, A, B)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{sr} {ta} {tb}")


@pytest.mark.parametrize("accum", [None, "PLUS_FC64"])
def test_complex_empty_operands_and_zero_dimensions(gba, accum):
    """No entries in A, in B, in both; a 0-by-n and an n-by-0 product; C with entries is left as the definition says."""
    n = 12
    full = _rand(n, n, 0.3, 61)
    empty = Matrix.sparse_empty(n, n, Type.FC64)
    for A, B in ((empty, full), (full, empty), (empty, empty)):
        for C0 in (Matrix.sparse_empty(n, n, Type.FC64), _rand(n, n, 0.2, 62)):
            dC = C0.to_device()
            gba.mxm(dC, None, accum, "PLUS_TIMES_FC64", A.to_device(), B.to_device(), None)
            p, v = oracle.mxm_spec(C0, None, accum, "PLUS_TIMES_FC64", A, B)
            assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{A.nvals}/{B.nvals}/{C0.nvals}/{accum}")
    for (m, k, q) in ((0, n, n), (n, 0, n), (n, n, 0)):
        A = Matrix.sparse_empty(m, k, Type.FC64)
        B = Matrix.sparse_empty(k, q, Type.FC64)
        dC = Matrix.sparse_empty(m, q, Type.FC64).to_device()
        gba.mxm(dC, None, accum, "PLUS_TIMES_FC64", A.to_device(), B.to_device(), None)
        got = dC.to_host()
        assert got.nvals == 0 and (got.nrows, got.ncols) == (m, q)


def _typed(n, t, seed, density=0.2, by_row=True):
    rng = np.random.default_rng(seed)
    S = sp.random(n, n, density=density, random_state=rng, format="csr").astype(np.complex128)
    k = len(S.data)
    if t in (Type.FC32, Type.FC64):
        S.data = rng.integers(-3, 4, k) + 0.5 + 1j * rng.integers(-3, 4, k)
    else:
        S.data = rng.integers(1, 6, k) + 0j
        S = S.real
    return Matrix.from_scipy(S, type=t, by_row=by_row)


@pytest.mark.parametrize("tc", [Type.FP64, Type.FP32, Type.INT32, Type.UINT8, Type.BOOL, Type.FC32])
@pytest.mark.parametrize("accum,desc,masked", [(None, "", False), ("PLUS", "", False), (None, "S", True), ("SECOND", "RSC", True), ("PLUS", "T0", True)])
def test_complex_product_into_a_C_of_another_type(gba, tc, accum, desc, masked):
    """C of another built-in type under a complex semiring: T is cast to C's type before the accumulator and the mask
    see it (Source/GB_mxm.c:84-99, Source/GB_casting.h:211-446: the real part; bool = either part non-zero)."""
    n = 30
    A, B = _typed(n, Type.FC64, 71), _typed(n, Type.FC64, 72, by_row=False)
    C0 = _typed(n, tc, 73, density=0.15)
    M = _mask(n, n, 0.4, 74) if masked else None
    tname = {v: k for k, v in vars(Type).items() if isinstance(v, int)}[tc]
    if tc == Type.BOOL and accum is not None:
        accum = {"PLUS": "LOR", "SECOND": "SECOND"}[accum]
    acc = None if accum is None else f"{accum}_{tname}"
    dC = C0.to_device()
    rep = gba.mxm(dC, M.to_device() if masked else None, acc, "PLUS_TIMES_FC64", A.to_device(), B.to_device(), desc)
    assert rep.typecast & 4
    p, v = oracle.mxm_spec(C0, M, acc, "PLUS_TIMES_FC64", A, B, desc)
    got = dC.to_host()
    assert got.type == tc
    assert_dense_same(got, p, v, exact=True, what=f"{tname}/{acc}/{desc}")


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP64", Type.FP64), ("MIN_PLUS_INT32", Type.INT32), ("LOR_LAND_BOOL", Type.BOOL)])
@pytest.mark.parametrize("accum,desc,masked", [(None, "", False), ("PLUS_FC64", "", False), (None, "S", True), ("MINUS_FC64", "RC", True)])
def test_real_product_into_a_complex_C(gba, sr, t, accum, desc, masked):
    n = 30
    A, B = _typed(n, t, 81), _typed(n, t, 82, by_row=False)
    C0 = _typed(n, Type.FC64, 83, density=0.15)
    M = _mask(n, n, 0.4, 84) if masked else None
    dC = C0.to_device()
    rep = gba.mxm(dC, M.to_device() if masked else None, accum, sr, A.to_device(), B.to_device(), desc)
    assert rep.typecast & 4
    p, v = oracle.mxm_spec(C0, M, accum, sr, A, B, desc)
    got = dC.to_host()
    assert got.type == Type.FC64
    assert_dense_same(got, p, v, exact=True, what=f"{sr}/{accum}/{desc}")
