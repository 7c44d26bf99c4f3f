"""GPU parity for the front end: gba_mxm / gba_mxv / gba_vxm against the definition
(Test/GB_spec_mxm.m restated in oracle.mxm_spec), and method selection against the
dispatch the reference was observed to take."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, DeviceMatrix, Type, Sparsity, Method, GrbError, Info, gen
from graphblas_amd.binding import NP_TYPES
from oracle import oracle
from util import assert_dense_same

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _rand(n, m, density, seed, t=Type.FP64, by_row=True, lo=1, hi=6):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csr")
    S.data = rng.integers(lo, hi, len(S.data)).astype(np.float64)
    return Matrix.from_scipy(S, type=t, by_row=by_row)


@pytest.mark.parametrize("desc", ["", "T0", "T1", "T0T1"])
@pytest.mark.parametrize("fmt", ["rrr", "ccc", "rcr", "crc"])
def test_mxm_all_transposes_and_formats(gba, desc, fmt):
    """C = A*B for every descriptor transpose and CSR/CSC mix (Source/GB_AxB_meta.c:207-263)."""
    n = 60
    A = _rand(n, n, 0.1, 1, by_row=fmt[1] == "r")
    B = _rand(n, n, 0.1, 2, by_row=fmt[2] == "r")
    C0 = Matrix.sparse_empty(n, n, Type.FP64, by_row=fmt[0] == "r")
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, None, "PLUS_TIMES_FP64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, None, None, "PLUS_TIMES_FP64", A, B, desc)
    got = dC.to_host()
    assert got.is_csc == C0.is_csc
    assert_dense_same(got, p, v, exact=True, what=f"{desc}/{fmt}")


@pytest.mark.parametrize("desc", ["S", "SC", "", "C", "RS", "RSC"])
@pytest.mark.parametrize("mkind", ["sparse", "bitmap"])
def test_mxm_masked_into_empty_C(gba, desc, mkind):
    n = 80
    A, B = _rand(n, n, 0.08, 3), _rand(n, n, 0.08, 4)
    rng = np.random.default_rng(5)
    if mkind == "sparse":
        M = _rand(n, n, 0.2, 6, t=Type.INT8, lo=0, hi=2)
    else:
        M = Matrix.dense(rng.integers(0, 2, (n, n)).astype(np.int8), present=rng.random((n, n)) < 0.6)
    C0 = Matrix.sparse_empty(n, n, Type.FP64)
    dA, dB, dM, dC = A.to_device(), B.to_device(), M.to_device(), C0.to_device()
    rep = gba.mxm(dC, dM, None, "PLUS_TIMES_FP64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, M, None, "PLUS_TIMES_FP64", A, B, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{desc}/{mkind}")
    assert rep.mask_applied


@pytest.mark.parametrize("desc", ["", "R", "S", "RS", "SC", "RSC", "C", "T0", "ST1"])
@pytest.mark.parametrize("accum", [None, "PLUS_FP64", "MIN_FP64", "SECOND_FP64"])
@pytest.mark.parametrize("masked", [False, True])
def test_mxm_into_a_sparse_C_that_holds_entries(gba, desc, accum, masked):
    """C<M> = accum (C, A*B) with a sparse, non-empty C: the write-back of GB_accum_mask
    (Source/GB_accum_mask.c), every combination of accum / mask / complement / replace."""
    if not masked and any(ch in desc for ch in "SC"):
        pytest.skip("mask modifiers without a mask")
    n = 70
    A, B = _rand(n, n, 0.08, 21), _rand(n, n, 0.08, 22)
    C0 = _rand(n, n, 0.15, 23)
    M = _rand(n, n, 0.3, 24, t=Type.INT8, lo=0, hi=2) if masked else None
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    dM = M.to_device() if masked else None
    gba.mxm(dC, dM, accum, "PLUS_TIMES_FP64", dA, dB, desc)
    p, v = oracle.mxm_spec(C0, M, accum, "PLUS_TIMES_FP64", A, B, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"{desc}/{accum}/{masked}")


def test_accumulating_over_iterations_keeps_C_consistent(gba):
    """C += A*B three times in a row (C jumbled after each round is waited on by the next)."""
    n = 120
    A, B = _rand(n, n, 0.05, 31), _rand(n, n, 0.05, 32)
    C0 = _rand(n, n, 0.02, 33)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    S = C0.to_scipy().toarray()
    P = A.to_scipy().toarray() @ B.to_scipy().toarray()
    for _ in range(3):
        gba.mxm(dC, None, "PLUS_FP64", "PLUS_TIMES_FP64", dA, dB, None)
        S = S + P
    got = dC.to_host()
    assert np.array_equal(got.to_scipy().toarray(), S)
    assert got.nvals == np.count_nonzero((C0.to_scipy().toarray() != 0) | (P != 0))


def test_mxv_into_a_sparse_vector_with_entries(gba):
    """w += A*u with a sparse w and a full u: T is bitmap, so w is made bitmap for the update."""
    n = 200
    A = _rand(n, n, 0.04, 41, t=Type.FP32)
    rng = np.random.default_rng(42)
    u = Matrix.full_vector(rng.integers(1, 5, n).astype(np.float32))
    idx = rng.choice(n, 20, replace=False)
    w0 = Matrix.sparse_vector(n, idx, rng.integers(1, 5, 20).astype(np.float32))
    dA, du, dw = A.to_device(), u.to_device(), w0.to_device()
    gba.mxv(dw, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, du, None)
    p, v = oracle.mxm_spec(w0, None, "PLUS_FP32", "PLUS_TIMES_FP32", A, u, "")
    assert_dense_same(dw.to_host(), p, v, exact=True)


def test_mxm_triangle_count_call(gba):
    """C<L,struct> = L*L' exactly as LAGraph issues it: GrB_mxm(C, L, NULL, PLUS_PAIR_INT64, L, L, GrB_DESC_ST1)."""
    G = gen.rmat(9, 16, seed=42, symmetric=True, drop_diagonal=True, values="ones", type=Type.INT64)
    L = gen.tril_strict(G)
    C0 = Matrix.sparse_empty(L.nrows, L.ncols, Type.INT64)
    dL, dC = L.to_device(), C0.to_device()
    rep = gba.mxm(dC, dL, None, "PLUS_PAIR_INT64", dL, dL, "ST1")
    assert Method.NAMES[rep.method] == "dot3"
    got = dC.to_host()
    p, v = oracle.mxm_spec(C0, L, None, "PLUS_PAIR_INT64", L, L, "ST1")
    assert_dense_same(got, p, v, exact=True)
    S = L.to_scipy(); S.data[:] = 1
    assert int(got.coo()[2].sum()) == int((S @ S.T).multiply(S).sum())


@pytest.mark.parametrize("call", ["mxv", "mxv_T0", "vxm", "vxm_T1"])
@pytest.mark.parametrize("wkind,accum", [("empty", None), ("full", "PLUS_FP32"), ("bitmap", "PLUS_FP32"), ("full", "MIN_FP32")])
def test_mxv_vxm_dense_vectors(gba, call, wkind, accum):
    """PageRank-shape products with a full input vector, every orientation the reference distinguishes."""
    n = 300
    A = _rand(n, n, 0.03, 7, t=Type.FP32)
    rng = np.random.default_rng(8)
    u = Matrix.full_vector(rng.integers(1, 5, n).astype(np.float32))
    if wkind == "empty":
        w0 = Matrix.sparse_vector(n, [], np.zeros(0, np.float32))
    elif wkind == "full":
        w0 = Matrix.full_vector(rng.integers(1, 5, n).astype(np.float32))
    else:
        w0 = Matrix.bitmap_vector(rng.integers(1, 5, n).astype(np.float32), rng.random(n) < 0.5)
    dA, du, dw = A.to_device(), u.to_device(), w0.to_device()
    sr = "PLUS_TIMES_FP32"
    if call.startswith("mxv"):
        desc = "T0" if call.endswith("T0") else ""
        rep = gba.mxv(dw, None, accum, sr, dA, du, desc)
        p, v = oracle.mxm_spec(w0, None, accum, sr, A, u, desc)
    else:
        desc = "T1" if call.endswith("T1") else ""
        rep = gba.vxm(dw, None, accum, sr, du, dA, desc)
        # w' = u'*A  ==  w = A'*u with the multiplier flipped (Source/GrB_vxm.c:56-72)
        p, v = oracle.mxm_spec(w0, None, accum, sr, A, u, "", flipxy=True, A_tran=(desc != "T1"), B_tran=False)
    assert_dense_same(dw.to_host(), p, v, exact=True, what=f"{call}/{wkind}/{accum}")
    if wkind == "full" and accum == "PLUS_FP32":
        assert rep.done_in_place
        assert Method.NAMES[rep.method] in ("dot4", "saxpy4")


def test_bfs_levels_match_reference_semantics(gba):
    """Level-synchronous BFS with q<!pi,struct,replace> = q'*A, ANY_SECONDI_INT64: frontier patterns are
    exact; every parent is a frontier neighbour (ANY may pick any of them)."""
    G = gen.rmat(10, 8, seed=42, symmetric=True, drop_diagonal=True, values="ones", type=Type.BOOL)
    n = G.nrows
    S = G.to_scipy().astype(bool).tocsr()
    dA = G.to_device()
    src = int(np.argmax(np.diff(G.p)))
    pi_b = np.zeros(n, np.int8); pi_x = np.zeros(n, np.int64)
    pi_b[src] = 1; pi_x[src] = src
    q = Matrix.sparse_vector(n, [src], np.array([src], np.int64))
    visited = {src}
    frontier = {src}
    methods = []
    for level in range(6):
        dq = q.to_device()
        dpi = Matrix.bitmap_vector(pi_x, pi_b).to_device()
        dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
        rep = gba.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")
        methods.append(Method.NAMES[rep.method])
        w = dw.to_host()
        r, _, parents = w.coo()
        want = set()
        for k in frontier:
            want.update(S.indices[S.indptr[k]:S.indptr[k + 1]].tolist())
        want -= visited
        assert set(r.tolist()) == want, f"level {level}"
        for v_, p_ in zip(r, parents):
            assert p_ in frontier and S[p_, v_]
        if not want:
            break
        visited |= want; frontier = want
        pi_b[r] = 1; pi_x[r] = parents
        # keep the frontier sparse or bitmap as the reference would (bitmap once it is large)
        if len(r) > n // 8:
            b = np.zeros(n, np.int8); x = np.zeros(n, np.int64); b[r] = 1; x[r] = parents
            q = Matrix.bitmap_vector(x, b)
        else:
            q = Matrix.sparse_vector(n, r, parents)
    assert "saxpy3" in methods


def test_sssp_relaxation_min_plus(gba):
    """d<-> = d min.+ A with accum MIN, d full and aliased with the input: bit-exact with integer weights."""
    A = gen.rmat(10, 8, seed=3, values="int", type=Type.FP64)
    n = A.nrows
    d = np.full(n, np.inf); d[0] = 0
    S = A.to_scipy().tocsr()
    dA = A.to_device()
    for it in range(4):
        dd = Matrix.full_vector(d).to_device()
        rep = gba.vxm(dd, None, "MIN_FP64", "MIN_PLUS_FP64", dd, dA, None)
        nd = d.copy()
        for k in np.nonzero(np.isfinite(d))[0]:
            cols = S.indices[S.indptr[k]:S.indptr[k + 1]]; w = S.data[S.indptr[k]:S.indptr[k + 1]]
            np.minimum.at(nd, cols, d[k] + w)
        got = dd.to_host()
        assert got.sparsity == Sparsity.FULL
        assert np.array_equal(got.x, nd), f"iteration {it}"
        assert Method.NAMES[rep.method] == "saxbit"        # aliasing rules out the in-place saxpy4
        d = nd


def test_errors_mirror_grb_info(gba):
    A = _rand(10, 12, 0.3, 9).to_device()
    C = DeviceMatrix.empty(10, 10, Type.FP64)
    with pytest.raises(GrbError) as e:
        gba.mxm(C, None, None, "PLUS_TIMES_FP64", A, A, None)
    assert e.value.info == Info.DIMENSION_MISMATCH
    gba.mxm(C, None, None, "PLUS_TIMES_FP64", A, A, "T1")       # 10x12 * 12x10 is fine
    B = _rand(10, 10, 0.3, 10, t=Type.INT32).to_device()
    rep = gba.mxm(C, None, None, "PLUS_TIMES_FP64", B, B, None)  # int32 operands are cast to the multiplier's fp64
    assert rep.typecast == 3
    rep = gba.mxm(C, None, None, gba.gba_semiring(gba.Op.PLUS, gba.Op.TIMES, Type.FC64, Type.FC64, Type.FC64), A, A, "T1")
    assert rep.typecast == 7                                     # real operands under a complex semiring into a real C
    with pytest.raises(GrbError) as e:
        gba.mxm(C, None, None, gba.gba_semiring(gba.Op.PLUS, gba.Op.USER, Type.FP64, Type.FP64, Type.FP64), A, A, "T1")
    assert e.value.info == Info.NOT_IMPLEMENTED                  # an operator without a built-in code stays on the CPU library


def _operand(kind, n, A, rng, t):
    dt = NP_TYPES[t]
    if kind == "full":
        return Matrix.full_vector(rng.integers(1, 5, n).astype(dt))
    if kind == "bitmap":
        return Matrix.bitmap_vector(rng.integers(1, 5, n).astype(dt), rng.random(n) < 0.5)
    if kind == "sparse_small":
        idx = rng.choice(n, max(n // 64, 1), replace=False)
        return Matrix.sparse_vector(n, idx, rng.integers(1, 5, len(idx)).astype(dt))
    if kind == "sparse_empty":
        return Matrix.sparse_vector(n, [], np.zeros(0, dt))
    raise ValueError(kind)


def test_method_selection_matches_observed_dispatch(gba):
    cases = json.load(open(os.path.join(GOLDEN, "dispatch_observed.json")))["cases"]
    rng = np.random.default_rng(11)
    for c in cases:
        sr = gba.semiring(c["semiring"])
        t = sr.xtype if c["semiring"].split("_")[1] not in ("SECONDI",) else Type.INT64
        zt = sr.ztype
        if c["call"] == "mxm_AA":
            A = gen.rmat(12, 16, seed=42).to_device()
            C = DeviceMatrix.empty(4096, 4096, Type.FP64)
            rep = gba.mxm_select(C, None, None, sr, A, A, c["desc"])
        elif c["call"] == "mxm_LLT":
            G = gen.rmat(12, 16, seed=42, symmetric=True, drop_diagonal=True, values="ones", type=Type.INT64)
            L = gen.tril_strict(G).to_device()
            C = DeviceMatrix.empty(4096, 4096, Type.INT64)
            rep = gba.mxm_select(C, L, None, sr, L, L, c["desc"])
        else:
            G = gen.rmat(12, 16, seed=42, symmetric=True, type=t if t != Type.INT64 else Type.INT64)
            n = G.nrows
            A = G.to_device()
            w = _operand(c["w"], n, G, rng, zt).to_device()
            u = w if c["u"] == "alias_w" else _operand(c["u"], n, G, rng, t).to_device()
            mask = _operand(c["mask"], n, G, rng, Type.INT64).to_device() if c.get("mask") else None
            fn = gba.lib.gba_mxv if c["call"] == "mxv" else gba.lib.gba_vxm
            # select-only: same decision code, reached through gba_mxm_select with the operand order
            # and flags GrB_mxv / GrB_vxm pass to GB_mxm (Source/GrB_mxv.c:57-66, GrB_vxm.c:63-72)
            from graphblas_amd.host import _call3
            rep = _select_vec(gba, c["call"], w, mask, c["accum"], sr, A, u, c["desc"])
        assert Method.NAMES[rep.method] == c["method"], (c["name"], Method.NAMES[rep.method], c["burble"])
        assert rep.done_in_place == c["in_place"], c["name"]
        if "swapped" in c:
            assert rep.swapped == c["swapped"], c["name"]


def _select_vec(gba, call, w, mask, accum, sr, A, u, desc):
    """Run the real mxv/vxm on copies (selection and execution share one code path) and report the method."""
    wc = w.to_host().to_device()
    if call == "mxv":
        return gba.mxv(wc, mask, accum, sr, A, u if u is not w else wc, desc)
    return gba.vxm(wc, mask, accum, sr, u if u is not w else wc, A, desc)


@pytest.mark.parametrize("case", ["T0", "cast", "host"])
def test_row_slice_views_that_keep_absolute_offsets(gba, case):
    """A row slice of a sparse matrix is a view whose p keeps the absolute positions (p[0] != 0).  The paths that
    walk the entries of an operand directly -- the explicit transpose behind a descriptor transpose, the
    typecast of an operand, the staging of a host view -- must take entries p[0] .. p[nvec), not 0 .. nvals."""
    n, lo, hi = 90, 31, 77
    t = Type.FP32 if case == "cast" else Type.FP64
    A = _rand(n, n, 0.15, 11, t=t)
    B = _rand(hi - lo if case == "T0" else n, n, 0.12, 12)
    S = A.to_scipy()[lo:hi]
    Asub = Matrix.from_scipy(S.astype(NP_TYPES[t]), type=t)
    desc = "T0" if case == "T0" else ""
    C0 = Matrix.sparse_empty(n if case == "T0" else hi - lo, n, Type.FP64)
    p, v = oracle.mxm_spec(C0, None, None, "PLUS_TIMES_FP64", Asub, B, desc)
    dB, dC = B.to_device(), C0.to_device()
    if case == "host":
        import ctypes

        class HostView:                                     # gba_matrix over host arrays (mem = host), p not rebased
            def __init__(self, M):
                self.keep = []
                self.v = M.view(self.keep)

            def ref(self):
                return ctypes.byref(self.v)

        hv = HostView(Matrix(A.vlen, hi - lo, t, A.sparsity, A.is_csc, p=A.p[lo:hi + 1], i=A.i, x=A.x,
                             nvals=int(A.p[hi] - A.p[lo])))
        gba.mxm(dC, None, None, "PLUS_TIMES_FP64", hv, dB, desc)
    else:
        dA = A.to_device()
        gba.mxm(dC, None, None, "PLUS_TIMES_FP64", dA.row_slice(lo, hi), dB, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=case)


def test_output_that_is_its_own_operand_does_not_leak(gba):
    """C = C*B, and q<!v> = q*A of a traversal: the arrays C held before the call belong to nobody else and
    must go back to the pool (Source/GB_mxm.c:48 allows the aliasing)."""
    import ctypes
    n = 300
    B = _rand(n, n, 0.02, 21)
    Cm = _rand(n, n, 0.02, 22)
    dB, dC = B.to_device(), Cm.to_device()
    want = Cm.to_scipy()
    live = []
    for _ in range(6):
        gba.mxm(dC, None, None, "PLUS_TIMES_FP64", dC, dB, "")
        want = want @ B.to_scipy()
        idle, lv = ctypes.c_size_t(0), ctypes.c_size_t(0)
        gba.lib.gba_pool_stats(ctypes.byref(idle), ctypes.byref(lv))
        live.append(lv.value)
    got = dC.to_host().to_scipy()
    assert abs(got - want).max() <= 1e-9 * abs(want).max()
    # the product fills in, so C itself grows; what must not grow is one old copy of C per call
    sizes = np.diff(live)
    assert live[-1] - live[0] < 4 * (got.nnz * 16 + (n + 1) * 8), (live, got.nnz)


def test_jumbled_results_feed_the_next_call(gba):
    """Every sparse product comes back jumbled (as the reference leaves it).  Used as the mask or as a dot-product
    operand of the next call it is sorted inside the call, on a copy (GB_MATRIX_WAIT, Source/GB_mxm.c:105-116;
    GB_MATRIX_WAIT_IF_JUMBLED, Source/GB_AxB_dot.c:78-87) -- the caller need not wait on it first."""
    n = 120
    A, B = _rand(n, n, 0.06, 31), _rand(n, n, 0.06, 32)
    dA, dB = A.to_device(), B.to_device()
    d1 = DeviceMatrix.empty(n, n, Type.FP64)
    gba.mxm(d1, None, None, "PLUS_TIMES_FP64", dA, dA, "")
    C1 = d1.to_host()
    assert C1.jumbled
    # jumbled mask of a saxpy product
    C0 = Matrix.sparse_empty(n, n, Type.FP64)
    d2 = C0.to_device()
    gba.mxm(d2, d1, None, "PLUS_TIMES_FP64", dB, dB, "S")
    p, v = oracle.mxm_spec(C0, C1, None, "PLUS_TIMES_FP64", B, B, "S")
    assert_dense_same(d2.to_host(), p, v, exact=True, what="jumbled mask")
    # jumbled operand of a masked dot product (C<M,struct> = A*B' takes dot3, like the triangle count)
    M = _rand(n, n, 0.3, 33, t=Type.INT8, lo=1, hi=2)
    d3 = C0.to_device()
    rep = gba.mxm(d3, M.to_device(), None, "PLUS_TIMES_FP64", dB, d1, "ST1")
    assert Method.NAMES[rep.method] == "dot3"
    p, v = oracle.mxm_spec(C0, M, None, "PLUS_TIMES_FP64", B, C1, "ST1")
    assert_dense_same(d3.to_host(), p, v, exact=True, what="jumbled dot operand")
    assert d1.to_host().jumbled                              # the caller's matrix is untouched


@pytest.mark.parametrize("sr,t,b_iso,want_iso", [("ANY_PAIR_BOOL", Type.BOOL, False, True), ("MIN_SECOND_FP64", Type.FP64, True, True),
                                                 ("MIN_SECOND_FP64", Type.FP64, False, False), ("PLUS_PAIR_INT64", Type.INT64, False, False),
                                                 ("LOR_LAND_BOOL", Type.BOOL, True, False)])
def test_iso_valued_products_come_back_iso(gba, sr, t, b_iso, want_iso):
    """C = A*B replaces an empty C: when the semiring and the operands make every entry equal
    (Source/GB_AxB_iso.c:117-330: PAIR under a monoid that keeps 1, SECOND/FIRST of an iso operand under
    ANY/MIN/MAX/LOR/LAND/BOR/BAND, two iso operands) C is handed over iso with one stored value."""
    n = 70
    A, B = _rand(n, n, 0.1, 31, t=t, lo=1, hi=2), _rand(n, n, 0.1, 32, t=t, lo=1, hi=4)
    if b_iso:
        B = Matrix(B.vlen, B.vdim, t, Sparsity.SPARSE, B.is_csc, p=B.p, i=B.i, x=np.array([3], NP_TYPES[t]), iso=True, nvals=B.nvals)
    C0 = Matrix.sparse_empty(n, n, t)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, None, sr, dA, dB, "")
    got = dC.to_host()
    assert got.iso == want_iso
    if want_iso:
        assert len(got.x) == 1
    p, v = oracle.mxm_spec(C0, None, None, sr, A, B, "")
    assert_dense_same(got, p, v, exact=True, what=sr)
    for d in (dA, dB, dC):
        d.free()


@pytest.mark.parametrize("desc", ["", "S", "C", "RS"])
def test_costly_sparse_mask_is_discarded_and_applied_afterwards(gba, desc):
    """A product of a few dozen multiply-adds under a sparse mask of thousands of entries: the reference drops
    the mask from the multiply (flops < 0.01 * mask work, Source/GB_AxB_saxpy3_slice_balanced.c:22,309-403) and
    applies it in the write-back; the result is the same, the report says the mask was not applied."""
    n = 600
    A, B = _rand(n, n, 0.0004, 41), _rand(n, n, 0.0004, 42)
    A = Matrix.from_scipy((A.to_scipy() + sp.diags((np.arange(n) % 7 == 0).astype(np.float64))).tocsr(), type=Type.FP64)
    B = Matrix.from_scipy((B.to_scipy() + sp.diags((np.arange(n) % 5 == 0).astype(np.float64))).tocsr(), type=Type.FP64)
    M = _rand(n, n, 0.9, 43, t=Type.INT8, lo=0, hi=2)
    C0 = _rand(n, n, 0.01, 44)
    dA, dB, dM, dC = A.to_device(), B.to_device(), M.to_device(), C0.to_device()
    rep = gba.mxm(dC, dM, None, "PLUS_TIMES_FP64", dA, dB, desc)
    assert Method.NAMES[rep.method] == "saxpy3" and not rep.mask_applied
    p, v = oracle.mxm_spec(C0, M, None, "PLUS_TIMES_FP64", A, B, desc)
    assert_dense_same(dC.to_host(), p, v, exact=True, what=desc)
    for d in (dA, dB, dM, dC):
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("failure", ["dimension", "accum_op"])
def test_a_hypersparse_C_is_untouched_by_a_call_that_fails(gba, failure):
    """A call that fails validation (or is declined) must hand a hypersparse C back as it came -- hypersparse, the
    same list of vectors, the same entries -- so that a caller which falls back to the CPU library finds its C
    unchanged (the front end expands a hypersparse C to a pointer per vector before it works on it)."""
    g = gba
    rng = np.random.default_rng(5)
    n = 300
    S = sp.random(n, n, density=0.002, format="csr", random_state=7, dtype=np.float64)
    C0 = Matrix.from_scipy(S, type=Type.FP64).to_hyper()
    assert C0.nvec < n // 2
    A = Matrix.from_scipy(sp.random(n + (1 if failure == "dimension" else 0), n, density=0.02, format="csr", random_state=8), type=Type.FP64)
    B = Matrix.from_scipy(sp.random(n, n, density=0.02, format="csr", random_state=9), type=Type.FP64)
    dC, dA, dB = C0.to_device(), A.to_device(), B.to_device()
    accum = "POW_FP64" if failure == "accum_op" else None       # an accumulator the write-back has no code for
    with pytest.raises(GrbError) as err:
        g.mxm(dC, None, accum, "PLUS_TIMES_FP64", dA, dB, None)
    assert err.value.info == (Info.DIMENSION_MISMATCH if failure == "dimension" else Info.NOT_IMPLEMENTED)
    back = dC.to_host()
    assert back.sparsity == Sparsity.HYPERSPARSE and back.nvec == C0.nvec
    assert np.array_equal(back.h, C0.h) and np.array_equal(back.p, C0.p)
    assert np.array_equal(back.i, C0.i) and np.array_equal(back.x, C0.x)
    for d in (dC, dA, dB):
        d.free()


@pytest.mark.parametrize("ckind", ["sparse", "full", "bitmap"])
@pytest.mark.parametrize("accum,masked", [("PLUS_FP64", False), ("MIN_FP64", True), (None, True)])
def test_an_iso_valued_C_comes_in(gba, ckind, accum, masked):
    """C<M> = accum (C, A*B) where C arrives iso-valued (one stored value for all its entries, as an earlier iso
    product or GrB_assign of a scalar leaves it; Source/GB_AxB_iso.c:79, Source/GB_accum_mask.c): the result is that
    of the same C with its values written out."""
    n = 60
    A, B = _rand(n, n, 0.1, 41), _rand(n, n, 0.1, 42)
    M = _rand(n, n, 0.3, 43) if masked else None
    if ckind == "sparse":
        P = _rand(n, n, 0.2, 44)
        C_iso = Matrix(P.vlen, P.vdim, Type.FP64, Sparsity.SPARSE, P.is_csc, p=P.p, i=P.i, x=np.array([2.5]), iso=True, nvals=P.nvals)
        C_out = Matrix(P.vlen, P.vdim, Type.FP64, Sparsity.SPARSE, P.is_csc, p=P.p, i=P.i, x=np.full(P.nvals, 2.5))
    elif ckind == "full":
        C_iso = Matrix(n, n, Type.FP64, Sparsity.FULL, True, x=np.array([2.5]), iso=True)
        C_out = Matrix(n, n, Type.FP64, Sparsity.FULL, True, x=np.full(n * n, 2.5))
    else:
        b = (np.random.default_rng(45).random(n * n) < 0.4).astype(np.int8)
        C_iso = Matrix(n, n, Type.FP64, Sparsity.BITMAP, True, x=np.array([2.5]), b=b, iso=True)
        C_out = Matrix(n, n, Type.FP64, Sparsity.BITMAP, True, x=np.full(n * n, 2.5), b=b)
    dA, dB, dC = A.to_device(), B.to_device(), C_iso.to_device()
    dM = M.to_device() if masked else None
    gba.mxm(dC, dM, accum, "PLUS_TIMES_FP64", dA, dB, "")
    p, v = oracle.mxm_spec(C_out, M, accum, "PLUS_TIMES_FP64", A, B, "")
    assert_dense_same(dC.to_host(), p, v, exact=True, what=f"iso C {ckind}/{accum}/{masked}")
    for d in (dA, dB, dC, dM):
        if d is not None:
            d.free()


def test_hyper_switch_conforms_the_result(gba):
    """The reference ends GB_mxm by re-conforming C (Source/GB_conform_hyper.c): sparse -> hypersparse when the
    non-empty vectors are at most hyper_switch * vdim, hypersparse -> sparse when they are more than twice that.  A C
    that carries no hyper_switch keeps the form it came in with."""
    n = 400
    rng = np.random.default_rng(5)
    few = np.zeros((n, n)); few[:, [7, 90, 301]] = (rng.random((n, 3)) < 0.2) * rng.integers(1, 5, (n, 3))
    B_few = Matrix.from_scipy(sp.csc_matrix(few), type=Type.FP64, by_row=False)          # three non-empty columns
    A, B_many = _rand(n, n, 0.05, 51), _rand(n, n, 0.05, 52)
    dA, dBf, dBm = A.to_device(), B_few.to_device(), B_many.to_device()

    def product(dB, B, C0, switch):
        dC = C0.to_device()
        dC.view.hyper_switch = switch
        gba.mxm(dC, None, None, "PLUS_TIMES_FP64", dA, dB, "")
        got = dC.to_host()
        p, v = oracle.mxm_spec(Matrix.sparse_empty(n, n, Type.FP64, by_row=False), None, None, "PLUS_TIMES_FP64", A, B, "")
        assert_dense_same(got, p, v, exact=True, what=f"hyper_switch {switch}")
        dC.free()
        return got

    empty = Matrix.sparse_empty(n, n, Type.FP64, by_row=False)
    got = product(dBf, B_few, empty, 0.0625)
    assert got.sparsity == Sparsity.HYPERSPARSE and list(got.h) == [7, 90, 301]          # 3 <= 0.0625 * 400
    assert product(dBf, B_few, empty, 0.0).sparsity == Sparsity.SPARSE                   # never asked for
    assert product(dBm, B_many, empty, 0.0625).sparsity == Sparsity.SPARSE               # 400 non-empty vectors
    assert product(dBm, B_many, empty.to_hyper(), 0.0625).sparsity == Sparsity.SPARSE    # came in hypersparse: more than 2 * 25
    assert product(dBm, B_many, empty.to_hyper(), 0.0).sparsity == Sparsity.HYPERSPARSE  # keeps the form it came in with
    assert product(dBf, B_few, empty.to_hyper(), 0.0625).sparsity == Sparsity.HYPERSPARSE
    for d in (dA, dBf, dBm):
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("by_row", [False, True])
@pytest.mark.parametrize("accum", [None, "PLUS_FP64"])
def test_a_hypersparse_inner_dimension_of_2_40(gba, by_row, accum):
    """C = A*B with an inner dimension of 2^40 that the hypersparse operand lists 300 vectors of: nothing of that
    size can be built, the reference looks vectors up through the hyper-hash (Source/Shared/GB_hyper_hash_lookup.h:16);
    here the dimension is renamed to the list.  The other operand has entries at indices the list does not hold (they
    meet empty vectors).  Expected: the product over the listed indices, formed on the host with scipy."""
    rng = np.random.default_rng(123)
    big, m, n, nk = 2**40, 70, 50, 300
    listed = np.sort(rng.choice(big, nk, replace=False).astype(np.int64))
    other = np.setdiff1d(rng.choice(big, 200).astype(np.int64), listed)
    Xs = sp.random(m, nk, density=0.05, random_state=rng, format="csc"); Xs.data = rng.integers(1, 5, len(Xs.data)).astype(np.float64)
    Xs = Xs[:, np.diff(Xs.indptr) > 0] if False else Xs
    keep = np.diff(Xs.indptr) > 0                                    # h lists non-empty vectors only
    Xs, lst = Xs[:, keep].tocsc(), listed[keep]
    Ys = sp.random(len(lst) + len(other), n, density=0.08, random_state=rng, format="csc")
    Ys.data = rng.integers(1, 5, len(Ys.data)).astype(np.float64)
    allk = np.concatenate([lst, other]); order = np.argsort(allk, kind="stable")
    Ysorted = Ys.tocsr()[order].tocsc(); Ysorted.sort_indices()
    # the listed operand: vectors over the inner dimension; the other one: indices over it
    X = Matrix(m, big, Type.FP64, Sparsity.HYPERSPARSE, not by_row, p=Xs.indptr.astype(np.int64), i=Xs.indices.astype(np.int64),
               x=Xs.data, h=lst)
    Y = Matrix(big, n, Type.FP64, Sparsity.SPARSE, not by_row, p=Ysorted.indptr.astype(np.int64),
               i=allk[order][Ysorted.indices].astype(np.int64), x=Ysorted.data)
    want = (Xs @ Ys.tocsr()[:len(lst)]).toarray()                    # m x n: the listed part of the product
    C0s = sp.random(m, n, density=0.1, random_state=rng, format="csc"); C0s.data = rng.integers(1, 5, len(C0s.data)).astype(np.float64)
    if by_row:
        # the same arrays read by row: X is nk... the product is Y' * X' = (X*Y)', stored by row
        A, B = Y, X                                                  # A: n x 2^40 by row (long rows), B: 2^40 x m by row, hypersparse
        C0 = Matrix.from_scipy(C0s.T.tocsr(), type=Type.FP64)        # n x m by row
        want = want.T
    else:
        A, B = X, Y
        C0 = Matrix.from_scipy(C0s, type=Type.FP64, by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, accum, "PLUS_TIMES_FP64", dA, dB, None)
    got = dC.to_host().to_scipy().toarray()
    c0 = C0.to_scipy().toarray()
    assert np.array_equal(got, c0 + want if accum else want)
    for d in (dA, dB, dC):
        d.free()
