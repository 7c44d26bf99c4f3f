"""The drop-in layer EXECUTED: shim/grb_shim.c (GrB_mxm / GrB_vxm / GrB_mxv with the reference's prototypes) linked
against tests/fake_grb.c -- a test double of the CPU library's public interface, NOT the reference -- and against
libgraphblas_amd.so.  The calls are the ones the benchmark configurations make, with the handle aliasing they have
as a caller writes them: C = A*A, C<L> = L*L', q<!pi> = q'*A with q in and out, d = d min.+ A.  Results are compared
with the oracle / scipy; a call the device declines must reach the double's GrB_mxm stub with every handle intact.

Both shared objects are built by __graft_entry__.build() where the reference's header exists, and travel with the
tree (tests/build/)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "build")
pytestmark = pytest.mark.gpu

FP32, FP64, INT64 = 10, 11, 8               # GrB_Type_Code (Include/GraphBLAS.h:4081-4094)
NPT = {FP32: np.float32, FP64: np.float64, INT64: np.int64}
HYPER, SPARSE, BITMAP, FULL = 1, 2, 4, 8


@pytest.fixture(scope="module")
def libs(gba):
    shim_path = os.path.join(BUILD, "libgrb_shim_test.so")
    assert os.path.exists(shim_path), "tests/build/libgrb_shim_test.so is missing: run __graft_entry__.build() where /root/reference exists"
    shim = C.CDLL(shim_path, mode=C.RTLD_GLOBAL)            # loads libfake_grb.so behind it: RTLD_NEXT finds the stubs there
    fake = C.CDLL(os.path.join(BUILD, "libfake_grb.so"), mode=C.RTLD_GLOBAL)
    vp, i64 = C.c_void_p, C.c_int64
    fake.fake_matrix_sparse.restype = vp
    fake.fake_matrix_sparse.argtypes = [C.c_int, i64, i64, C.c_int, C.c_int, i64, vp, vp, vp, vp, C.c_int, C.c_int]
    fake.fake_matrix_dense.restype = vp
    fake.fake_matrix_dense.argtypes = [C.c_int, i64, i64, C.c_int, vp, vp, i64]
    fake.fake_vector_sparse.restype = vp
    fake.fake_vector_sparse.argtypes = [C.c_int, i64, i64, vp, vp]
    fake.fake_vector_dense.restype = vp
    fake.fake_vector_dense.argtypes = [C.c_int, i64, vp, vp, i64]
    for f in (fake.fake_matrix_look, fake.fake_vector_look):
        f.restype = None
        f.argtypes = [vp, C.POINTER(i64), C.POINTER(vp)]
    fake.fake_matrix_free.argtypes = [vp]
    fake.fake_vector_free.argtypes = [vp]
    fake.fake_set_allocator.argtypes = [vp, vp, vp, vp]
    for f in (shim.GrB_mxm, shim.GrB_vxm, shim.GrB_mxv):
        f.restype = C.c_int
        f.argtypes = [vp] * 7
    return shim, fake


def obj(fake, name):
    return C.c_void_p.in_dll(fake, name).value


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_csr(fake, S, code, by_row=True):
    S = (S.tocsr() if by_row else S.tocsc()).astype(NPT[code])
    S.sort_indices()
    p, i, x = S.indptr.astype(np.int64), S.indices.astype(np.int64), np.ascontiguousarray(S.data)
    return fake.fake_matrix_sparse(code, S.shape[0], S.shape[1], int(by_row), 0, 0, ptr(p), None, ptr(i), ptr(x), 0, 0)


def empty_matrix(fake, n, m, code, by_row=True):
    p = np.zeros((n if by_row else m) + 1, np.int64)
    return fake.fake_matrix_sparse(code, n, m, int(by_row), 0, 0, ptr(p), None, ptr(np.zeros(1, np.int64)),
                                   ptr(np.zeros(1, NPT[code])), 0, 0)


def look(fake, h, code, vector=False):
    """Contents of a handle as (fields, scipy matrix or dense arrays)."""
    fields = (C.c_int64 * 9)()
    arrays = (C.c_void_p * 5)()
    (fake.fake_vector_look if vector else fake.fake_matrix_look)(h, fields, arrays)
    sparsity, by_row, iso, jumbled, held, nvals, nvec, nrows, ncols = list(fields)
    dt = NPT[code]

    def arr(k, n, t):
        if not arrays[k] or n == 0:
            return np.zeros(0, t)
        return np.ctypeslib.as_array(C.cast(arrays[k], C.POINTER(np.ctypeslib.as_ctypes_type(t))), shape=(n,)).copy()
    out = {"sparsity": sparsity, "by_row": bool(by_row), "iso": bool(iso), "jumbled": bool(jumbled), "held": bool(held),
           "nvals": nvals, "shape": (nrows, ncols)}
    if vector:
        if sparsity == SPARSE:
            out["i"], out["x"] = arr(2, nvals, np.int64), arr(3, nvals, dt)
        else:
            out["x"] = arr(3, nrows, dt)
            out["b"] = arr(4, nrows, np.int8) if sparsity == BITMAP else np.ones(nrows, np.int8)
        return out
    if sparsity in (SPARSE, HYPER):
        nv = nrows if by_row else ncols
        if sparsity == HYPER:                                   # the listed vectors only: spread them out again
            hp, hh = arr(0, nvec + 1, np.int64), arr(1, nvec, np.int64)
            counts = np.zeros(nv, np.int64); counts[hh] = np.diff(hp)
            p = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            out["h"] = hh
        else:
            p = arr(0, nv + 1, np.int64)
        i, x = arr(2, int(p[-1]), np.int64), arr(3, int(p[-1]), dt)
        M = (sp.csr_matrix if by_row else sp.csc_matrix)((x, i, p), shape=(nrows, ncols))
        M.sort_indices()
        out["S"] = M.tocsr()
    return out


def rmat_small(scale=8, ef=8, seed=3, values="int"):
    from graphblas_amd import gen
    return gen.rmat(scale, ef, seed=seed, values=values).to_scipy().tocsr()


def test_mxm_with_the_same_handle_as_both_inputs(libs):
    """GrB_mxm (C, NULL, NULL, GrB_PLUS_TIMES_SEMIRING_FP64, A, A, NULL): configurations 1 and 3 as a caller writes them."""
    shim, fake = libs
    S = rmat_small()
    A = make_csr(fake, S, FP64)
    Cm = empty_matrix(fake, S.shape[0], S.shape[1], FP64)
    before = fake.fake_cpu_calls(0)
    info = shim.GrB_mxm(Cm, None, None, obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP64"), A, A, None)
    assert info == 0
    assert fake.fake_cpu_calls(0) == before, "the call fell through to the CPU library's stub"
    got = look(fake, Cm, FP64)
    want = (S @ S).tocsr()
    want.sort_indices()
    assert got["held"] and got["by_row"]
    assert (got["S"] != want).nnz == 0 and got["S"].nnz == want.nnz        # integer-valued doubles: exact
    a = look(fake, A, FP64)
    assert a["held"] and (a["S"] != S).nnz == 0, "the input handle did not get its arrays back"
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)


def test_masked_dot_with_mask_and_both_inputs_aliased(libs):
    """GrB_mxm (C, L, NULL, GxB_PLUS_PAIR_INT64, L, L, GrB_DESC_ST1): the triangle count, one handle in three roles."""
    shim, fake = libs
    S = rmat_small(values="ones")
    Sym = ((S + S.T) != 0).astype(np.int64)
    L = sp.tril(Sym, k=-1).tocsr()
    Lh = make_csr(fake, L, INT64)
    Cm = empty_matrix(fake, L.shape[0], L.shape[1], INT64)
    before = fake.fake_cpu_calls(0)
    info = shim.GrB_mxm(Cm, Lh, None, obj(fake, "GxB_PLUS_PAIR_INT64"), Lh, Lh, obj(fake, "GrB_DESC_ST1"))
    assert info == 0 and fake.fake_cpu_calls(0) == before
    got = look(fake, Cm, INT64)["S"]
    want = (L @ L.T).multiply(L != 0).tocsr()
    want.eliminate_zeros(); want.sort_indices()
    got.eliminate_zeros()
    assert (got != want).nnz == 0
    assert int(got.sum()) == int(want.sum())                # the triangle count
    lh = look(fake, Lh, INT64)
    assert lh["held"] and (lh["S"] != L).nnz == 0
    fake.fake_matrix_free(Lh); fake.fake_matrix_free(Cm)


def test_frontier_step_with_the_frontier_as_input_and_output(libs):
    """GrB_vxm (q, pi, NULL, GxB_ANY_SECONDI_INT64, q, A, GrB_DESC_RSC): q is read and replaced, pi is a bitmap mask."""
    shim, fake = libs
    S = rmat_small(values="ones")
    A = ((S + S.T) != 0).astype(np.int64).tocsr()
    n = A.shape[0]
    Ah = make_csr(fake, A, INT64)
    frontier = np.array([0, 1, 5], np.int64)
    visited = np.zeros(n, np.int8); visited[frontier] = 1
    pi = fake.fake_vector_dense(INT64, n, ptr(visited), ptr(np.zeros(n, np.int64)), int(visited.sum()))
    q = fake.fake_vector_sparse(INT64, n, len(frontier), ptr(frontier), ptr(frontier.copy()))
    before = fake.fake_cpu_calls(1)
    info = shim.GrB_vxm(q, pi, None, obj(fake, "GxB_ANY_SECONDI_INT64"), q, Ah, obj(fake, "GrB_DESC_RSC"))
    assert info == 0 and fake.fake_cpu_calls(1) == before
    got = look(fake, q, INT64, vector=True)
    reach = np.unique(A[frontier].indices)
    want = np.setdiff1d(reach, frontier)
    if got["sparsity"] == SPARSE:
        idx, par = got["i"], got["x"]
        order = np.argsort(idx); idx, par = idx[order], par[order]
    else:
        idx = np.nonzero(got["b"])[0]; par = got["x"][idx]
    assert np.array_equal(idx, want)
    for v, p in zip(idx, par):                              # ANY: any frontier vertex adjacent to v is a valid parent
        assert p in frontier and A[p, v] != 0
    assert look(fake, pi, INT64, vector=True)["held"]
    fake.fake_matrix_free(Ah); fake.fake_vector_free(q); fake.fake_vector_free(pi)


def test_relaxation_with_the_distance_vector_as_input_and_output(libs):
    """GrB_vxm (d, NULL, GrB_MIN_FP64, GrB_MIN_PLUS_SEMIRING_FP64, d, A, NULL): d = min (d, d min.+ A), d full."""
    shim, fake = libs
    S = rmat_small(values="int")
    A = S.tocsr().astype(np.float64)
    n = A.shape[0]
    Ah = make_csr(fake, A, FP64)
    d0 = np.full(n, 1e30); d0[0] = 0.0; d0[3] = 2.0
    d = fake.fake_vector_dense(FP64, n, None, ptr(d0.copy()), n)
    before = fake.fake_cpu_calls(1)
    info = shim.GrB_vxm(d, None, obj(fake, "GrB_MIN_FP64"), obj(fake, "GrB_MIN_PLUS_SEMIRING_FP64"), d, Ah, None)
    assert info == 0 and fake.fake_cpu_calls(1) == before
    got = look(fake, d, FP64, vector=True)
    Ac = A.tocoo()
    want = d0.copy()
    np.minimum.at(want, Ac.col, d0[Ac.row] + Ac.data)
    assert got["sparsity"] == FULL and np.array_equal(got["x"], want)
    fake.fake_matrix_free(Ah); fake.fake_vector_free(d)


def test_an_accumulator_of_another_type_than_C_is_taken(libs):
    """GrB_mxm (C, NULL, GrB_PLUS_INT64, GrB_PLUS_TIMES_SEMIRING_FP64, A, A, NULL) with an FP64 C: where C and the
    product both have an entry the sum is made in int64 (both truncated), the other entries keep their fractions."""
    shim, fake = libs
    S = rmat_small(values="int").astype(np.float64)
    S.data += 0.5                                                    # halves: products and sums stay exact in fp64
    A = make_csr(fake, S, FP64)
    C0 = sp.random(S.shape[0], S.shape[1], density=0.05, format="csr", random_state=4, dtype=np.float64)
    C0.data = np.round(C0.data * 64) / 8 + 0.125
    Cm = make_csr(fake, C0, FP64)
    before = fake.fake_cpu_calls(0)
    info = shim.GrB_mxm(Cm, None, obj(fake, "GrB_PLUS_INT64"), obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP64"), A, A, None)
    assert info == 0 and fake.fake_cpu_calls(0) == before, "the call went to the CPU library"
    got = look(fake, Cm, FP64)["S"]
    T = (S @ S).tocsr()
    both = C0.multiply(T != 0).astype(bool)                          # pattern of the intersection
    c, t = C0.toarray(), T.toarray()
    want = np.where(both.toarray(), np.trunc(c) + np.trunc(t), np.where(c != 0, c, t))
    pattern = (c != 0) | ((S != 0).astype(np.int64) @ (S != 0).astype(np.int64)).toarray().astype(bool)
    assert np.array_equal(got.toarray(), want) and got.nnz == int(pattern.sum())
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)


def test_a_declined_call_reaches_the_cpu_library_with_its_handles_intact(libs):
    """A semiring the device path has no code for (PLUS_POW): the stub behind dlsym (RTLD_NEXT) is reached, and C, A
    hold exactly what they held."""
    shim, fake = libs
    S = rmat_small()
    A = make_csr(fake, S, FP64)
    C0 = sp.random(S.shape[0], S.shape[1], density=0.01, format="csr", random_state=1, dtype=np.float64)
    Cm = make_csr(fake, C0, FP64)
    before = fake.fake_cpu_calls(0)
    info = shim.GrB_mxm(Cm, None, None, obj(fake, "GxB_PLUS_POW_FP64"), A, A, None)
    assert info == 0
    assert fake.fake_cpu_calls(0) == before + 1, "the declined call did not reach the next GrB_mxm"
    assert fake.fake_cpu_saw_intact() == 1, "a handle was still empty when the CPU library was called"
    c, a = look(fake, Cm, FP64), look(fake, A, FP64)
    C0.sort_indices()
    assert c["held"] and (c["S"] != C0).nnz == 0 and c["S"].nnz == C0.nnz
    assert a["held"] and (a["S"] != S).nnz == 0
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)


def test_a_failing_device_call_is_passed_on_not_returned(libs):
    """A dimension mismatch is the CPU library's to report (with its own message on C): the shim passes the call on."""
    shim, fake = libs
    S = rmat_small()
    A = make_csr(fake, S, FP64)
    Cm = empty_matrix(fake, S.shape[0] + 1, S.shape[1], FP64)
    before = fake.fake_cpu_calls(0)
    shim.GrB_mxm(Cm, None, None, obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP64"), A, A, None)
    assert fake.fake_cpu_calls(0) == before + 1 and fake.fake_cpu_saw_intact() == 1
    assert look(fake, Cm, FP64)["held"] and look(fake, A, FP64)["held"]
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)


def test_pagerank_step_on_arrays_of_the_device_visible_allocator(libs, gba):
    """The application installed gba_host_malloc / _calloc / _realloc / _free through GxB_init: the handles hold pinned
    memory the device can address, GrB_mxv reads A and x in place (GBA_MEM_HOST_VISIBLE) and y comes back in such memory."""
    shim, fake = libs
    from graphblas_amd.binding import lib
    fns = [C.cast(getattr(lib, n), C.c_void_p) for n in ("gba_host_malloc", "gba_host_calloc", "gba_host_realloc", "gba_host_free")]
    fake.fake_set_allocator(*fns)
    try:
        S = rmat_small(scale=10, ef=16, values="uniform").astype(np.float32)
        n = S.shape[0]
        A = make_csr(fake, S, FP32)
        xs = (np.arange(n) % 7 + 1).astype(np.float32)
        x = fake.fake_vector_dense(FP32, n, None, ptr(xs), n)
        y0 = np.ones(n, np.float32)
        y = fake.fake_vector_dense(FP32, n, None, ptr(y0.copy()), n)
        before = fake.fake_cpu_calls(2)
        for _ in range(2):                                   # the second call runs on what the first packed back
            info = shim.GrB_mxv(y, None, obj(fake, "GrB_PLUS_FP32"), obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP32"), A, x, None)
            assert info == 0
        assert fake.fake_cpu_calls(2) == before
        got = look(fake, y, FP32, vector=True)
        want = y0.astype(np.float64) + 2 * (S.astype(np.float64) @ xs.astype(np.float64))
        np.testing.assert_allclose(got["x"].astype(np.float64), want, rtol=64 * np.finfo(np.float32).eps * 4)
        fields = (C.c_int64 * 9)(); arrays = (C.c_void_p * 5)()
        fake.fake_vector_look(y, fields, arrays)
        lib.gba_is_device_visible.restype = C.c_int
        lib.gba_is_device_visible.argtypes = [C.c_void_p]
        assert lib.gba_is_device_visible(arrays[3]) == 1, "y was packed back in memory of another allocator"
        fake.fake_matrix_free(A); fake.fake_vector_free(x); fake.fake_vector_free(y)
    finally:
        libc = C.CDLL(None)
        fake.fake_set_allocator(*[C.cast(getattr(libc, n), C.c_void_p) for n in ("malloc", "calloc", "realloc", "free")])


@pytest.mark.parametrize("switch,want", [(0.0625, "hyper"), (0.0, "sparse")])
def test_the_result_is_conformed_by_the_hyper_switch_of_C(libs, switch, want):
    """C = A*A where only four of 256 rows of the result hold entries: with C's hyper_switch at the library's default
    the handle comes back hypersparse (non-empty vectors <= hyper_switch * vdim, Source/GB_conform_hyper.c), with 0
    (never asked for) it stays sparse; the entries are the same."""
    shim, fake = libs
    n = 256
    rng = np.random.default_rng(8)
    D = np.zeros((n, n))
    for r in (3, 77, 130, 250):
        D[r, rng.choice(n, 12, replace=False)] = rng.integers(1, 5, 12)
    S = sp.csr_matrix(D)
    A = make_csr(fake, S, FP64)
    Cm = empty_matrix(fake, n, n, FP64)
    fake.fake_matrix_set_hyper_switch.argtypes = [C.c_void_p, C.c_double]
    fake.fake_matrix_set_hyper_switch(Cm, switch)
    before = fake.fake_cpu_calls(0)
    assert shim.GrB_mxm(Cm, None, None, obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP64"), A, A, None) == 0
    assert fake.fake_cpu_calls(0) == before
    got = look(fake, Cm, FP64)
    prod = (S @ S).tocsr(); prod.sort_indices()
    assert (got["S"] != prod).nnz == 0 and got["S"].nnz == prod.nnz
    if want == "hyper":
        assert got["sparsity"] == HYPER
        assert np.array_equal(got["h"], np.flatnonzero(np.diff(prod.indptr)))
    else:
        assert got["sparsity"] == SPARSE
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)


def test_an_iso_valued_output_is_taken(libs):
    """C += A*A where C holds one stored value for all its entries (as GrB_assign of a scalar leaves it): the call
    stays on the device and the result is that of the same C with its values written out."""
    shim, fake = libs
    S = rmat_small(scale=7)
    n = S.shape[0]
    A = make_csr(fake, S, FP64)
    P = sp.random(n, n, density=0.05, random_state=np.random.default_rng(2), format="csr")
    P.sort_indices()
    p, i = P.indptr.astype(np.int64), P.indices.astype(np.int64)
    Cm = fake.fake_matrix_sparse(FP64, n, n, 1, 0, 0, ptr(p), None, ptr(i), ptr(np.array([2.0])), 1, 0)
    before = fake.fake_cpu_calls(0)
    assert shim.GrB_mxm(Cm, None, obj(fake, "GrB_PLUS_FP64"), obj(fake, "GrB_PLUS_TIMES_SEMIRING_FP64"), A, A, None) == 0
    assert fake.fake_cpu_calls(0) == before, "the call fell through to the CPU library's stub"
    got = look(fake, Cm, FP64)
    Pw = sp.csr_matrix((np.full(P.nnz, 2.0), P.indices, P.indptr), shape=P.shape)
    want = (Pw + S @ S).tocsr(); want.sort_indices()
    assert not got["iso"] and got["S"].nnz == want.nnz and (got["S"] != want).nnz == 0
    fake.fake_matrix_free(A); fake.fake_matrix_free(Cm)
