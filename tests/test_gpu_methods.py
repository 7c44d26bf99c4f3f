"""GPU parity for dot3, dot2, dot4, saxpy4, saxpy5, saxbit, transpose and wait, through the C ABI."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, Sparsity, gen
from oracle import oracle
from util import assert_same, assert_sums_within_bound

pytestmark = pytest.mark.gpu


def _rand(n, m, density, seed, lo=1, hi=9, t=Type.FP64, fmt="csc"):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format=fmt)
    S.data = rng.integers(lo, hi, len(S.data)).astype(np.float64)
    return Matrix.from_scipy(S, type=t, by_row=False)


def _free(*ds):
    for d in ds:
        if d is not None:
            d.free()


# ------------------------------------------------------------------ dot3 ----
@pytest.mark.parametrize("scale", [8, 12])
def test_dot3_triangle_count_bit_exact(gba, scale):
    """C<L,struct> = L*L' with PLUS_PAIR_INT64: every C(i,j), every zombie, and the total."""
    G = gen.rmat(scale, 16, seed=42, symmetric=True, drop_diagonal=True, values="ones", by_row=False)
    L = gen.tril_strict(G)
    dL = L.to_device()
    out, rep = gba.AxB_dot3(dL, True, dL, dL, "PLUS_PAIR_INT64")
    got = out.to_host()
    want = oracle.dot3(L, True, L, L, "PLUS_PAIR_INT64")
    assert got.nzombies == want.nzombies
    assert np.array_equal(got.p, want.p)
    assert np.array_equal(got.i, want.i)                 # zombies are flipped in the same places
    live = want.i >= 0
    assert np.array_equal(got.x[live], want.x[live])
    S = L.to_scipy(); S.data[:] = 1
    assert int(got.x[live].sum()) == int((S @ S.T).multiply(S).sum())
    _free(dL, out)


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP64", Type.FP64), ("MIN_PLUS_INT32", Type.INT32),
                                  ("MAX_FIRSTJ_INT64", Type.INT64), ("LOR_LAND_BOOL", Type.BOOL),
                                  ("PLUS_TIMES_FP32", Type.FP32)])
@pytest.mark.parametrize("struct", [True, False])
def test_dot3_semirings_and_valued_mask(gba, sr, t, struct):
    A = _rand(300, 200, 0.05, 1, t=t)
    B = _rand(300, 250, 0.05, 2, t=t)
    M = _rand(200, 250, 0.1, 3, lo=0, hi=2, t=Type.INT8)
    dA, dB, dM = A.to_device(), B.to_device(), M.to_device()
    out, _ = gba.AxB_dot3(dM, struct, dA, dB, sr)
    got = out.to_host()
    want = oracle.dot3(M, struct, A, B, sr)
    assert got.nzombies == want.nzombies
    assert_same(got, want, exact=True, what=sr)
    _free(dA, dB, dM, out)


def test_dot3_long_lists_use_the_wave_path(gba):
    """Hub vectors (thousands of entries) against short ones: wave kernel and gallop branches."""
    n = 20000
    rng = np.random.default_rng(4)
    r = np.concatenate([rng.integers(0, n, 6000), rng.integers(0, n, 6000), rng.integers(0, n, 40000)])
    c = np.concatenate([np.zeros(6000, np.int64), np.ones(6000, np.int64), rng.integers(0, n, 40000)])
    S = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n)).tocsc()
    S.sum_duplicates(); S.data[:] = 1
    A = Matrix.from_scipy(S, by_row=False)
    Ms = sp.coo_matrix((np.ones(5), ([0, 1, 0, 5, 1], [0, 0, 1, 1, 7])), shape=(n, n)).tocsc()
    M = Matrix.from_scipy(Ms, by_row=False)
    dA, dM = A.to_device(), M.to_device()
    out, _ = gba.AxB_dot3(dM, True, dA, dA, "PLUS_PAIR_INT64")
    assert_same(out.to_host(), oracle.dot3(M, True, A, A, "PLUS_PAIR_INT64"), exact=True)
    _free(dA, dM, out)


# ------------------------------------------------------------- dot2 / dot4 ----
@pytest.mark.parametrize("sr,t,exact", [("PLUS_TIMES_FP32", Type.FP32, True), ("PLUS_TIMES_FP64", Type.FP64, True),
                                        ("MIN_PLUS_FP64", Type.FP64, True), ("PLUS_SECOND_INT32", Type.INT32, True),
                                        ("MAX_TIMES_INT8", Type.INT8, True), ("LOR_LAND_BOOL", Type.BOOL, True),
                                        ("MIN_FIRSTI_INT64", Type.INT64, True)])
@pytest.mark.parametrize("bkind", ["full", "bitmap"])
def test_dot2_spmv_and_spmm(gba, sr, t, exact, bkind):
    """C = A'*B with B dense: the method default row-major GrB_mxv takes (dot2, bitmap result)."""
    from graphblas_amd.binding import NP_TYPES
    A = _rand(500, 400, 0.03, 5, t=t)
    rng = np.random.default_rng(6)
    xb = rng.integers(0, 5, (500, 3)).astype(NP_TYPES[t])
    B = Matrix.dense(xb, present=(rng.random((500, 3)) < 0.7) if bkind == "bitmap" else None, by_row=False)
    dA, dB = A.to_device(), B.to_device()
    out, _ = gba.AxB_dot2(None, False, False, dA, dB, sr)
    got = out.to_host()
    assert got.sparsity == Sparsity.BITMAP
    assert_same(got, oracle.dot2(None, False, False, A, B, sr), exact=exact, what=sr)
    _free(dA, dB, out)


def test_dot2_uniform_fp32_within_tolerance(gba):
    A = gen.rmat(12, 16, seed=42, type=Type.FP32, by_row=False)
    x = (1.0 - np.random.default_rng(1).random(A.vlen)).astype(np.float32)
    B = Matrix.full_vector(x)
    dA, dB = A.to_device(), B.to_device()
    out, _ = gba.AxB_dot2(None, False, False, dA, dB, "PLUS_TIMES_FP32")
    # LDS tiles and wave reductions against the CPU's left-to-right sum: positive values, so both are held to the
    # a-priori bound with the length of A(:,i) as the number of products behind y(i)
    got, want = out.to_host(), oracle.dot2(None, False, False, A, B, "PLUS_TIMES_FP32")
    rows = got.coo()[0]
    assert_sums_within_bound(got, want, np.diff(A.p)[rows], what="dot2 fp32")
    _free(dA, dB, out)


@pytest.mark.parametrize("comp", [False, True])
def test_dot2_masked(gba, comp):
    A = _rand(200, 150, 0.05, 7)
    B = Matrix.dense(np.random.default_rng(8).integers(1, 5, (200, 4)).astype(np.float64), by_row=False)
    M = Matrix.dense(np.random.default_rng(9).integers(0, 2, (150, 4)).astype(np.int8), by_row=False)
    dA, dB, dM = A.to_device(), B.to_device(), M.to_device()
    out, _ = gba.AxB_dot2(dM, comp, False, dA, dB, "PLUS_TIMES_FP64")
    assert_same(out.to_host(), oracle.dot2(M, comp, False, A, B, "PLUS_TIMES_FP64"), exact=True)
    _free(dA, dB, dM, out)


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP32", Type.FP32), ("MIN_PLUS_FP64", Type.FP64), ("PLUS_TIMES_INT16", Type.INT16),
                                  ("MAX_MIN_UINT8", Type.UINT8), ("LXOR_LAND_BOOL", Type.BOOL)])
def test_dot4_in_place(gba, sr, t):
    """C += A'*B, C full: the PageRank-shape y += A*x with A held by row."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(10)
    A = _rand(600, 500, 0.02, 11, t=t, lo=0, hi=4)
    B = Matrix.dense(rng.integers(0, 4, (600, 2)).astype(NP_TYPES[t]), by_row=False)
    C0 = Matrix.dense(rng.integers(0, 4, (500, 2)).astype(NP_TYPES[t]), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    rep = gba.AxB_dot4(dC, dA, dB, sr)
    assert rep is not None
    assert_same(dC.to_host(), oracle.dot4(C0, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dC)


def test_dot_long_vectors(gba):
    """One vector of A with 50000 entries goes through the workgroup-per-vector kernel."""
    n = 60000
    rng = np.random.default_rng(12)
    r = np.concatenate([rng.choice(n, 50000, replace=False), rng.integers(0, n, 20000)])
    c = np.concatenate([np.full(50000, 3), rng.integers(0, 50, 20000)])
    S = sp.coo_matrix((rng.integers(1, 4, len(r)).astype(np.float64), (r, c)), shape=(n, 50)).tocsc()
    S.sum_duplicates()
    A = Matrix.from_scipy(S, by_row=False)
    B = Matrix.full_vector(rng.integers(1, 4, n).astype(np.float64))
    C0 = Matrix.full_vector(np.zeros(50))
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.AxB_dot4(dC, dA, dB, "PLUS_TIMES_FP64")
    assert_same(dC.to_host(), oracle.dot4(C0, A, B, "PLUS_TIMES_FP64"), exact=True)
    _free(dA, dB, dC)


# ------------------------------------------------------ saxpy4 / saxpy5 / saxbit ----
@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP32", Type.FP32), ("PLUS_TIMES_FP64", Type.FP64), ("MIN_PLUS_FP64", Type.FP64),
                                  ("MAX_TIMES_INT8", Type.INT8), ("PLUS_TIMES_UINT16", Type.UINT16), ("TIMES_PLUS_INT32", Type.INT32),
                                  ("LOR_LAND_BOOL", Type.BOOL), ("BXOR_BAND_UINT32", Type.UINT32), ("MIN_SECONDJ_INT64", Type.INT64)])
@pytest.mark.parametrize("bkind", ["full", "bitmap"])
def test_saxpy4_in_place(gba, sr, t, bkind):
    """C += A*B, C full, A sparse by column, B dense: GrB_vxm / GrB_mxv(T0) with accum."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(13)
    A = _rand(400, 300, 0.03, 14, t=t, lo=0, hi=3)
    xb = rng.integers(0, 3, (300, 2)).astype(NP_TYPES[t])
    B = Matrix.dense(xb, present=(rng.random((300, 2)) < 0.6) if bkind == "bitmap" else None, by_row=False)
    C0 = Matrix.dense(rng.integers(0, 3, (400, 2)).astype(NP_TYPES[t]), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    rep = gba.AxB_saxpy4(dC, dA, dB, sr)
    assert rep is not None
    assert_same(dC.to_host(), oracle.saxpy4(C0, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dC)


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP32", Type.FP32), ("PLUS_TIMES_FP64", Type.FP64), ("MIN_PLUS_FP64", Type.FP64),
                                  ("PLUS_TIMES_INT64", Type.INT64), ("BXOR_BAND_UINT32", Type.UINT32),
                                  ("MAX_SECOND_INT32", Type.INT32), ("PLUS_PAIR_INT64", Type.INT64), ("MIN_PLUS_FP32", Type.FP32),
                                  ("PLUS_FIRST_FP64", Type.FP64), ("TIMES_SECOND_UINT64", Type.UINT64)])
def test_saxpy4_large_goes_through_destination_bins(gba, sr, t):
    """More than 2^20 entries and B full: products are binned by destination and summed in LDS panels.
    The matrix has a vector longer than a tile, a run of empty vectors, a destination hit 30000 times,
    bins with several chunks and a bin with one; small integer values keep every sum exact."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(21)
    n, inner = 200_000, 150_000
    dest = np.concatenate([rng.integers(0, 60_000, 900_000), rng.integers(0, 180_000, 400_000),
                           rng.integers(180_000, n, 20_000), np.full(30_000, 7), rng.choice(n, 50_000, replace=False)])
    k = np.concatenate([rng.integers(0, 100_000, 1_320_000), rng.choice(100_000, 30_000, replace=False), np.full(50_000, 5)])
    S = sp.coo_matrix((rng.integers(1, 3, len(k)).astype(np.float64), (dest, k)), shape=(n, inner)).tocsc()
    S.sum_duplicates(); S.data = np.minimum(S.data, 3)
    assert S.nnz > (1 << 20)
    A = Matrix.from_scipy(S, type=t, by_row=False)
    B = Matrix.dense(rng.integers(0, 3, (inner, 2)).astype(NP_TYPES[t]), by_row=False)
    C0 = Matrix.dense(rng.integers(0, 3, (n, 2)).astype(NP_TYPES[t]), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    rep = gba.AxB_saxpy4(dC, dA, dB, sr)
    assert rep is not None
    assert_same(dC.to_host(), oracle.saxpy4(C0, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dC)


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP64", Type.FP64), ("MIN_PLUS_INT32", Type.INT32), ("MAX_FIRST_FP32", Type.FP32)])
@pytest.mark.parametrize("akind", ["full", "bitmap"])
def test_saxpy5_in_place(gba, sr, t, akind):
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(15)
    a = rng.integers(1, 4, (70, 300)).astype(NP_TYPES[t])
    A = Matrix.dense(a, present=(rng.random((70, 300)) < 0.8) if akind == "bitmap" else None, by_row=False)
    B = _rand(300, 40, 0.1, 16, t=t)
    C0 = Matrix.dense(rng.integers(0, 3, (70, 40)).astype(NP_TYPES[t]), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    rep = gba.AxB_saxpy5(dC, dA, dB, sr)
    assert rep is not None
    assert_same(dC.to_host(), oracle.saxpy5(C0, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dC)


@pytest.mark.parametrize("sr,t", [("MIN_PLUS_FP64", Type.FP64), ("PLUS_TIMES_INT32", Type.INT32), ("LOR_LAND_BOOL", Type.BOOL),
                                  ("PLUS_PAIR_INT64", Type.INT64), ("MAX_TIMES_UINT8", Type.UINT8)])
@pytest.mark.parametrize("mask", [None, ("bitmap", False), ("bitmap", True), ("sparse", False), ("sparse", True)])
def test_saxbit(gba, sr, t, mask):
    """C<#M> = A*B with C bitmap: the frontier step with a bitmap/full vector."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(17)
    A = _rand(500, 500, 0.02, 18, t=t, lo=0, hi=4)
    B = Matrix.dense(rng.integers(0, 4, (500, 2)).astype(NP_TYPES[t]), present=rng.random((500, 2)) < 0.3, by_row=False)
    M, comp = None, False
    if mask is not None:
        kind, comp = mask
        if kind == "bitmap":
            M = Matrix.dense(np.ones((500, 2), np.int8), present=rng.random((500, 2)) < 0.5, by_row=False)
        else:
            Ms = sp.random(500, 2, density=0.4, random_state=rng, format="csc")
            M = Matrix.from_scipy(Ms, type=Type.FP64, by_row=False)
    dA, dB = A.to_device(), B.to_device()
    dM = M.to_device() if M is not None else None
    out, _ = gba.AxB_saxbit(dM, comp, True, dA, dB, sr)
    assert_same(out.to_host(), oracle.saxbit(M, comp, True, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dM, out)


@pytest.mark.parametrize("nvec", [1, 8, 9])
def test_saxpy4_bins_with_iso_A_and_several_vectors(gba, nvec):
    """A iso (one stored value), 1, 8 and 9 dense vectors (nine is beyond what the bins take: the tile kernel
    runs), a destination count that is not a multiple of the bin width."""
    rng = np.random.default_rng(27)
    n, inner = 70_001, 60_000
    S = sp.random(n, inner, density=1_200_000 / (n * inner), random_state=rng, format="csc")
    S.data[:] = 2.0
    assert S.nnz > (1 << 20)
    A0 = Matrix.from_scipy(S, type=Type.FP64, by_row=False)
    A = Matrix(A0.vlen, A0.vdim, Type.FP64, Sparsity.SPARSE, True, p=A0.p, i=A0.i, x=np.array([2.0]), iso=True, nvals=A0.nvals)
    B = Matrix.dense(rng.integers(0, 3, (inner, nvec)).astype(np.float64), by_row=False)
    C0 = Matrix.dense(rng.integers(0, 3, (n, nvec)).astype(np.float64), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    assert gba.AxB_saxpy4(dC, dA, dB, "PLUS_TIMES_FP64") is not None
    want = C0.x.reshape(nvec, n).T + S @ B.x.reshape(nvec, inner).T
    assert np.array_equal(dC.to_host().x.reshape(nvec, n).T, want)
    _free(dA, dB, dC)


def _hub_matrix(n, inner, t, seed):
    """A sparse by column with more than four tiles of entries: one vector longer than a tile, a run of empty
    vectors, a destination hit from thousands of vectors; small integer values."""
    rng = np.random.default_rng(seed)
    dest = np.concatenate([rng.integers(0, n, 40_000), np.full(3_000, 7), rng.choice(n, 9_000, replace=False)])
    k = np.concatenate([rng.integers(0, inner // 2, 40_000), rng.choice(inner // 2, 3_000, replace=False), np.full(9_000, 5)])
    S = sp.coo_matrix((rng.integers(1, 3, len(k)).astype(np.float64), (dest, k)), shape=(n, inner)).tocsc()
    S.sum_duplicates(); S.data = np.minimum(S.data, 3)
    assert S.nnz > 4 * 2048
    return Matrix.from_scipy(S, type=t, by_row=False)


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP64", Type.FP64), ("MIN_SECONDJ_INT64", Type.INT64), ("MAX_TIMES_INT8", Type.INT8),
                                  ("LOR_LAND_BOOL", Type.BOOL), ("PLUS_FIRSTI_INT32", Type.INT32)])
@pytest.mark.parametrize("bkind", ["full", "bitmap"])
def test_saxpy4_over_entry_tiles(gba, sr, t, bkind):
    """Enough entries for the entry-balanced kernel (but too few, or a semiring or a bitmap B unfit, for the
    destination bins): hub vector, empty vectors, positional multipliers, narrow types widened and narrowed."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(23)
    n, inner = 15_000, 12_000
    A = _hub_matrix(n, inner, t, 24)
    B = Matrix.dense(rng.integers(0, 3, (inner, 2)).astype(NP_TYPES[t]),
                     present=(rng.random((inner, 2)) < 0.6) if bkind == "bitmap" else None, by_row=False)
    C0 = Matrix.dense(rng.integers(0, 3, (n, 2)).astype(NP_TYPES[t]), by_row=False)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    assert gba.AxB_saxpy4(dC, dA, dB, sr) is not None
    assert_same(dC.to_host(), oracle.saxpy4(C0, A, B, sr), exact=True, what=sr)
    _free(dA, dB, dC)


@pytest.mark.parametrize("sr,t", [("ANY_SECONDI_INT64", Type.INT64), ("PLUS_TIMES_INT32", Type.INT32), ("MIN_PLUS_FP64", Type.FP64)])
@pytest.mark.parametrize("mask", [None, ("bitmap", True), ("sparse", False)])
def test_saxbit_over_entry_tiles(gba, sr, t, mask):
    """The BFS step with a bitmap frontier at a size that takes the entry-balanced kernel.  ANY may keep any
    contributing product: for it only the pattern is compared, and that every value is one of the candidates."""
    from graphblas_amd.binding import NP_TYPES
    rng = np.random.default_rng(25)
    n, inner = 15_000, 12_000
    A = _hub_matrix(n, inner, t, 26)
    B = Matrix.dense(rng.integers(1, 4, (inner, 1)).astype(NP_TYPES[t]), present=rng.random((inner, 1)) < 0.4, by_row=False)
    M, comp = None, False
    if mask is not None:
        kind, comp = mask
        if kind == "bitmap":
            M = Matrix.dense(np.ones((n, 1), np.int8), present=rng.random((n, 1)) < 0.5, by_row=False)
        else:
            M = Matrix.from_scipy(sp.random(n, 1, density=0.4, random_state=rng, format="csc"), type=Type.FP64, by_row=False)
    dA, dB = A.to_device(), B.to_device()
    dM = M.to_device() if M is not None else None
    out, _ = gba.AxB_saxbit(dM, comp, True, dA, dB, sr)
    got, want = out.to_host(), oracle.saxbit(M, comp, True, A, B, sr)
    if sr.startswith("ANY"):
        assert_same(got, want, check_values=False, what=sr)
        r, c, v = got.coo()
        S, present = A.to_scipy().tocsr(), np.asarray(B.b).ravel() != 0
        for i, k in list(zip(r, v))[::37]:                      # SECONDI: the value is the index k of a contributing B(k)
            assert present[k] and S[i, k] != 0, (i, k)
    else:
        assert_same(got, want, exact=True, what=sr)
    _free(dA, dB, dM, out)


def test_saxbit_dense_times_dense(gba):
    rng = np.random.default_rng(19)
    A = Matrix.dense(rng.integers(0, 4, (30, 20)).astype(np.float64), present=rng.random((30, 20)) < 0.5, by_row=False)
    B = Matrix.dense(rng.integers(0, 4, (20, 10)).astype(np.float64), present=rng.random((20, 10)) < 0.5, by_row=False)
    dA, dB = A.to_device(), B.to_device()
    out, _ = gba.AxB_saxbit(None, False, False, dA, dB, "PLUS_TIMES_FP64")
    assert_same(out.to_host(), oracle.saxbit(None, False, False, A, B, "PLUS_TIMES_FP64"), exact=True)
    _free(dA, dB, out)


# ------------------------------------------------------------ wait / transpose ----
def test_wait_sorts_and_kills_zombies(gba):
    A = gen.rmat(10, 16, seed=3, by_row=False, values="int")
    dA = A.to_device()
    out, _, _ = gba.AxB_saxpy3(None, False, False, dA, dA, "PLUS_TIMES_FP64")
    assert out.view.jumbled
    gba.matrix_wait(out)
    got = out.to_host()
    assert not got.jumbled
    for j in range(got.vdim):
        seg = got.i[got.p[j]:got.p[j + 1]]
        assert np.all(np.diff(seg) > 0)
    assert_same(got, oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64", wait=True), exact=True)
    _free(out)
    G = gen.rmat(9, 8, seed=5, symmetric=True, drop_diagonal=True, values="ones", by_row=False)
    L = gen.tril_strict(G)
    dL = L.to_device()
    out, _ = gba.AxB_dot3(dL, True, dL, dL, "PLUS_PAIR_INT64")
    nz = out.view.nzombies
    assert nz > 0
    gba.matrix_wait(out)
    got = out.to_host()
    assert got.nzombies == 0 and got.nvals == L.nvals - nz and np.all(got.i >= 0)
    _free(dA, dL, out)


def test_transpose(gba):
    A = gen.rmat(10, 8, seed=9, by_row=False)
    dA = A.to_device()
    out = gba.transpose(dA)
    got = out.to_host()
    S = sp.csc_matrix((A.x, A.i, A.p), shape=(A.vlen, A.vdim))
    T = S.T.tocsc(); T.sort_indices()
    assert np.array_equal(got.p, T.indptr) and np.array_equal(got.i, T.indices) and np.array_equal(got.x, T.data)
    _free(dA, out)
