"""GPU parity: gba_AxB_saxpy3 / flopcount / cumsum against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, DeviceMatrix, Type, Sparsity, gen
from oracle import oracle
from util import assert_same, assert_sums_within_bound

pytestmark = pytest.mark.gpu


def _csc(S, t=Type.FP64):
    return Matrix.from_scipy(S, type=t, by_row=False)


def _rand(n, m, density, seed, lo=1, hi=9):
    rng = np.random.default_rng(seed)
    S = sp.random(n, m, density=density, random_state=rng, format="csc")
    S.data = rng.integers(lo, hi, len(S.data)).astype(np.float64)
    return S


def _run(gba, A, B, sr, M=None, comp=False, struct=True):
    dA, dB = A.to_device(), B.to_device()
    dM = M.to_device() if M is not None else None
    out, rep, applied = gba.AxB_saxpy3(dM, comp, struct, dA, dB, sr)
    got = out.to_host()
    for d in (dA, dB, dM, out):
        if d is not None:
            d.free()
    return got, rep, applied


def test_cumsum(gba):
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 2048, 2049, 100000, 3_000_001):
        c = rng.integers(0, 1000, n)
        got = gba.cumsum(c)
        assert np.array_equal(got, np.concatenate([[0], np.cumsum(c)]).astype(np.int64)), n


@pytest.mark.parametrize("scale,ef", [(8, 8), (10, 16)])
def test_rmat_plus_times_fp64(gba, scale, ef):
    A = gen.rmat(scale, ef, seed=42, by_row=False)
    got, rep, _ = _run(gba, A, A, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64")
    # the summation order differs (LDS and memory atomics against the CPU's left-to-right sum); the values are
    # positive, so both sides are held to the a-priori bound with the number of products behind every C(i,j)
    P = A.to_scipy().copy(); P.data[:] = 1
    counts = (P @ P).tocsr(); counts.sort_indices()
    assert_sums_within_bound(got, want, counts.tocoo().data, what=f"RMAT-{scale}")


def test_rmat_integer_valued_is_bit_exact(gba):
    A = gen.rmat(10, 16, seed=7, by_row=False, values="int")
    got, _, _ = _run(gba, A, A, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64")
    assert_same(got, want, exact=True)          # all partial sums are integers < 2^53


def test_low_degree_erdos_renyi_hash_bins(gba):
    A = gen.erdos_renyi(5000, 12000, seed=3, by_row=False, values="int")
    got, _, _ = _run(gba, A, A, "PLUS_TIMES_FP64")
    assert_same(got, oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64"), exact=True)


def test_every_bin_including_global(gba):
    """A with a few very heavy rows: outputs land in wave, block and global bins."""
    n = 40000
    rng = np.random.default_rng(5)
    rows = [rng.integers(0, n, 60000), np.zeros(30000, np.int64), rng.integers(0, n, 30000)]
    cols = [rng.integers(0, n, 60000), rng.integers(0, n, 30000), np.ones(30000, np.int64)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    S = sp.coo_matrix((rng.integers(1, 5, len(r)).astype(np.float64), (r, c)), shape=(n, n)).tocsc()
    S.sum_duplicates()
    A = _csc(S)
    got, _, _ = _run(gba, A, A, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64")
    assert want.nvals > 100000
    assert_same(got, want, exact=True)


def test_empty_and_ragged(gba):
    E = _csc(sp.csc_matrix((50, 50)))
    got, _, _ = _run(gba, E, E, "PLUS_TIMES_FP64")
    assert got.nvals == 0 and len(got.p) == 51
    A = _csc(_rand(30, 0, 0.0, 1))            # 30 x 0
    B = _csc(sp.csc_matrix((0, 20)))          # 0 x 20
    got, _, _ = _run(gba, A, B, "PLUS_TIMES_FP64")
    assert got.nvals == 0 and got.vlen == 30 and got.vdim == 20
    A = _csc(_rand(17, 33, 0.2, 2)); B = _csc(_rand(33, 5, 0.3, 3))
    got, _, _ = _run(gba, A, B, "PLUS_TIMES_FP64")
    assert_same(got, oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64"), exact=True)


SEMIRINGS = [
    ("PLUS_TIMES_INT32", Type.INT32), ("PLUS_TIMES_INT8", Type.INT8), ("PLUS_TIMES_UINT16", Type.UINT16),
    ("PLUS_TIMES_INT64", Type.INT64), ("PLUS_TIMES_FP32", Type.FP32),
    ("MIN_PLUS_FP64", Type.FP64), ("MAX_PLUS_INT32", Type.INT32), ("MIN_MAX_UINT8", Type.UINT8),
    ("MIN_FIRST_INT16", Type.INT16), ("MAX_SECOND_FP32", Type.FP32), ("TIMES_PLUS_INT32", Type.INT32),
    ("TIMES_MIN_FP64", Type.FP64), ("PLUS_PAIR_INT64", Type.INT64), ("PLUS_MINUS_INT32", Type.INT32),
    ("PLUS_RMINUS_INT64", Type.INT64), ("PLUS_DIV_INT32", Type.INT32), ("PLUS_RDIV_UINT8", Type.UINT8),
    ("PLUS_ISGT_INT32", Type.INT32), ("PLUS_LOR_FP64", Type.FP64), ("MAX_LAND_INT8", Type.INT8),
    ("LOR_LAND_BOOL", Type.BOOL), ("LXOR_LOR_BOOL", Type.BOOL), ("EQ_LXOR_BOOL", Type.BOOL), ("LAND_EQ_BOOL", Type.BOOL),
    ("LOR_GT_INT32", Type.INT32), ("LXOR_LE_FP64", Type.FP64), ("LAND_NE_UINT8", Type.UINT8), ("EQ_EQ_INT16", Type.INT16),
    ("BOR_BAND_UINT8", Type.UINT8), ("BXOR_BOR_UINT32", Type.UINT32), ("BXNOR_BXOR_UINT64", Type.UINT64),
    ("BAND_BOR_UINT16", Type.UINT16),
    ("MIN_FIRSTI_INT64", Type.INT64), ("MAX_FIRSTJ_INT32", Type.INT32), ("PLUS_SECONDJ_INT64", Type.INT64),
    ("MIN_SECONDI1_INT64", Type.INT64), ("TIMES_FIRSTI1_INT32", Type.INT32),
]


@pytest.mark.parametrize("sr,t", SEMIRINGS)
def test_builtin_semirings_bit_exact(gba, sr, t):
    """Integer, boolean, MIN/MAX, PAIR, bitwise and positional semirings agree bit for bit.
    fp32 PLUS_TIMES uses small integers, so every partial sum is exact."""
    S = _rand(300, 300, 0.03, 17, lo=0, hi=4)
    S.data[::7] = 0          # explicit zeros exercise LOR/LAND/comparators and division by zero
    A = _csc(S, t)
    got, _, _ = _run(gba, A, A, sr)
    assert_same(got, oracle.saxpy3(None, False, False, A, A, sr), exact=True, what=sr)


def test_any_monoid_is_a_valid_choice(gba):
    """ANY takes whichever product arrives first (the reference skips value checks for it,
    Test/GB_spec_compare.m:17-22): pattern must match; each value must be one of the products."""
    S = _rand(200, 200, 0.05, 23)
    A = _csc(S, Type.INT64)
    got, _, _ = _run(gba, A, A, "ANY_SECONDI_INT64")
    want = oracle.saxpy3(None, False, False, A, A, "ANY_SECONDI_INT64")
    assert_same(got, want, check_values=False)
    r, c, v = got.coo()
    pat = (S != 0).tocsr()
    for i, j, k in zip(r[:2000], c[:2000], v[:2000]):
        assert pat[i, k] and pat[k, j]


@pytest.mark.parametrize("comp", [False, True])
@pytest.mark.parametrize("struct", [True, False])
@pytest.mark.parametrize("mkind", ["sparse", "bitmap", "full"])
def test_masks(gba, comp, struct, mkind):
    S = _rand(400, 400, 0.02, 31)
    A = _csc(S)
    rng = np.random.default_rng(9)
    mv = rng.integers(0, 2, (400, 400)).astype(np.int8)      # valued mask: zeros are "false"
    if mkind == "sparse":
        Ms = sp.random(400, 400, density=0.1, random_state=rng, format="csc")
        Ms.data = rng.integers(0, 2, len(Ms.data)).astype(np.float64)
        M = Matrix.from_scipy(Ms, type=Type.INT8, by_row=False)
    elif mkind == "bitmap":
        M = Matrix.dense(mv, present=rng.random((400, 400)) < 0.5, by_row=False)
    else:
        M = Matrix.dense(mv, by_row=False)
    got, _, applied = _run(gba, A, A, "PLUS_TIMES_FP64", M=M, comp=comp, struct=struct)
    assert applied
    assert_same(got, oracle.saxpy3(M, comp, struct, A, A, "PLUS_TIMES_FP64"), exact=True)


@pytest.mark.parametrize("mask", [None, "M", "notM", "dense"])
def test_flopcount(gba, mask):
    A = gen.rmat(9, 8, seed=11, by_row=False)
    rng = np.random.default_rng(2)
    M, comp = None, False
    if mask in ("M", "notM"):
        Ms = sp.random(512, 512, density=0.02, random_state=rng, format="csc")
        M = Matrix.from_scipy(Ms, by_row=False)
        comp = mask == "notM"
    elif mask == "dense":
        M = Matrix.dense(np.ones((512, 512), np.int8), by_row=False)
    dA = A.to_device()
    dM = M.to_device() if M is not None else None
    mw, fl = gba.AxB_saxpy3_flopcount(dM, comp, dA, dA)
    omw, ofl = oracle.saxpy3_flopcount(M, comp, A, A)
    assert mw == omw
    assert np.array_equal(fl, ofl)


def test_checksum_property_large(gba):
    """Size-independent check at a scale the oracle would take long on: for PLUS_TIMES with
    integer values, sum(C) == sum_k colsum_A(k) * rowsum_B(k), exactly (all sums < 2^53)."""
    A = gen.rmat(14, 16, seed=42, by_row=False, values="int")
    got, rep, _ = _run(gba, A, A, "PLUS_TIMES_FP64")
    S = sp.csc_matrix((A.x, A.i, A.p), shape=(A.vlen, A.vdim))
    colsum = np.asarray(S.sum(axis=0)).ravel()
    rowsum = np.asarray(S.sum(axis=1)).ravel()
    assert float(got.x.sum()) == float((colsum * rowsum).sum())
    assert np.all(np.diff(got.p) >= 0) and got.p[-1] == len(got.i)
    # no duplicate index inside a vector
    vec = np.repeat(np.arange(got.vdim), np.diff(got.p))
    key = vec * got.vlen + got.i
    assert len(np.unique(key)) == len(key)


def test_partition_by_flops_matches_the_host_split(gba):
    """gba_partition_by_flops (the GB_pslice analogue the multi-GPU bench shards rows by) against the numpy
    split of the same per-vector flop count: contiguous, covering, and balanced to within one vector."""
    from graphblas_amd.dist import row_flops, split_by_flops
    A = gen.rmat(12, 16, seed=42)                                  # CSR
    dA = A.to_device()
    rf = row_flops(A.p, A.i) + np.diff(A.p)                        # the reference's count includes nnz B(:,j) itself
    prefix = np.concatenate([[0], np.cumsum(rf)]).astype(np.int64)
    for nparts in (1, 2, 3, 8):
        part = gba.partition_by_flops(nparts, dA, dA)
        want = split_by_flops(prefix, 0, A.nrows, nparts)
        assert part[0] == 0 and part[-1] == A.nrows and len(part) == nparts + 1
        assert all(a <= b for a, b in zip(part, part[1:]))
        assert all(abs(a - b) <= 1 for a, b in zip(part, want)), (part, want)
        loads = np.diff(prefix[np.array(part)])
        assert loads.max() <= prefix[-1] / nparts + rf.max()
    dA.free()


@pytest.mark.parametrize("sr,t", [("PLUS_TIMES_FP64", Type.FP64), ("ANY_SECONDI_INT64", Type.INT64), ("MIN_PLUS_FP32", Type.FP32),
                                  ("LOR_LAND_BOOL", Type.BOOL), ("PLUS_TIMES_INT8", Type.INT8)])
def test_one_pass_road_for_a_few_light_vectors(gba, monkeypatch, sr, t):
    """At most 256 vectors in B with few entries (the late levels of a traversal): accumulated and emitted in one pass,
    counts read back once.  Same C as the two-pass road (GBA_NO_ONE_PASS=1), masked or not, empty vectors included."""
    n = 20000
    rng = np.random.default_rng(31)
    A = _csc(_rand(n, n, 4e-4, seed=32), t)                    # about 8 entries per vector
    nb = 9
    cols = np.repeat(np.arange(nb), [0, 1, 3, 40, 200, 0, 700, 5, 64])
    rows = np.concatenate([rng.choice(n, k, replace=False) for k in (0, 1, 3, 40, 200, 0, 700, 5, 64)]).astype(np.int64)
    Bs = sp.coo_matrix((rng.integers(1, 5, len(rows)).astype(np.float64), (rows, cols)), shape=(n, nb)).tocsc()
    B = _csc(Bs, t)
    M = Matrix.dense(np.ones((n, nb), np.int8), present=rng.random((n, nb)) < 0.5, by_row=False)
    for mask, comp in ((None, False), (M, False), (M, True)):
        want = oracle.saxpy3(mask, comp, True, A, B, sr)
        for off in ("", "1"):
            if off:
                monkeypatch.setenv("GBA_NO_ONE_PASS", "1")
            else:
                monkeypatch.delenv("GBA_NO_ONE_PASS", raising=False)
            got, rep, _ = _run(gba, A, B, sr, mask, comp, True)
            what = f"{sr} mask={mask is not None} comp={comp} one_pass={not off}"
            if sr.startswith("ANY"):
                assert_same(got, want, check_values=False, what=what)
            else:
                assert_same(got, want, exact=True, what=what)


def test_one_pass_road_hands_over_when_a_vector_is_too_heavy(gba):
    """One vector of the few has more multiply-adds than the one-pass table takes: the call falls back to the
    two-pass road and the result is the same."""
    n = 30000
    rng = np.random.default_rng(33)
    S = _rand(n, n, 3e-4, seed=34).tolil()
    S[rng.choice(n, 9000, replace=False), 7] = 2                # A(:,7) is long
    A = _csc(S.tocsc())
    rows = np.concatenate([[7, 11, 12], rng.choice(n, 30, replace=False)]).astype(np.int64)
    cols = np.concatenate([[0, 0, 0], np.ones(30, np.int64)])
    B = _csc(sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, 2)).tocsc())
    got, rep, _ = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert np.diff(want.p)[0] > 6144
    assert_same(got, want, exact=True)


@pytest.mark.parametrize("side", ["right", "left"])
def test_a_jumbled_operand_of_a_large_product(gba, side):
    """A*(A*A) and (A*A)*A: the first product comes back jumbled and is an operand of the second (held by row, the
    right operand is the one whose vectors the kernels walk as sorted lists).  The library multiplies a
    sorted copy (its panel kernels need sorted lists) and leaves the caller's matrix as it was; the result is that of
    the sorted operand."""
    A = gen.rmat(11, 8, seed=4, type=Type.FP64, values="int")
    S = A.to_scipy()
    dA = A.to_device()
    dC = DeviceMatrix.empty(A.nrows, A.ncols, Type.FP64)
    gba.mxm(dC, None, None, "PLUS_TIMES_FP64", dA, dA, None)
    first = dC.to_host()
    assert first.jumbled and first.nvals >= 1 << 16
    dD = DeviceMatrix.empty(A.nrows, A.ncols, Type.FP64)
    if side == "right":
        gba.mxm(dD, None, None, "PLUS_TIMES_FP64", dA, dC, None)
    else:
        gba.mxm(dD, None, None, "PLUS_TIMES_FP64", dC, dA, None)
    again = dC.to_host()
    assert again.jumbled and np.array_equal(again.i, first.i) and np.array_equal(again.x, first.x)
    want = (S @ S @ S).tocsr(); want.sort_indices()
    got = dD.to_host().to_scipy(); got.sort_indices()
    assert got.nnz == want.nnz and (got != want).nnz == 0           # small integers: exact
    for d in (dA, dC, dD):
        d.free()


def _long(S, rows, vlen, t=Type.FP64):
    """The CSC matrix S (k rows) with row r renamed rows[r]: a matrix of vlen rows that holds few of its indices."""
    S = S.tocsc(); S.sort_indices()
    return Matrix(vlen, S.shape[1], t, Sparsity.SPARSE, True, p=S.indptr.astype(np.int64), i=rows[S.indices].astype(np.int64),
                  x=S.data.astype(np.float64 if t == Type.FP64 else np.int8))


def _triples(Mx):
    """(col, index, value) of a host CSC Matrix, vectors sorted by index."""
    col = np.repeat(np.arange(Mx.nvec), np.diff(Mx.p))
    o = np.lexsort((Mx.i, col))
    return col[o], Mx.i[o], np.asarray(Mx.x)[o]


@pytest.mark.parametrize("mask", [None, "plain", "complemented", "valued"])
@pytest.mark.parametrize("vlen", [2**31 - 1, 2**40])
def test_vectors_of_2_31_indices_or_more(gba, vlen, mask):
    """Source/GB_AxB_saxpy3.c:120 takes any vector length; the device kernels keep indices in 32 bits, so a longer
    matrix is multiplied over the ranks of the indices it holds.  Expected: the same product formed on the host over
    ranks (numpy unique + scipy) and renamed back; integer values, exact."""
    rng = np.random.default_rng(77)
    k, n, nrows = 400, 60, 3000
    rows = np.sort(rng.choice(vlen, nrows, replace=False).astype(np.int64))
    rows[-1] = vlen - 1                                              # the last index of the range is in use
    As, Bs = _rand(nrows, k, 0.02, 78), _rand(k, n, 0.05, 79)
    A, B = _long(As, rows, vlen), _csc(Bs)
    M = Ms = None
    if mask:
        Ms = _rand(nrows, n, 0.3, 80, lo=0, hi=2)                    # values 0 and 1: a valued mask differs from a structural one
        other = np.setdiff1d(rng.choice(vlen, 500).astype(np.int64), rows)           # mask entries where A holds nothing
        extra = sp.csc_matrix((np.ones(len(other)), (np.arange(len(other)), rng.integers(0, n, len(other)))), shape=(len(other), n))
        allrows = np.concatenate([rows, other]); order = np.argsort(allrows, kind="stable")
        both = sp.vstack([Ms, extra]).tocsr()[order].tocsc()
        M = _long(both, allrows[order], vlen, Type.INT8)
    comp, struct = mask == "complemented", mask != "valued"
    got, rep, applied = _run(gba, A, B, "PLUS_TIMES_FP64", M, comp, struct)
    T = (As @ Bs).tocsc()
    if mask:
        keep = (Ms != 0) if not struct else sp.csc_matrix((np.ones(len(Ms.data)), Ms.indices, Ms.indptr), shape=Ms.shape)
        keep = keep.astype(bool).toarray()
        if comp:
            keep = ~keep
        T = sp.csc_matrix(np.where(keep & (T.toarray() != 0), T.toarray(), 0.0))
    T.sort_indices()
    want = _long(T, rows, vlen)
    assert (got.vlen, got.vdim) == (vlen, n) and got.nvals == want.nvals
    for g, w in zip(_triples(got), _triples(want)):
        assert np.array_equal(g, w)


def test_the_front_end_takes_vectors_of_2_40_indices(gba):
    """gba_mxm with an accumulator into a C that holds entries, all three of 2^40 rows."""
    rng = np.random.default_rng(91)
    vlen, k, n, nrows = 2**40, 300, 40, 2000
    rows = np.sort(rng.choice(vlen, nrows, replace=False).astype(np.int64))
    As, Bs, Cs = _rand(nrows, k, 0.03, 92), _rand(k, n, 0.05, 93), _rand(nrows, n, 0.05, 94)
    A, B, C0 = _long(As, rows, vlen), _csc(Bs), _long(Cs, rows, vlen)
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    gba.mxm(dC, None, "PLUS_FP64", "PLUS_TIMES_FP64", dA, dB, None)
    gba.matrix_wait(dC)
    got = dC.to_host()
    want = (Cs + As @ Bs).tocsc(); want.sort_indices()
    want = _long(want, rows, vlen)
    assert (got.vlen, got.vdim) == (vlen, n)
    for g, w in zip(_triples(got), _triples(want)):
        assert np.array_equal(g, w)
    for d in (dA, dB, dC):
        d.free()
