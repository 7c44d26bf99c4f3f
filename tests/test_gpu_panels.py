"""GPU parity for the heavy-vector path of gba_AxB_saxpy3: output vectors with more entries than
one LDS table holds are swept in index panels (symbolic: bitmap panels that also record the
numeric panel boundaries; numeric: hashed or directly addressed panels).  Every case is compared
with the CPU oracle through the C ABI; values are small integers so that sums are exact."""
import numpy as np
import pytest
import scipy.sparse as sp

from graphblas_amd import Matrix, Type, gen
from oracle import oracle
from util import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _panel_path_only(monkeypatch, request):
    """A handful of very heavy vectors would take the team kernels (whole grid per vector); the cases
    here are about the panel kernels, so that path is switched off unless a test asks for it."""
    if "team" not in request.node.name:
        monkeypatch.setenv("GBA_TEAM_MAX_VECTORS", "0")


def _heavy_pair(n, ncols, deg, nhubs, hub_len, light_len, seed, dense_cols=0):
    """A (n x n) with `nhubs` long columns and short ones elsewhere; B (n x ncols) whose columns
    each name `deg` columns of A, hubs included: every C(:,j) is a heavy vector."""
    rng = np.random.default_rng(seed)
    hubs = rng.choice(n, nhubs, replace=False)
    rows = [rng.integers(0, n, n * light_len)]
    cols = [np.repeat(np.arange(n), light_len)]
    for h in hubs:
        rows.append(rng.choice(n, hub_len, replace=False)); cols.append(np.full(hub_len, h))
    for d in rng.choice(n, dense_cols, replace=False) if dense_cols else []:
        rows.append(np.arange(n)); cols.append(np.full(n, d))     # completely dense columns of A
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsc()
    A.sum_duplicates(); A.data[:] = rng.integers(1, 4, len(A.data))
    bc, br = [], []
    for j in range(ncols):
        k = np.unique(np.concatenate([hubs[: max(1, nhubs - j)], rng.integers(0, n, deg)]))
        br.append(k); bc.append(np.full(len(k), j))
    br, bc = np.concatenate(br), np.concatenate(bc)
    B = sp.coo_matrix((rng.integers(1, 4, len(br)).astype(np.float64), (br, bc)), shape=(n, ncols)).tocsc()
    return A, B


def _run(gba, A, B, sr, M=None, comp=False, struct=True):
    dA, dB = A.to_device(), B.to_device()
    dM = M.to_device() if M is not None else None
    out, rep, _ = gba.AxB_saxpy3(dM, comp, struct, dA, dB, sr)
    got = out.to_host()
    for d in (dA, dB, dM, out):
        if d is not None:
            d.free()
    return got, rep


def _mat(S, t=Type.FP64):
    return Matrix.from_scipy(S, type=t, by_row=False)


def test_hashed_and_direct_panels(gba):
    """deg > one search step, hub runs longer than a wave, sparse and dense index ranges."""
    As, Bs = _heavy_pair(60000, 6, 5000, 12, 20000, 3, seed=1)
    A, B = _mat(As), _mat(Bs)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert rep.bin_vectors[3] == 6 and rep.panels > 6 and np.diff(want.p).min() > 8192
    assert_same(got, want, exact=True)


def _sorted_within_vectors(M):
    cnt = np.diff(M.p)
    inner = np.ones(len(M.i), bool); inner[M.p[:-1][cnt > 0]] = False
    return bool(np.all(np.diff(M.i)[inner[1:]] > 0))


@pytest.mark.parametrize("ranked", ["1", "0"])
def test_ranked_and_hashed_numeric_sweeps(gba, monkeypatch, ranked):
    """The numeric sweep ranks products in the pattern the symbolic sweep wrote (default) or, without that pattern
    (GBA_PANEL_RANKED=0, or no room for it), hashes them: same C either way; the ranked sweep leaves the heavy
    vectors in index order."""
    monkeypatch.setenv("GBA_PANEL_RANKED", ranked)
    As, Bs = _heavy_pair(60000, 6, 5000, 12, 20000, 3, seed=21)
    A, B = _mat(As), _mat(Bs)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert rep.bin_vectors[3] == 6 and rep.panels > 6
    assert_same(got, want, exact=True)
    if ranked == "1":
        assert _sorted_within_vectors(got)
    # masked: entries the mask rules out are absent from the pattern, their products find no bit
    rng = np.random.default_rng(22)
    Ms = sp.random(60000, 6, density=0.4, random_state=rng, format="csc"); Ms.data[:] = 1
    M = _mat(Ms, Type.BOOL)
    for comp in (False, True):
        got, rep = _run(gba, A, B, "MIN_PLUS_FP64", M, comp, True)
        assert_same(got, oracle.saxpy3(M, comp, True, A, B, "MIN_PLUS_FP64"), exact=True, what=f"comp={comp}")


def test_sparse_stretches_are_hashed_dense_ones_ranked(gba):
    """Very sparse heavy vectors next to dense ones: a stretch of 6144 entries that spans more indices than the rank
    table covers is hashed, narrower ones are ranked; both kinds alternate inside one vector."""
    n = 4_000_000
    As, Bs = _heavy_pair(n, 24, 12000, 2, 6000, 1, seed=23)        # (enough entries in B for the panel path)
    Bs = Bs.tolil(); Bs[n - 1, 0] = 1; Bs = Bs.tocsc()                 # one neighbour at the far end
    dense = sp.coo_matrix((np.ones(40000), (np.arange(1_000_000, 1_040_000), np.zeros(40000, int))), shape=(n, n)).tocsc()
    As = (As + dense).tocsc()                                          # A(:,0): a dense run of 40000 indices
    Bs = Bs.tolil(); Bs[0, :] = 1; Bs = Bs.tocsc()                     # every vector names it
    A, B = _mat(As), _mat(Bs)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    cnt = np.diff(want.p)
    assert rep.bin_vectors[3] == 24 and cnt.min() > 4096 + 40000
    assert_same(got, want, exact=True)


def test_fully_dense_vectors_use_direct_panels(gba):
    As, Bs = _heavy_pair(20000, 3, 300, 4, 5000, 2, seed=2, dense_cols=2)
    dense_k = np.flatnonzero(np.diff(As.indptr) == 20000)
    extra = sp.coo_matrix((np.ones(len(dense_k) * 3), (np.tile(dense_k, 3), np.repeat(np.arange(3), len(dense_k)))),
                          shape=Bs.shape)
    Bs = (Bs + extra).tocsc()                                     # every B(:,j) names the dense columns of A
    A, B = _mat(As), _mat(Bs)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert np.all(np.diff(want.p) == 20000)
    assert_same(got, want, exact=True)


def test_several_symbolic_panels(gba):
    """vlen above the width of one symbolic bitmap panel: boundaries carry over panel edges."""
    n = 1_700_000
    As, Bs = _heavy_pair(n, 3, 30000, 6, 150000, 2, seed=3)
    A, B = _mat(As), _mat(Bs)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert rep.bin_vectors[3] == 3 and np.diff(want.p).min() > 100000
    assert_same(got, want, exact=True)


@pytest.mark.parametrize("sr", ["PLUS_TIMES_FP64", "MIN_FIRSTJ_INT64"])
def test_hub_vectors_are_cut_into_several_tasks(gba, sr):
    """Vectors of > 4e6 multiply-adds are split at recorded boundaries and emitted by several
    workgroups, each at its own offset inside C(:,j)."""
    t = Type.FP64 if sr.endswith("FP64") else Type.INT64
    As, Bs = _heavy_pair(400_000, 2, 3000, 48, 220_000, 1, seed=11)
    A, B = _mat(As, t), _mat(Bs, t)
    dA, dB = A.to_device(), B.to_device()
    mw, fl = gba.AxB_saxpy3_flopcount(None, False, dA, dB)
    dA.free(); dB.free()
    assert np.diff(fl).max() > 2 * 4_000_000
    got, rep = _run(gba, A, B, sr)
    want = oracle.saxpy3(None, False, False, A, B, sr)
    assert_same(got, want, exact=True, what=sr)


def _few_long_neighbours(n, degs, hub_len, seed, nhubs=520):
    """B(:,j) names degs[j] columns of A, all of them long: heavy vectors with few neighbours, whose
    runs inside one panel are long (the searches are shared by groups of lanes)."""
    rng = np.random.default_rng(seed)
    hubs = rng.choice(n, nhubs, replace=False)
    rows, cols = [], []
    for h in hubs:
        ln = int(hub_len * rng.uniform(0.3, 1.0))
        rows.append(rng.choice(n, ln, replace=False)); cols.append(np.full(ln, h))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((rng.integers(1, 4, len(rows)).astype(np.float64), (rows, cols)), shape=(n, n)).tocsc()
    br = [np.sort(rng.choice(hubs, d, replace=False)) for d in degs]
    bc = [np.full(len(k), j) for j, k in enumerate(br)]
    br, bc = np.concatenate(br), np.concatenate(bc)
    B = sp.coo_matrix((rng.integers(1, 4, len(br)).astype(np.float64), (br, bc)), shape=(n, len(degs))).tocsc()
    return A, B


@pytest.mark.parametrize("sr", ["PLUS_TIMES_FP64", "MIN_FIRSTJ_INT64"])
def test_heavy_vectors_with_few_neighbours(gba, sr):
    """1 to 512 neighbours per vector: every group size of the shared search, and the per-thread search above."""
    t = Type.FP64 if sr.endswith("FP64") else Type.INT64
    degs = [1, 2, 3, 5, 8, 9, 16, 17, 31, 33, 64, 100, 128, 200, 256, 400, 512, 513, 520]
    As, Bs = _few_long_neighbours(150_000, degs, 40_000, seed=5)
    A, B = _mat(As, t), _mat(Bs, t)
    got, rep = _run(gba, A, B, sr)
    want = oracle.saxpy3(None, False, False, A, B, sr)
    assert rep.bin_vectors[3] == len(degs)
    assert_same(got, want, exact=True, what=sr)


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape", ["many_short_lists", "few_long_lists"])
def test_vectors_below_one_stretch_are_counted_in_one_walk(gba, shape, masked):
    """Vectors of 2049 .. 8447 multiply-adds go through the panel kernels but cannot reach a boundary: the symbolic
    kernel counts them in one walk with a hash set instead of bitmap sweeps, and records no word list.  Indices
    collide often (few rows), and a vector just above the limit takes the sweeps next to them."""
    if shape == "many_short_lists":
        degs = [300, 450, 513, 600, 690] * 8                # (more than 16 384 entries in B: not the one-launch road of small products)
        As, Bs = _few_long_neighbours(5_000, degs, 12, seed=21, nhubs=700)
    else:
        degs = [2, 3, 4, 5, 9]
        As, Bs = _few_long_neighbours(9_000, degs, 1_900, seed=22, nhubs=60)
    A, B = _mat(As), _mat(Bs)
    dA, dB = A.to_device(), B.to_device()
    mw, fl = gba.AxB_saxpy3_flopcount(None, False, dA, dB)
    dA.free(); dB.free()
    per = np.diff(fl)
    assert ((per > 2048) & (per < 8448)).sum() >= 3, per
    M = None
    if masked:
        Ms = sp.random(As.shape[0], len(degs), density=0.5, random_state=np.random.default_rng(23), format="csc")
        Ms.data[:] = 1
        M = _mat(Ms, Type.BOOL)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64", M, False, True)
    want = oracle.saxpy3(M, False, True, A, B, "PLUS_TIMES_FP64")
    if not masked:
        assert rep.bin_vectors[3] >= 3, list(rep.bin_vectors)       # they did take the panel kernels
    assert_same(got, want, exact=True, what=f"{shape} masked={masked} flops={per.tolist()}")


def test_few_neighbours_split_into_tasks(gba):
    """A vector of more than 4e6 multiply-adds is split, and a later task starts its sweep in the middle of the
    lists: the first search is for the task's first index."""
    degs = [6, 12, 40]
    As, Bs = _few_long_neighbours(1_200_000, degs, 1_000_000, seed=6, nhubs=40)
    A, B = _mat(As), _mat(Bs)
    dA, dB = A.to_device(), B.to_device()
    mw, fl = gba.AxB_saxpy3_flopcount(None, False, dA, dB)
    dA.free(); dB.free()
    assert np.diff(fl).max() > 2 * 4_000_000 and np.diff(fl)[1] > 4_000_000
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    assert_same(got, want, exact=True)


def test_few_neighbours_masked(gba):
    degs = [6, 40, 250]
    As, Bs = _few_long_neighbours(100_000, degs, 30_000, seed=7)
    A, B = _mat(As), _mat(Bs)
    rng = np.random.default_rng(8)
    Ms = sp.random(100_000, len(degs), density=0.3, random_state=rng, format="csc")
    Ms.data[:] = 1
    M = _mat(Ms, Type.BOOL)
    for comp in (False, True):
        got, rep = _run(gba, A, B, "PLUS_TIMES_FP64", M, comp, True)
        want = oracle.saxpy3(M, comp, True, A, B, "PLUS_TIMES_FP64")
        assert_same(got, want, exact=True, what=f"comp={comp}")


@pytest.mark.parametrize("sr,t", [
    ("PLUS_TIMES_INT8", Type.INT8), ("PLUS_TIMES_UINT16", Type.UINT16), ("PLUS_TIMES_FP32", Type.FP32),
    ("MIN_PLUS_FP64", Type.FP64), ("MAX_MIN_INT32", Type.INT32), ("LOR_LAND_BOOL", Type.BOOL),
    ("PLUS_PAIR_INT64", Type.INT64), ("MIN_FIRSTJ_INT64", Type.INT64), ("MAX_SECONDI_INT32", Type.INT32),
    ("BXOR_BAND_UINT32", Type.UINT32), ("TIMES_PLUS_INT64", Type.INT64), ("PLUS_ISGE_FP64", Type.FP64),
])
def test_semirings_on_heavy_vectors(gba, sr, t):
    As, Bs = _heavy_pair(30000, 3, 2500, 8, 9000, 3, seed=4)
    A, B = _mat(As, t), _mat(Bs, t)
    got, rep = _run(gba, A, B, sr)
    assert rep.bin_vectors[3] == 3
    assert_same(got, oracle.saxpy3(None, False, False, A, B, sr), exact=True, what=sr)


def test_any_monoid_on_heavy_vectors(gba):
    As, Bs = _heavy_pair(30000, 2, 2500, 8, 9000, 3, seed=5)
    A, B = _mat(As, Type.INT64), _mat(Bs, Type.INT64)
    got, _ = _run(gba, A, B, "ANY_FIRSTJ_INT64")
    want = oracle.saxpy3(None, False, False, A, B, "ANY_FIRSTJ_INT64")
    assert_same(got, want, check_values=False)
    # each value k must be a witness: A(i,k) and B(k,j) both present
    r, c, v = got.coo()
    Ap, Bp = (As != 0).tocsr(), (Bs != 0).tocsc()
    for i, j, k in zip(r[::997], c[::997], v[::997]):
        assert Ap[i, k] and Bp[k, j]


@pytest.mark.parametrize("comp", [False, True])
@pytest.mark.parametrize("mkind", ["sparse", "bitmap"])
def test_masked_heavy_vectors(gba, mkind, comp):
    n = 30000
    As, Bs = _heavy_pair(n, 3, 2500, 8, 9000, 3, seed=6)
    A, B = _mat(As), _mat(Bs)
    rng = np.random.default_rng(8)
    if mkind == "sparse":
        Ms = sp.random(n, 3, density=0.5, random_state=rng, format="csc")
        Ms.data = rng.integers(0, 2, len(Ms.data)).astype(np.float64)
        M = Matrix.from_scipy(Ms, type=Type.INT8, by_row=False)
    else:
        M = Matrix.dense(rng.integers(0, 2, (n, 3)).astype(np.int8), present=rng.random((n, 3)) < 0.7, by_row=False)
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64", M=M, comp=comp, struct=False)
    want = oracle.saxpy3(M, comp, False, A, B, "PLUS_TIMES_FP64")
    assert np.diff(want.p).max() > 4096
    assert_same(got, want, exact=True)


@pytest.mark.parametrize("sr,t", [
    ("PLUS_TIMES_FP64", Type.FP64), ("PLUS_TIMES_INT8", Type.INT8), ("MIN_PLUS_FP64", Type.FP64),
    ("ANY_PAIR_BOOL", Type.BOOL), ("MAX_SECONDI_INT64", Type.INT64), ("LOR_LAND_BOOL", Type.BOOL),
])
def test_team_kernels_for_a_few_hub_vectors(gba, sr, t):
    """Up to 32 vectors of > 2e5 multiply-adds each are computed by the whole grid (dense accumulator
    + bitmap, monoid atomics); lighter vectors of the same product keep their own paths."""
    As, Bs = _heavy_pair(60000, 5, 5000, 12, 30000, 3, seed=13)
    light = sp.random(60000, 40, density=2e-4, random_state=np.random.default_rng(3), format="csc")
    light.data[:] = 2
    Bs = sp.hstack([Bs, light]).tocsc()
    A, B = _mat(As, t), _mat(Bs, t)
    got, rep = _run(gba, A, B, sr)
    want = oracle.saxpy3(None, False, False, A, B, sr)
    assert rep.bin_vectors[3] == 0 and np.diff(want.p)[:5].min() > 8192      # hub vectors were not binned
    assert_same(got, want, exact=(not sr.startswith("ANY")), check_values=(not sr.startswith("ANY")), what=sr)


@pytest.mark.parametrize("comp", [False, True])
def test_team_kernels_masked_frontier_step(gba, comp):
    """One output vector with a bitmap mask: the BFS/SSSP frontier shape."""
    n = 50000
    As, Bs = _heavy_pair(n, 1, 4000, 10, 30000, 3, seed=14)
    A, B = _mat(As), _mat(Bs)
    rng = np.random.default_rng(15)
    M = Matrix.dense(np.ones((n, 1), np.int8), present=rng.random((n, 1)) < 0.4, by_row=False)
    got, rep = _run(gba, A, B, "MIN_PLUS_FP64", M=M, comp=comp, struct=True)
    want = oracle.saxpy3(M, comp, True, A, B, "MIN_PLUS_FP64")
    assert rep.bin_vectors[3] == 0 and want.nvals > 4096
    assert_same(got, want, exact=True)


def test_heavy_vectors_sorted_on_request(gba):
    """do_sort (GrB_wait on the jumbled result): vectors longer than the LDS sort take the device
    radix sort; the result must be the oracle's sorted C entry for entry, in order."""
    As, Bs = _heavy_pair(60000, 4, 5000, 12, 20000, 3, seed=12)
    A, B = _mat(As), _mat(Bs)
    dA, dB = A.to_device(), B.to_device()
    out, rep, _ = gba.AxB_saxpy3(None, False, True, dA, dB, "PLUS_TIMES_FP64", do_sort=1)
    got = out.to_host()
    for d in (dA, dB, out):
        d.free()
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64", wait=True)
    assert not got.jumbled and np.diff(want.p).min() > 4096
    assert np.array_equal(got.p, want.p) and np.array_equal(got.i, want.i) and np.array_equal(got.x, want.x)


def test_jumbled_A_takes_the_dense_accumulator_path(gba):
    """The panel sweep needs sorted A(:,k); a jumbled A must still give the same C."""
    As, Bs = _heavy_pair(30000, 2, 2500, 8, 9000, 3, seed=7)
    A, B = _mat(As), _mat(Bs)
    want = oracle.saxpy3(None, False, False, A, B, "PLUS_TIMES_FP64")
    rng = np.random.default_rng(1)
    for k in range(A.vdim):                                       # shuffle inside every vector
        lo, hi = A.p[k], A.p[k + 1]
        perm = rng.permutation(hi - lo)
        A.i[lo:hi] = A.i[lo:hi][perm]; A.x[lo:hi] = A.x[lo:hi][perm]
    A.jumbled = True
    got, rep = _run(gba, A, B, "PLUS_TIMES_FP64")
    assert rep.panels == 0
    assert_same(got, want, exact=True)


def test_rmat15_square_checksum_and_pattern(gba):
    """RMAT-15 squared (tens of millions of entries in C, many heavy vectors): checksum identity,
    entry count and no duplicates."""
    A = gen.rmat(15, 16, seed=42, by_row=False, values="int")
    dA = A.to_device()
    out, rep, _ = gba.AxB_saxpy3(None, False, True, dA, dA, "PLUS_TIMES_FP64")
    assert rep.panels > 1000
    got = out.to_host(); out.free(); dA.free()
    S = sp.csc_matrix((A.x, A.i, A.p), shape=(A.vlen, A.vdim))
    colsum = np.asarray(S.sum(axis=0)).ravel(); rowsum = np.asarray(S.sum(axis=1)).ravel()
    assert float(got.x.sum()) == float((colsum * rowsum).sum())
    pat = (S != 0).astype(np.int64)
    want_nnz = (pat @ pat).nnz
    assert got.nvals == want_nnz == got.p[-1]
    vec = np.repeat(np.arange(got.vdim), np.diff(got.p))
    assert len(np.unique(vec * got.vlen + got.i)) == got.nvals


def test_rmat18_front_end_in_row_chunks_properties(gba):
    """The bench's workload one size below the largest the host can still check entry by entry: C = A*A at
    RMAT-18 (2.9e9 multiply-adds, 1.3e9 entries) through gba_mxm in flop-balanced row chunks, sorted on
    request.  Size-independent properties per chunk: every row's sum equals (A (A 1)) for integer values
    (exact in fp64), indices strictly increase inside a row (no duplicates, in range), row counts add up to
    the multiply-add bound, and the entry count of the whole product matches the known value."""
    from graphblas_amd import DeviceMatrix
    from graphblas_amd.binding import gba_matrix
    from graphblas_amd.dist import row_flops, chunks_by_flops
    import graphblas_amd as g
    A = gen.rmat(18, 16, seed=42, type=Type.FP64, values="int")            # CSR
    n = A.nrows
    S = sp.csr_matrix((A.x, A.i, A.p), shape=(n, n))
    want_rowsum = S @ (S @ np.ones(n))
    rf = row_flops(A.p, A.i)
    prefix = np.concatenate([[0], np.cumsum(rf)]).astype(np.int64)
    dA = A.to_device()
    total = 0
    for lo, hi in chunks_by_flops(prefix, 0, n, 400_000_000):
        rows = dA.row_slice(lo, hi)
        C = DeviceMatrix.empty(hi - lo, n, Type.FP64)
        g.mxm(C, None, None, "PLUS_TIMES_FP64", rows, dA, g.descriptor(do_sort=1))
        got = C.to_host(); C.free()
        assert not got.jumbled and got.nzombies == 0
        cnt = np.diff(got.p)
        assert np.all(cnt <= rf[lo:hi]) and np.all((cnt > 0) == (rf[lo:hi] > 0))
        assert got.i.min() >= 0 and got.i.max() < n
        inner = np.ones(len(got.i), bool); inner[got.p[:-1][cnt > 0]] = False    # first entry of each row
        assert np.all(np.diff(got.i)[inner[1:]] > 0), f"rows {lo}:{hi}: indices not strictly increasing"
        sums = np.add.reduceat(got.x, got.p[:-1][cnt > 0])
        assert np.array_equal(sums, want_rowsum[lo:hi][cnt > 0]), f"rows {lo}:{hi}: row sums differ"
        total += int(got.p[-1])
    dA.free()
    assert total == 1282259348            # nnz (A*A) at RMAT-18, edge factor 16, seed 42 (the bench reports the same)
