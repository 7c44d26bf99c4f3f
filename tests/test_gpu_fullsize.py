"""The configurations the path is quoted on, at full size, checked through properties that do not need a full
CPU replay: SpMV at RMAT-22 against a double-precision product with a stated bound, the triangle count of
RMAT-20 against the oracle's total and scipy on sampled rows, BFS at RMAT-22 level by level, and SSSP
(Bellman-Ford over MIN_PLUS_FP64) at RMAT-20 bit for bit against the same iteration in numpy."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.csgraph as csgraph

import graphblas_amd as g
from graphblas_amd import Matrix, Type, gen
from oracle import oracle

pytestmark = pytest.mark.gpu


def _free(*ds):
    for d in ds:
        d.free()


def test_spmv_rmat22_fp32_both_forms_within_the_rounding_bound(gba):
    """y = A*x (dot4, gather) and y' = x'*A (saxpy4 through destination bins) with PLUS_TIMES_FP32 at RMAT-22.
    Reference: the same fp32 inputs multiplied in fp64.  Bound per entry: a sum of m positive fp32 products in
    any order is within (m + 1) * 2^-24 * sum|a*x| of the exact sum (one rounding per product, at most m - 1 per
    addition chain); the values are positive, so sum|a*x| is the reference itself."""
    A = gen.rmat(22, 16, seed=42, type=Type.FP32)                       # CSR
    n = A.nrows
    x = (np.random.default_rng(1).random(n) + 0.5).astype(np.float32)
    S = A.to_scipy().astype(np.float64)
    dA, dx = A.to_device(), Matrix.full_vector(x).to_device()
    eps = 2.0 ** -24
    for form in ("mxv", "vxm"):
        dy = Matrix.full_vector(np.zeros(n, np.float32)).to_device()
        if form == "mxv":
            rep = g.mxv(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, dx)
            ref, terms = S @ x.astype(np.float64), np.diff(S.indptr)
        else:
            rep = g.vxm(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dx, dA)
            ref, terms = S.T @ x.astype(np.float64), np.bincount(S.indices, minlength=n)
        assert g.Method.NAMES[rep.method] == ("dot4" if form == "mxv" else "saxpy4")
        y = dy.to_host().x.astype(np.float64)
        bound = (terms + 1) * eps * ref + 1e-30
        worst = float(np.max(np.abs(y - ref) / bound))
        assert worst <= 1.0, (form, worst)
        assert np.array_equal(y == 0, ref == 0), form                    # empty rows / columns stay untouched
        dy.free()
    _free(dA, dx)


def test_triangle_count_rmat20_exact(gba):
    """C<L,struct> = L*L' with PLUS_PAIR_INT64 (dot3): the total equals the oracle's, and every C(i,j) of 256
    sampled rows equals scipy's integer product (L[i,:] * L') .* L[i,:]."""
    A = gen.rmat(20, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    L = gen.tril_strict(A)
    n = L.nrows
    dL = L.to_device()
    C = g.DeviceMatrix.empty(n, n, Type.INT64)
    rep = g.mxm(C, dL, None, "PLUS_PAIR_INT64", dL, dL, g.descriptor("ST1"))
    assert g.Method.NAMES[rep.method] == "dot3"
    Ch = C.to_host()
    live = Ch.i >= 0
    want = oracle.dot3(L, True, L, L, "PLUS_PAIR_INT64")
    wlive = want.i >= 0
    assert int(Ch.x[live].sum()) == int(want.x[wlive].sum())
    assert np.array_equal(Ch.p, want.p) and np.array_equal(Ch.i, want.i)
    assert np.array_equal(Ch.x[live], want.x[wlive])
    S = L.to_scipy().astype(np.int64); S.data[:] = 1
    rows = np.arange(0, n, 4096)
    R = S[rows]
    ref = (R @ S.T.tocsr()).multiply(R).tocsr()
    ref.sort_indices()
    for q, i in enumerate(rows):
        seg = slice(Ch.p[i], Ch.p[i + 1])
        keep = Ch.i[seg] >= 0
        assert np.array_equal(Ch.i[seg][keep], ref.indices[ref.indptr[q]:ref.indptr[q + 1]]), i
        assert np.array_equal(Ch.x[seg][keep], ref.data[ref.indptr[q]:ref.indptr[q + 1]]), i
    _free(dL, C)


def test_bfs_rmat22_levels_and_parents(gba):
    """Level-synchronous BFS, q<!pi,struct,replace> = q'*A with ANY_SECONDI_INT64 per level: the set found at
    every level is exactly the set at that distance (scipy), and every parent is a frontier vertex of the level
    before with an edge to its child.  (Which parent ANY picks is free; the reference makes no promise either.)"""
    A = gen.rmat(22, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    n = A.nrows
    S = A.to_scipy()
    src = int(np.argmax(np.diff(A.p)))
    dist = csgraph.shortest_path(S, method="D", unweighted=True, indices=src)
    dA = A.to_device()
    pi_b = np.zeros(n, np.int8); pi_x = np.zeros(n, np.int64)
    pi_b[src] = 1; pi_x[src] = src
    q = Matrix.sparse_vector(n, [src], np.array([src], np.int64))
    keys = None
    for level in range(64):
        dq = q.to_device()
        dpi = Matrix.bitmap_vector(pi_x, pi_b).to_device()
        dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
        g.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")
        r, _, parents = dw.to_host().coo()
        _free(dq, dpi, dw)
        want = np.flatnonzero(dist == level + 1)
        assert np.array_equal(np.sort(r), want), level
        if len(r) == 0:
            break
        order = np.argsort(r)
        r, parents = r[order], parents[order]
        assert np.all(dist[parents] == level), level
        # the edge parent -> child exists (about 2000 children per level): the child is in the parent's sorted row
        for k in range(0, len(r), max(len(r) // 2000, 1)):
            row = A.i[A.p[parents[k]]:A.p[parents[k] + 1]]
            j = int(np.searchsorted(row, r[k]))
            assert j < len(row) and row[j] == r[k], (level, int(parents[k]), int(r[k]))
        pi_b[r] = 1; pi_x[r] = parents
        if len(r) > n // 8:
            b = np.zeros(n, np.int8); x = np.zeros(n, np.int64); b[r] = 1; x[r] = parents
            q = Matrix.bitmap_vector(x, b)
        else:
            q = Matrix.sparse_vector(n, r, parents)
    assert int(pi_b.sum()) == int(np.isfinite(dist).sum())
    dA.free()


def test_bfs_rmat24_validates_as_a_breadth_first_tree(gba):
    """The same traversal at RMAT-24 (16.8M vertices, 2 x 2.6e8 directed edges after symmetrisation), checked the way
    Graph500 validates a BFS without a second traversal: (1) the source is its own parent at level 0; (2) every
    other reached vertex has a parent one level up with an edge to it; (3) the two ends of every edge are either
    both reached with levels at most one apart, or both unreached.  By (2) the parents form a path of level(v)
    edges from the source, so level(v) is at least the distance; by (3), walking a shortest path from the source,
    the level grows by at most one per edge, so it is at most the distance: the levels are the distances."""
    A = gen.rmat(24, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    n = A.nrows
    deg = np.diff(A.p)
    src = int(np.argmax(deg))
    dA = A.to_device()
    level_of = np.full(n, -1, np.int32); parent = np.full(n, -1, np.int64)
    level_of[src] = 0; parent[src] = src
    pi_b = np.zeros(n, np.int8); pi_x = np.zeros(n, np.int64)
    pi_b[src] = 1; pi_x[src] = src
    q = Matrix.sparse_vector(n, [src], np.array([src], np.int64))
    for level in range(1, 64):
        dq = q.to_device()
        dpi = Matrix.bitmap_vector(pi_x, pi_b).to_device()
        dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
        g.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")
        r, _, parents = dw.to_host().coo()
        _free(dq, dpi, dw)
        if len(r) == 0:
            break
        assert np.all(level_of[r] == -1), level                      # the mask kept visited vertices out
        level_of[r] = level; parent[r] = parents
        pi_b[r] = 1; pi_x[r] = parents
        if len(r) > n // 8:
            b = np.zeros(n, np.int8); x = np.zeros(n, np.int64); b[r] = 1; x[r] = parents
            q = Matrix.bitmap_vector(x, b)
        else:
            q = Matrix.sparse_vector(n, r, parents)
    dA.free()
    reached = level_of >= 0
    assert reached.sum() > n // 4 and level_of.max() >= 4
    kids = np.flatnonzero(reached); kids = kids[kids != src]
    assert np.all(level_of[parent[kids]] == level_of[kids] - 1)     # (2) one level up ...
    # ... with an edge to the child: the child is in the parent's sorted row (every child, in slabs)
    for lo in range(0, len(kids), 1 << 22):
        c = kids[lo:lo + (1 << 22)]; pr = parent[c]
        # position of c in row pr by a vectorised binary search over [A.p[pr], A.p[pr + 1])
        a, b = A.p[pr].copy(), A.p[pr + 1].copy()
        while True:
            open_ = a < b
            if not open_.any():
                break
            mid = (a + b) >> 1
            less = np.zeros(len(c), bool)
            less[open_] = A.i[mid[open_]] < c[open_]
            a = np.where(open_ & less, mid + 1, a)
            b = np.where(open_ & ~less, mid, b)
        assert np.all(a < A.p[pr + 1]) and np.all(A.i[np.minimum(a, len(A.i) - 1)] == c)
    # (3) over every edge, in slabs of rows
    for lo in range(0, n, 1 << 21):
        hi = min(lo + (1 << 21), n)
        e0, e1 = int(A.p[lo]), int(A.p[hi])
        lu = np.repeat(level_of[lo:hi], deg[lo:hi])
        lv = level_of[A.i[e0:e1]]
        assert np.all((lu >= 0) == (lv >= 0))
        assert np.all(np.abs(lu - lv) <= 1)


def test_sssp_rmat20_bellman_ford_bit_exact(gba):
    """d = min(d, d min.+ A) until it stops changing, MIN_PLUS_FP64 through gba_vxm with the MIN accumulator
    (saxpy4, in place).  Every candidate is one fp64 addition of the same two numbers on both sides and MIN is
    exact, so the distances agree bit for bit with the same synchronous iteration in numpy."""
    A = gen.rmat(20, 16, seed=42, type=Type.FP64)                       # CSR, weights uniform in (0, 1]
    n = A.nrows
    rows = np.repeat(np.arange(n), np.diff(A.p))
    src = int(np.argmax(np.diff(A.p)))
    d = np.full(n, np.inf); d[src] = 0.0
    dA = A.to_device()
    for it in range(64):
        dcur = Matrix.full_vector(d).to_device()
        dnew = Matrix.full_vector(d.copy()).to_device()
        rep = g.vxm(dnew, None, "MIN_FP64", "MIN_PLUS_FP64", dcur, dA)
        assert g.Method.NAMES[rep.method] == "saxpy4"
        got = dnew.to_host().x
        _free(dcur, dnew)
        want = d.copy()
        np.minimum.at(want, A.i, d[rows] + A.x)
        assert np.array_equal(got, want), it
        if np.array_equal(want, d):
            break
        d = want
    assert it > 2 and np.isfinite(d).sum() > n // 4
    dA.free()


def test_spmv_on_row_block_views(gba):
    """The row block a rank owns is a view that keeps absolute offsets (p[0] != 0).  y_I += A(I,:)*x through the
    tiled gather kernel and y' += x_I'*A(I,:) through the destination bins (more than 2^20 entries), with
    small integer values so that every sum is exact, against scipy on the same rows."""
    n, lo, hi = 1 << 17, 3_000, 125_000
    A = gen.rmat(17, 16, seed=5, type=Type.FP64, values="int")            # CSR
    S = A.to_scipy()[lo:hi]
    assert S.nnz > (1 << 20)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 3, n).astype(np.float64)
    xI = rng.integers(0, 3, hi - lo).astype(np.float64)
    dA = A.to_device()
    blk = dA.row_slice(lo, hi)
    dx, dxI = Matrix.full_vector(x).to_device(), Matrix.full_vector(xI).to_device()
    dy = Matrix.full_vector(np.ones(hi - lo)).to_device()
    rep = g.mxv(dy, None, "PLUS_FP64", "PLUS_TIMES_FP64", blk, dx)
    assert g.Method.NAMES[rep.method] == "dot4"
    assert np.array_equal(dy.to_host().x, 1.0 + S @ x)
    dz = Matrix.full_vector(np.ones(n)).to_device()
    rep = g.vxm(dz, None, "PLUS_FP64", "PLUS_TIMES_FP64", dxI, blk)
    assert g.Method.NAMES[rep.method] == "saxpy4"
    assert np.array_equal(dz.to_host().x, 1.0 + S.T @ xI)
    _free(dA, dx, dxI, dy, dz)
