"""The oracle against results of the reference library itself: entry counts, checksums, multiply counts,
triangle counts and frontier sizes that the reference produced with its built-in semirings
(PLUS_TIMES_FP64 through saxpy3, PLUS_PAIR_INT64 through dot3, ANY_SECONDI_INT64 through saxpy3/saxbit) on
RMAT inputs that gen.rmat_lcg rebuilds edge for edge (tests/golden/reference_run_checksums.json).
Every value of these inputs is a small integer, so each sum is exact and the comparison is equality."""
import json
import os

import numpy as np
import pytest

from graphblas_amd import Matrix, Type, gen
from oracle import oracle

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_run_checksums.json")))
SPGEMM = {c["scale"]: c for c in GOLDEN["spgemm_plus_times_fp64"]}
TRI = {c["scale"]: c for c in GOLDEN["triangle_count_plus_pair_int64"]}


def rounds_to(x, recorded):
    """recorded is x printed with four significant digits"""
    return float(f"{x:.3e}") == recorded


@pytest.mark.parametrize("scale", [10, 16])
def test_oracle_spgemm_equals_the_reference_run(scale):
    want = SPGEMM[scale]
    A = gen.rmat_lcg(scale, by_row=False)
    assert A.nvals == want["nnz_A"]
    C = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64")
    assert C.nvals == want["nnz_C"]
    assert float(C.x.sum()) == float(want["sum_C"])
    if "multiplies_rounded" in want:                 # the reference reports axbwork = multiplies + nnz (B)
        _, fl = oracle.saxpy3_flopcount(None, False, A, A)
        assert rounds_to(int(fl[-1]) - A.nvals, want["multiplies_rounded"])


def test_oracle_triangle_count_equals_the_reference_run():
    want = TRI[12]
    S = gen.rmat_lcg(12, symmetric=True, drop_diagonal=True, type=Type.INT64)
    L = gen.tril_strict(S)
    assert L.nvals == want["nnz_L"]
    C = oracle.dot3(L, True, L, L, "PLUS_PAIR_INT64")          # C<L,struct> = L*L'
    assert int(C.x[C.i >= 0].sum()) == want["triangles"]


def test_oracle_bfs_frontiers_equal_the_reference_run():
    """q<!visited,struct,replace> = q'*A, ANY_SECONDI_INT64, from vertex 0 of the symmetrised RMAT-18."""
    want = GOLDEN["bfs_any_secondi_int64"][0]
    A = gen.rmat_lcg(want["scale"], symmetric=True, type=Type.INT64, by_row=False)
    assert float(f"{A.nvals:.1e}") == want["nnz_symmetrised_rounded"]
    n = A.nrows
    visited = np.zeros(n, bool); visited[want["source"]] = True
    frontier = np.array([want["source"]], np.int64)
    sizes = [1]
    while len(sizes) < len(want["frontier_sizes"]):
        q = Matrix.sparse_vector(n, frontier, frontier.copy())
        seen = np.flatnonzero(visited)
        M = Matrix.sparse_vector(n, seen, np.ones(len(seen), np.int64))
        w = oracle.saxpy3(M, True, True, A, q, "ANY_SECONDI_INT64", wait=True)
        frontier = w.i[: w.p[-1]]
        assert not visited[frontier].any()
        assert np.isin(w.x[: w.p[-1]], q.i).all()      # SECONDI: the parent is a vertex of the frontier before
        visited[frontier] = True
        sizes.append(len(frontier))
    assert sizes == want["frontier_sizes"]
