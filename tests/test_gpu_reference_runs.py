"""The device path against results of the reference library itself (tests/golden/reference_run_checksums.json):
C = A*A with PLUS_TIMES_FP64 at RMAT-10/16/18 (entry count and checksum of C, multiply count), the triangle
count C<L,struct> = L*L' with PLUS_PAIR_INT64 at RMAT-12/18, and the frontier sizes of a BFS with
ANY_SECONDI_INT64 at RMAT-18, on inputs that gen.rmat_lcg rebuilds edge for edge.  All values are small
integers: every sum is exact in fp64 and the comparison is equality."""
import json
import os

import numpy as np
import pytest

import graphblas_amd as g
from graphblas_amd import Matrix, Type, gen

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_run_checksums.json")))


def rounds_to(x, recorded):
    return float(f"{x:.3e}") == recorded


@pytest.mark.parametrize("case", GOLDEN["spgemm_plus_times_fp64"], ids=lambda c: f"rmat{c['scale']}")
def test_spgemm_equals_the_reference_run(gba, case):
    A = gen.rmat_lcg(case["scale"])                                       # CSR, the reference's default
    n = A.nrows
    assert A.nvals == case["nnz_A"]
    dA = A.to_device()
    dC = g.DeviceMatrix.empty(n, n, Type.FP64)
    rep = g.mxm(dC, None, None, "PLUS_TIMES_FP64", dA, dA, None)
    assert g.Method.NAMES[rep.method] == case.get("method", "saxpy3")
    assert dC.view.nvals == case["nnz_C"]
    if "multiplies_rounded" in case:                  # the reference reports axbwork = multiplies + nnz (B)
        _, fl = g.AxB_saxpy3_flopcount(None, False, dA, dA)
        assert rounds_to(int(fl[-1]) - A.nvals, case["multiplies_rounded"])
    # checksum of C on the device: y = C * 1 (row sums), then the n sums on the host
    ones = Matrix.full_vector(np.ones(n)).to_device()
    dy = Matrix.full_vector(np.zeros(n)).to_device()
    g.mxv(dy, None, "PLUS_FP64", "PLUS_TIMES_FP64", dC, ones)
    assert float(dy.to_host().x.sum()) == float(case["sum_C"])
    for d in (dA, dC, ones, dy):
        d.free()


@pytest.mark.parametrize("case", GOLDEN["triangle_count_plus_pair_int64"], ids=lambda c: f"rmat{c['scale']}")
def test_triangle_count_equals_the_reference_run(gba, case):
    S = gen.rmat_lcg(case["scale"], symmetric=True, drop_diagonal=True, type=Type.INT64)
    L = gen.tril_strict(S)
    assert L.nvals == case["nnz_L"]
    dL = L.to_device()
    dC = g.DeviceMatrix.empty(L.nrows, L.ncols, Type.INT64)
    rep = g.mxm(dC, dL, None, "PLUS_PAIR_INT64", dL, dL, "ST1")
    assert g.Method.NAMES[rep.method] == case["method"]
    C = dC.to_host()
    assert int(C.x[C.i >= 0].sum()) == case["triangles"]
    dL.free(); dC.free()


def test_bfs_frontiers_equal_the_reference_run(gba):
    case = GOLDEN["bfs_any_secondi_int64"][0]
    A = gen.rmat_lcg(case["scale"], symmetric=True, type=Type.INT64)
    n, src = A.nrows, case["source"]
    dA = A.to_device()
    pi_b = np.zeros(n, np.int8); pi_x = np.zeros(n, np.int64)
    pi_b[src] = 1; pi_x[src] = src
    q = Matrix.sparse_vector(n, [src], np.array([src], np.int64))
    sizes, methods = [1], []
    while len(sizes) < len(case["frontier_sizes"]):
        dq, dpi = q.to_device(), Matrix.bitmap_vector(pi_x, pi_b).to_device()
        dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
        rep = g.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")
        methods.append(g.Method.NAMES[rep.method])
        r, _, parents = dw.to_host().coo()
        for d in (dq, dpi, dw):
            d.free()
        assert not pi_b[r].any()
        pi_b[r] = 1; pi_x[r] = parents
        sizes.append(len(r))
        # the reference's frontier became bitmap after the first level (its conform rule on a vector this full)
        if len(r) > n // 16:
            b = np.zeros(n, np.int8); x = np.zeros(n, np.int64); b[r] = 1; x[r] = parents
            q = Matrix.bitmap_vector(x, b)
        else:
            q = Matrix.sparse_vector(n, r, parents)
    assert sizes == case["frontier_sizes"]
    assert methods[0] == case["methods"][0]
    dA.free()
