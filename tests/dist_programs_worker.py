"""Worker of tests/test_zx_dist_programs.py: the row-block programs of bench_paths_dist.py (spmv, bfs, sssp, tricount)
with ONE rank and their collectives over RCCL (GBA_BENCH_FORCE_DIST=1), against the same programs without a
process group and against scipy.  Runs in a process of its own: torch is imported before the library is loaded."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402,F401  (before graphblas_amd loads its library)
import bench_paths_dist as bpd                  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    args = argparse.Namespace(scale=scale, iters=3, gpus=1)
    os.environ.pop("GBA_BENCH_FORCE_DIST", None)
    plain = bpd.Ranks(1)
    assert not plain.multi
    want = {w: bpd.run(w, args, plain) for w in bpd.PROGRAMS}
    os.environ["GBA_BENCH_FORCE_DIST"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    forced = bpd.Ranks(1)
    assert forced.multi and forced.dist.get_backend() == "nccl"
    got = {w: bpd.run(w, args, forced) for w in bpd.PROGRAMS}
    for w in bpd.PROGRAMS:
        assert got[w][1]["collective"] and not want[w][1]["collective"], w
    assert np.array_equal(got["spmv"][0], want["spmv"][0]), "spmv: x differs"
    assert np.array_equal(got["sssp"][0], want["sssp"][0]), "sssp: distances differ"
    assert np.array_equal(got["bfs"][0][0], want["bfs"][0][0]), "bfs: levels differ"
    assert got["tricount"][0] == want["tricount"][0], "tricount differs"
    print("one rank over RCCL equals the plain run", flush=True)

    # and against scipy on the host
    import scipy.sparse as sp
    from scipy.sparse import csgraph
    from graphblas_amd import gen, Type
    A = gen.rmat(scale, 16, seed=42, type=Type.FP32)
    S = sp.csr_matrix((A.x, A.i, A.p), shape=(A.nrows, A.nrows)).astype(np.float64)
    x = np.linspace(0.5, 1.5, A.nrows).astype(np.float32).astype(np.float64)
    for _ in range(args.iters):
        y = S @ x
        x = y / np.abs(y).max()
    assert np.allclose(got["spmv"][0], x, rtol=2e-4, atol=1e-7), "spmv: x is not the host iteration"
    G = gen.rmat(scale, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    Sg = sp.csr_matrix((np.ones(G.nvals), G.i, G.p), shape=(G.nrows, G.nrows))
    src = int(np.argmax(np.diff(G.p)))
    dist = csgraph.shortest_path(Sg, method="D", unweighted=True, indices=src)
    level_of, parent = got["bfs"][0]
    reached = np.isfinite(dist)
    assert np.array_equal(level_of >= 0, reached) and np.array_equal(level_of[reached], dist[reached].astype(np.int64))
    kids = np.nonzero(reached & (np.arange(G.nrows) != src))[0]
    assert np.all(level_of[parent[kids]] == level_of[kids] - 1), "bfs: a parent is not on the level before"
    assert np.all(np.asarray(Sg[kids, parent[kids]]).ravel() == 1), "bfs: a parent without an edge to its child"
    W = gen.rmat(scale, 16, seed=42, type=Type.FP64)
    Sw = sp.csr_matrix((W.x, W.i, W.p), shape=(W.nrows, W.nrows))
    sw = int(np.argmax(np.diff(W.p)))
    dw = csgraph.shortest_path(Sw, method="BF", directed=True, indices=sw)
    assert np.array_equal(np.isfinite(got["sssp"][0]), np.isfinite(dw))
    fin = np.isfinite(dw)
    assert np.allclose(got["sssp"][0][fin], dw[fin], rtol=1e-12, atol=0), "sssp: distances are not scipy's"
    L = gen.tril_strict(G)
    Sl = sp.csr_matrix((np.ones(L.nvals, np.int64), L.i, L.p), shape=(L.nrows, L.nrows))
    assert got["tricount"][0] == int((Sl @ Sl.T).multiply(Sl).sum()), "tricount is not scipy's"
    print("programs equal scipy", flush=True)
    forced.dist.destroy_process_group()


if __name__ == "__main__":
    main()
