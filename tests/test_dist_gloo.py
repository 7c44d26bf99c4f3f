"""The N > 1 path on CPU: two processes over gloo shard C = A*A by flop-balanced row blocks,
all-gather the right-hand operand from the blocks, multiply their own rows (here with the CPU
oracle standing in for the device kernels, which need a GPU) and together reproduce the
single-process product exactly."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from graphblas_amd.dist import row_flops, split_by_flops, chunks_by_flops  # noqa: E402


def test_partitions_cover_rows_once_and_balance_flops():
    from graphblas_amd import gen
    A = gen.rmat(12, 16, seed=42)
    rf = row_flops(A.p, A.i)
    # definition check on a few rows
    deg = np.diff(A.p)
    for r in (0, 1, 17, A.nrows - 1):
        assert rf[r] == deg[A.i[A.p[r]:A.p[r + 1]]].sum()
    prefix = np.concatenate([[0], np.cumsum(rf)])
    for nparts in (1, 2, 3, 8):
        cuts = split_by_flops(prefix, 0, A.nrows, nparts)
        assert cuts[0] == 0 and cuts[-1] == A.nrows and len(cuts) == nparts + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        loads = np.diff(prefix[cuts])
        heaviest_row = rf.max()
        assert loads.max() <= prefix[-1] / nparts + heaviest_row          # contiguous blocks: off by one row at most
    chunks = chunks_by_flops(prefix, 100, 3000, 2e5)
    assert chunks[0][0] == 100 and chunks[-1][1] == 3000
    assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    for lo, hi in chunks:
        assert hi > lo and (prefix[hi] - prefix[lo] <= 2e5 or hi == lo + 1)
    assert chunks_by_flops(prefix, 5, 5, 1e3) == []


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from graphblas_amd import gen, Matrix, Type
    from graphblas_amd.dist import RowBlockExchange
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A = gen.rmat(9, 8, seed=42, values="int")                     # CSR; integer values: sums are exact
    rf = row_flops(A.p, A.i)
    prefix = np.concatenate([[0], np.cumsum(rf)])
    part = split_by_flops(prefix, 0, A.nrows, world)
    lo, hi = part[rank], part[rank + 1]
    e0, e1 = int(A.p[lo]), int(A.p[hi])
    my_i, my_x = torch.from_numpy(A.i[e0:e1].copy()), torch.from_numpy(A.x[e0:e1].copy())
    Bi = torch.full((A.nvals,), -1, dtype=torch.int64)
    Bx = torch.zeros(A.nvals, dtype=torch.float64)
    RowBlockExchange(dist, torch, A.p, part, rank, torch.device("cpu"), torch.int64, torch.float64).gather(my_i, my_x, Bi, Bx)
    assert np.array_equal(Bi.numpy(), A.i) and np.array_equal(Bx.numpy(), A.x)
    # C(I,:) = A(I,:) * B with the gathered B.  CSR operands are the transposed CSC problem C' = B' A(I,:)'.
    Bfull = Matrix(A.nrows, A.nrows, Type.FP64, A.sparsity, False, p=A.p, i=Bi.numpy(), x=Bx.numpy())
    p = (A.p[lo:hi + 1] - A.p[lo]).astype(np.int64)
    Ablk = Matrix(A.nrows, hi - lo, Type.FP64, A.sparsity, False, p=p, i=A.i[e0:e1], x=A.x[e0:e1])
    Cblk = oracle.saxpy3(None, False, False, Bfull, Ablk, "PLUS_TIMES_FP64", wait=True)
    np.savez(os.path.join(outdir, f"c{rank}.npz"), p=Cblk.p, i=Cblk.i, x=Cblk.x, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo_reproduce_the_product(tmp_path):
    import torch.multiprocessing as mp
    from graphblas_amd import gen
    from oracle import oracle
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    A = gen.rmat(9, 8, seed=42, values="int")
    want = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64", wait=True)
    blocks = [np.load(tmp_path / f"c{r}.npz") for r in range(world)]
    assert blocks[0]["lo"] == 0 and blocks[0]["hi"] == blocks[1]["lo"] and blocks[1]["hi"] == A.nrows
    p = np.concatenate([blocks[0]["p"], blocks[1]["p"][1:] + blocks[0]["p"][-1]])
    i = np.concatenate([b["i"] for b in blocks]); x = np.concatenate([b["x"] for b in blocks])
    assert np.array_equal(p, want.p) and np.array_equal(i, want.i) and np.array_equal(x, want.x)


def _bfs_levels(S, src, rows=None, exchange=None, torch=None):
    """Level-synchronous BFS on the symmetric pattern S (scipy CSR): per level, w(I)<!visited(I)> = S(I,:) q with
    ANY_SECONDI -- any frontier neighbour is a valid parent; the smallest one is taken so that runs compare
    exactly -- for the rows I = rows this caller owns.  Returns the list of (indices, parents) per level."""
    n = S.shape[0]
    lo, hi = (0, n) if rows is None else rows
    visited = np.zeros(n, bool); visited[src] = True
    q_idx = np.array([src], np.int64)
    levels = []
    while len(q_idx):
        inq = np.zeros(n, bool); inq[q_idx] = True
        found, parents = [], []
        for i in range(lo, hi):
            if visited[i]:
                continue
            nb = S.indices[S.indptr[i]:S.indptr[i + 1]]
            nb = nb[inq[nb]]
            if len(nb):
                found.append(i); parents.append(nb.min())
        found, parents = np.array(found, np.int64), np.array(parents, np.int64)
        if exchange is not None:
            gi, gp = exchange.sparse(torch.from_numpy(found), torch.from_numpy(parents))
            found, parents = gi.numpy(), gp.numpy()
        levels.append((found, parents))
        visited[found] = True
        q_idx = found
    return levels


def _worker_vectors(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    from graphblas_amd import gen, Type
    from graphblas_amd.dist import VectorBlockExchange, FrontierExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # --- x <- A x, three times, y distributed by flop-balanced row blocks, x all-gathered
    A = gen.rmat(9, 8, seed=42, type=Type.FP32)
    n = A.nrows
    S = sp.csr_matrix((A.x, A.i, A.p), shape=(n, n))
    prefix = np.concatenate([[0], np.cumsum(np.diff(A.p))])          # SpMV work = entries per row
    part = split_by_flops(prefix, 0, n, world)
    lo, hi = part[rank], part[rank + 1]
    x = torch.from_numpy(np.linspace(0.5, 1.5, n).astype(np.float32))
    xchg = VectorBlockExchange(dist, torch, part, rank, torch.device("cpu"), torch.float32)
    for _ in range(3):
        mine = torch.from_numpy((S[lo:hi] @ x.numpy()).astype(np.float32))
        full = torch.empty(n, dtype=torch.float32)
        xchg.gather(mine, full)
        x = full / full.abs().max()
    # --- BFS from the heaviest vertex, the new frontier exchanged as index lists; SSSP-style dense MIN reduce
    G = gen.rmat(9, 8, seed=7, symmetric=True, drop_diagonal=True, values="ones")
    Sg = sp.csr_matrix((G.x, G.i, G.p), shape=(G.nrows, G.nrows))
    deg = np.diff(G.p)
    gpart = split_by_flops(np.concatenate([[0], np.cumsum(deg)]), 0, G.nrows, world)
    fx = FrontierExchange(dist, torch, rank, world, torch.device("cpu"))
    levels = _bfs_levels(Sg, int(np.argmax(deg)), rows=(gpart[rank], gpart[rank + 1]), exchange=fx, torch=torch)
    d = torch.full((G.nrows,), float("inf"), dtype=torch.float64)
    d[gpart[rank]:gpart[rank + 1]] = torch.arange(gpart[rank], gpart[rank + 1], dtype=torch.float64) % 7
    fx.dense_min(d)
    b = torch.zeros(G.nrows, dtype=torch.int8); b[gpart[rank]:gpart[rank + 1]:3] = 1
    fx.dense_or(b)
    np.savez(os.path.join(outdir, f"v{rank}.npz"), x=x.numpy(), d=d.numpy(), b=b.numpy(),
             **{f"i{k}": lv[0] for k, lv in enumerate(levels)}, **{f"p{k}": lv[1] for k, lv in enumerate(levels)},
             nlevels=len(levels), gpart=np.array(gpart))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_spmv_iterations_and_bfs_levels(tmp_path):
    """x <- A x with the all-gather of x, and BFS with the frontier exchange, on two ranks over gloo: every rank ends
    with exactly the single-process result."""
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    from graphblas_amd import gen, Type
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_vectors, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    A = gen.rmat(9, 8, seed=42, type=Type.FP32)
    S = sp.csr_matrix((A.x, A.i, A.p), shape=(A.nrows, A.nrows))
    x = np.linspace(0.5, 1.5, A.nrows).astype(np.float32)
    for _ in range(3):
        y = (S @ x).astype(np.float32)
        x = y / np.abs(y).max()
    G = gen.rmat(9, 8, seed=7, symmetric=True, drop_diagonal=True, values="ones")
    Sg = sp.csr_matrix((G.x, G.i, G.p), shape=(G.nrows, G.nrows))
    want_levels = _bfs_levels(Sg, int(np.argmax(np.diff(G.p))))
    assert len(want_levels) > 3
    for r in range(world):
        got = np.load(tmp_path / f"v{r}.npz")
        assert np.array_equal(got["x"], x), f"rank {r}: x after three products differs"
        assert int(got["nlevels"]) == len(want_levels)
        for k, (wi, wp) in enumerate(want_levels):
            assert np.array_equal(got[f"i{k}"], wi) and np.array_equal(got[f"p{k}"], wp), (r, k)
        gp = got["gpart"]
        assert np.array_equal(got["d"], np.arange(G.nrows) % 7)
        wb = np.zeros(G.nrows, np.int8)
        for q in range(world):
            wb[gp[q]:gp[q + 1]:3] = 1
        assert np.array_equal(got["b"], wb)


def _worker_triangles(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from graphblas_amd import gen, Matrix, Type
    from graphblas_amd.dist import ScalarReduce
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A = gen.rmat(9, 8, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    L = gen.tril_strict(A)                                        # CSR
    # rows of the mask split by the work of the masked dot products: sum over L(i,j) of |L(i,:)| + |L(j,:)|
    deg = np.diff(L.p)
    rows = np.repeat(np.arange(L.nrows), deg)
    work = np.bincount(rows, weights=(deg[rows] + deg[L.i]).astype(np.float64), minlength=L.nrows)
    prefix = np.concatenate([[0], np.cumsum(work)])
    part = split_by_flops(prefix, 0, L.nrows, world)
    lo, hi = part[rank], part[rank + 1]
    e0, e1 = int(L.p[lo]), int(L.p[hi])
    p = (L.p[lo:hi + 1] - L.p[lo]).astype(np.int64)
    Lblk = Matrix(L.vlen, hi - lo, Type.INT64, L.sparsity, L.is_csc, p=p, i=L.i[e0:e1], x=L.x[e0:e1])
    # C(I,:)<L(I,:)> = L(I,:) * L'.  In the method's terms (vectors = rows): C_I<M> = A_I' B_I with A_I = L, B_I = M = L(I,:)
    C = oracle.dot3(Lblk, True, L, Lblk, "PLUS_PAIR_INT64")
    mine = int(C.x[C.i >= 0].sum())
    total = ScalarReduce(dist, torch, world, torch.device("cpu")).sum_int64(mine)
    np.savez(os.path.join(outdir, f"t{rank}.npz"), mine=mine, total=total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_count_the_triangles_of_the_whole_graph(tmp_path):
    """dot3 sharded by row blocks of the mask with L replicated: the per-rank sums add up to the one-process count."""
    import torch.multiprocessing as mp
    from graphblas_amd import gen, Type
    from oracle import oracle
    world, port = 2, 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_triangles, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    A = gen.rmat(9, 8, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    L = gen.tril_strict(A)
    C = oracle.dot3(L, True, L, L, "PLUS_PAIR_INT64")
    want = int(C.x[C.i >= 0].sum())
    got = [np.load(tmp_path / f"t{r}.npz") for r in range(world)]
    assert all(int(g["total"]) == want for g in got)
    assert sum(int(g["mine"]) for g in got) == want and all(int(g["mine"]) > 0 for g in got)


def _worker_uneven(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from graphblas_amd.dist import all_gather_in_place, FrontierExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts = [5, 0, 3]
    offs = [0, 5, 5]
    mine = torch.arange(counts[rank], dtype=torch.int64) + 100 * rank
    out = torch.full((8,), -1, dtype=torch.int64)
    all_gather_in_place(dist, torch, mine, out, offs, counts, rank)
    fx = FrontierExchange(dist, torch, rank, world, torch.device("cpu"))
    gi, gv = fx.sparse(mine, mine.to(torch.float64) * 0.5)
    np.savez(os.path.join(outdir, f"u{rank}.npz"), out=out.numpy(), gi=gi.numpy(), gv=gv.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_pieces_of_unequal_length_land_in_place_on_three_ranks(tmp_path):
    """The exact-size all-gather with an empty piece in the middle: every rank ends with the concatenation, nothing
    is padded, and the frontier exchange built on it returns the same lists everywhere."""
    import torch.multiprocessing as mp
    world, port = 3, 35500 + (os.getpid() % 2000)
    mp.spawn(_worker_uneven, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = np.array([0, 1, 2, 3, 4, 200, 201, 202], np.int64)
    for r in range(world):
        got = np.load(tmp_path / f"u{r}.npz")
        assert np.array_equal(got["out"], want)
        assert np.array_equal(got["gi"], want) and np.array_equal(got["gv"], want * 0.5)


def _worker_panels(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from graphblas_amd import gen, Matrix, Type
    from graphblas_amd.dist import ColumnPanelExchange
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A = gen.rmat(9, 8, seed=42, values="int")
    n = A.nrows
    rf = row_flops(A.p, A.i)
    part = split_by_flops(np.concatenate([[0], np.cumsum(rf)]), 0, n, world)
    lo, hi = part[rank], part[rank + 1]
    e0, e1 = int(A.p[lo]), int(A.p[hi])
    npanels = 3
    # the exchange is set up from this rank's own block only
    xc = ColumnPanelExchange(dist, torch, part, rank, n, A.p[lo:hi + 1], A.i[e0:e1], npanels, torch.device("cpu"))
    my_i = torch.from_numpy(A.i[e0:e1][xc.perm].copy()); my_x = torch.from_numpy(A.x[e0:e1][xc.perm].copy())
    assert xc.cuts[0] == 0 and xc.cuts[-1] == n and sum(xc.panel_nnz) == A.nvals
    assert max(xc.panel_nnz) < 0.6 * A.nvals                          # a rank holds a panel, not the matrix
    room = max(xc.panel_nnz)
    Bi = torch.empty(room, dtype=torch.int64); Bx = torch.empty(room, dtype=torch.float64)
    p = (A.p[lo:hi + 1] - A.p[lo]).astype(np.int64)
    Ablk = Matrix(n, hi - lo, Type.FP64, A.sparsity, False, p=p, i=A.i[e0:e1], x=A.x[e0:e1])
    S = A.to_scipy()
    rows = [[] for _ in range(hi - lo)]
    for c in range(npanels):
        Bi.fill_(-1)
        xc.gather(c, my_i, my_x, Bi, Bx)
        nz = xc.panel_nnz[c]
        Sc = S[:, xc.cuts[c]:xc.cuts[c + 1]].tocsr(); Sc.sort_indices()       # the panel, from the whole matrix
        assert np.array_equal(xc.panel_p[c], Sc.indptr) and np.array_equal(Bi.numpy()[:nz], Sc.indices + xc.cuts[c])
        assert np.array_equal(Bx.numpy()[:nz], Sc.data)
        Bc = Matrix(n, n, Type.FP64, A.sparsity, False, p=xc.panel_p[c], i=Bi.numpy()[:nz].copy(), x=Bx.numpy()[:nz].copy())
        Cc = oracle.saxpy3(None, False, False, Bc, Ablk, "PLUS_TIMES_FP64", wait=True)       # C(I, J_c), CSR
        assert Cc.nvals == 0 or (Cc.i.min() >= xc.cuts[c] and Cc.i.max() < xc.cuts[c + 1])
        for r in range(hi - lo):
            rows[r].append((Cc.i[Cc.p[r]:Cc.p[r + 1]], Cc.x[Cc.p[r]:Cc.p[r + 1]]))
    ci = [np.concatenate([q[0] for q in r]) for r in rows]; cx = [np.concatenate([q[1] for q in r]) for r in rows]
    np.savez(os.path.join(outdir, f"c{rank}.npz"), p=np.concatenate([[0], np.cumsum([len(v) for v in ci])]),
             i=np.concatenate(ci), x=np.concatenate(cx), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_column_panels_of_B_give_the_same_product_on_two_ranks(tmp_path):
    """B exchanged one column panel at a time (for a B that does not fit whole on every rank): each rank sets the
    exchange up from its own row block, every gathered panel equals that column range of the whole matrix, and the
    panels of C(I, J_c) = A(I,:) B(:, J_c) laid side by side are the rows of the one-process product."""
    import torch.multiprocessing as mp
    from graphblas_amd import gen
    from oracle import oracle
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_panels, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    A = gen.rmat(9, 8, seed=42, values="int")
    want = oracle.saxpy3(None, False, False, A, A, "PLUS_TIMES_FP64", wait=True)
    blocks = [np.load(tmp_path / f"c{r}.npz") for r in range(world)]
    assert blocks[0]["lo"] == 0 and blocks[0]["hi"] == blocks[1]["lo"] and blocks[1]["hi"] == A.nrows
    p = np.concatenate([blocks[0]["p"], blocks[1]["p"][1:] + blocks[0]["p"][-1]])
    i = np.concatenate([b["i"] for b in blocks]); x = np.concatenate([b["x"] for b in blocks])
    assert np.array_equal(p, want.p) and np.array_equal(i, want.i) and np.array_equal(x, want.x)
