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
