"""Worker of tests/test_zy_rccl_single.py: runs in a process of its own (torch is imported before the library is
loaded, as in bench.py, so that both use the HIP runtime that torch ships), one rank over RCCL on GPU 0."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402  (before graphblas_amd loads its library)
import torch.distributed as dist                # noqa: E402


def collectives(dev):
    from graphblas_amd.dist import _all_gather_padded, FrontierExchange, ScalarReduce
    assert dist.get_backend() == "nccl"
    for dtype in (torch.int64, torch.float64, torch.float32, torch.int8):
        send = torch.arange(1000, device=dev).to(dtype)
        flat = _all_gather_padded(dist, torch, send, 1000, 1)
        assert flat.is_cuda and torch.equal(flat, send)
    fx = FrontierExchange(dist, torch, 0, 1, dev)
    d = torch.tensor([3.0, float("inf"), 1.5], dtype=torch.float64, device=dev)
    fx._all_reduce(d, dist.ReduceOp.MIN)
    assert d.cpu().tolist() == [3.0, float("inf"), 1.5]
    b = torch.tensor([0, 1, 0, 1], dtype=torch.int8, device=dev)
    fx._all_reduce(b, dist.ReduceOp.MAX)
    assert b.cpu().tolist() == [0, 1, 0, 1]
    t = torch.tensor([123456789012], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    assert int(t.item()) == 123456789012
    assert ScalarReduce(dist, torch, 1, dev).sum_int64(7) == 7
    dist.barrier()


def gather_feeds_the_library(dev):
    """What a rank of the N > 1 bench does in a step, with RCCL in the loop: its row block of A goes through the
    padded all-gather into the torch tensors that the library then reads as the right-hand operand."""
    import graphblas_amd as g
    from graphblas_amd import gen, Type, DeviceMatrix
    from graphblas_amd.binding import gba_matrix
    from graphblas_amd.dist import _all_gather_padded
    g.init(0)                                   # after RCCL is up, the order of bench.py
    A = gen.rmat(10, 8, seed=3, type=Type.FP64)
    n, nnz = A.nrows, A.nvals
    Bp = torch.from_numpy(A.p).to(dev)
    si, sx = torch.from_numpy(A.i).to(dev), torch.from_numpy(A.x).to(dev)
    Bi = _all_gather_padded(dist, torch, si, nnz, 1)
    Bx = _all_gather_padded(dist, torch, sx, nnz, 1)
    torch.cuda.synchronize()
    m = gba_matrix()
    m.vlen, m.vdim, m.nvec, m.nvals = n, n, n, nnz
    m.p, m.i, m.x = Bp.data_ptr(), Bi.data_ptr(), Bx.data_ptr()
    m.type, m.sparsity, m.mem, m.is_csc = Type.FP64, g.Sparsity.SPARSE, 0, False
    dB = DeviceMatrix(m, owns=False)
    dC = DeviceMatrix.empty(n, n, Type.FP64)
    g.mxm(dC, None, None, "PLUS_TIMES_FP64", dB, dB, None)
    S = A.to_scipy()
    want = (S @ S).tocsr()
    got = dC.to_host().to_scipy()
    assert got.nnz == want.nnz
    assert np.allclose(got.toarray(), want.toarray(), rtol=1e-13, atol=0)
    dC.free()
    g.sync()


def main():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        collectives(dev)
        print("collectives ok", flush=True)
        gather_feeds_the_library(dev)
        print("gather ok", flush=True)
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
