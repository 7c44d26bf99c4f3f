"""The programs of bench_paths.py on N GPUs of one node: one process per GPU, every product sharded by row blocks (of
its output; of A and the frontier for the traversal), the exchanges of graphblas_amd.dist between the products.

  spmv      x <- A x / max|A x|   rank r computes y(I_r) = A(I_r,:) x (gba_mxv, dot4), the blocks are all-gathered into
                                  the next x (VectorBlockExchange); every rank then holds all of y and scales it itself
  bfs       w_r<!pi,struct,replace> = q(I_r)' A(I_r,:) with ANY_SECONDI_INT64 (gba_vxm: the push step of the one-GPU
            program on the rows of A this rank holds and its share of the frontier), the lists of vertices found
            exchanged (FrontierExchange.sparse) and merged: a vertex reached from several blocks keeps one parent
  sssp      d(I_r) = min(d(I_r), A'(I_r,:) min.+ d) with MIN_PLUS_FP64 and the MIN accumulator (gba_mxv, dot4 in
            place), the blocks all-gathered (VectorBlockExchange)
  tricount  C(I_r,:)<L(I_r,:),struct> = L(I_r,:) L' with PLUS_PAIR_INT64 (gba_mxm, dot3) against the whole L, only the
            per-rank sums travel (ScalarReduce)

`run(what, args)` starts nothing: the caller (bench_paths.py) has started the ranks.  With one rank and
GBA_BENCH_FORCE_DIST=1 the same code runs with its collectives over RCCL, which is how a one-GPU box exercises it;
such a run is not a scaling measurement.  torch is imported before the library so that both use the HIP runtime that
torch ships.
"""
import os
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


class Ranks:
    """Process group and device of this rank."""

    def __init__(self, gpus):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != gpus:
            sys.exit(f"bench_paths.py: --gpus {gpus} but the launcher started {self.world} rank(s) (WORLD_SIZE)")
        self.multi = self.world > 1 or bool(os.environ.get("GBA_BENCH_FORCE_DIST"))
        self.dist = None
        if self.multi:
            import torch.distributed as dist
            self.dist = dist
            if not dist.is_initialized():
                if "MASTER_ADDR" not in os.environ:
                    import socket
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
                sys.stdout.flush()
                saved = os.dup(1)                      # RCCL's banner goes to the standard error, the JSON lines stay alone
                os.dup2(2, 1)
                if os.environ.get("GBA_BENCH_REHEARSE"):
                    self.local_rank = 0                # every rank on device 0 over gloo: a rehearsal, not a measurement
                    torch.cuda.set_device(0)
                    dist.init_process_group("gloo")
                else:
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                dist.barrier()
                torch.cuda.synchronize()
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        self.dev = torch.device("cuda", self.local_rank)
        import graphblas_amd as g
        g.init(self.local_rank)
        self.g = g

    def fence(self):
        self.g.sync()
        self.torch.cuda.synchronize()
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.multi:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def _row_block(M, lo, hi):
    """Rows [lo, hi) of a CSR Matrix as a Matrix of its own (row pointers rebased)."""
    from graphblas_amd import Matrix
    e0, e1 = int(M.p[lo]), int(M.p[hi])
    return Matrix(M.vlen, hi - lo, M.type, M.sparsity, M.is_csc, p=(M.p[lo:hi + 1] - M.p[lo]).astype(np.int64),
                  i=M.i[e0:e1], x=M.x[e0:e1])


def _full_vector_over(t, type_code):
    """The library's view of a torch tensor as a full vector (no copy; the tensor stays the owner)."""
    from graphblas_amd import DeviceMatrix, Sparsity
    from graphblas_amd.binding import gba_matrix
    m = gba_matrix()
    m.vlen, m.vdim, m.nvec, m.nvals = t.numel(), 1, 1, t.numel()
    m.x = t.data_ptr()
    m.type, m.sparsity, m.mem, m.is_csc = type_code, Sparsity.FULL, 0, True
    return DeviceMatrix(m, owns=False)


class _DeviceArray:
    """A device array of the library for torch.as_tensor (no copy; the library stays the owner)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _int64_over(torch, ptr, count, dev):
    return torch.as_tensor(_DeviceArray(ptr, count, "<i8"), device=dev)


def _split(prefix, n, world):
    from graphblas_amd.dist import split_by_flops
    return split_by_flops(np.asarray(prefix, np.int64), 0, n, world)


def spmv(R, args):
    """x <- A x / max|A x|, `iters` times.  Returns the final x (host) and the record."""
    g, torch = R.g, R.torch
    from graphblas_amd import gen, Type
    from graphblas_amd.dist import VectorBlockExchange
    A = gen.rmat(args.scale, 16, seed=42, type=Type.FP32)
    n, nnz = A.nrows, A.nvals
    part = _split(A.p, n, R.world)                                   # SpMV work of a row = its entries
    lo, hi = part[R.rank], part[R.rank + 1]
    dA = _row_block(A, lo, hi).to_device()
    x = torch.from_numpy(np.linspace(0.5, 1.5, n).astype(np.float32)).to(R.dev)
    y = torch.zeros(hi - lo, dtype=torch.float32, device=R.dev)
    nxt = torch.empty(n, dtype=torch.float32, device=R.dev)
    xv, yv = _full_vector_over(x, Type.FP32), _full_vector_over(y, Type.FP32)
    xchg = VectorBlockExchange(R.dist, torch, part, R.rank, R.dev, torch.float32, collective=R.multi)
    top = torch.zeros(1, dtype=torch.float32, device=R.dev)

    def iterate():
        y.zero_()
        torch.cuda.synchronize()
        g.mxv(yv, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, xv)      # y(I) += A(I,:) x, in place (dot4)
        g.sync()
        xchg.gather(y, nxt)
        torch.amax(nxt.abs(), dim=0, keepdim=True, out=top)          # every rank holds the whole vector: no reduce needed
        torch.div(nxt, top, out=x)
        torch.cuda.synchronize()

    for _ in range(2):
        iterate()
    x.copy_(torch.from_numpy(np.linspace(0.5, 1.5, n).astype(np.float32)))
    R.fence()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        iterate()
    R.fence()
    sec = R.max_over_ranks(time.perf_counter() - t0)
    ms = sec * 1e3 / args.iters
    alg = nnz * 12 + (n + 1) * 8 + 2 * n * 4
    out = x.cpu().numpy()
    dA.free()
    rec = dict(case="x <- A x / max|A x|, rows of y sharded, x all-gathered", n_gpus=R.world, n=n, nnz=nnz, iterations=args.iters,
               ms_per_iteration=round(ms, 4), algorithmic_bytes=alg, achieved=round(alg / (ms * 1e-3) / 1e9, 1), unit="GB/s",
               rows_per_rank=[part[r + 1] - part[r] for r in range(R.world)], collective=bool(R.multi),
               workload=f"RMAT-{args.scale} ef=16 seed=42 SpMV PLUS_TIMES_FP32, CSR int64")
    return out, rec


def bfs(R, args):
    """Level-synchronous BFS from the highest-degree vertex.  Returns (levels, parents) on the host and the record:
    levels[v] = distance of v from the source (-1: not reached), parents[v] = the parent the ANY monoid picked."""
    g, torch = R.g, R.torch
    from graphblas_amd import gen, Matrix, Type
    from graphblas_amd.dist import FrontierExchange
    A = gen.rmat(args.scale, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    n, nnz = A.nrows, A.nvals
    part = _split(A.p, n, R.world)
    lo, hi = part[R.rank], part[R.rank + 1]
    dA = _row_block(A, lo, hi).to_device()
    src = int(np.argmax(np.diff(A.p)))
    fx = FrontierExchange(R.dist, torch, R.rank, R.world, R.dev, collective=R.multi)

    def traverse():
        level_of = np.full(n, -1, np.int64); parent = np.full(n, -1, np.int64)
        level_of[src] = 0; parent[src] = src
        pi_b = np.zeros(n, np.int8); pi_b[src] = 1
        q_idx = np.array([src], np.int64)                            # the whole frontier, ascending, on every rank
        per_level = []
        spent = {"product": 0.0, "exchange": 0.0}                    # inside gba_vxm / inside the exchange, this rank
        R.fence()
        t0 = time.perf_counter()
        for level in range(1, 64):
            # this rank's share of the frontier: the vertices whose rows of A it holds
            a, b = np.searchsorted(q_idx, [lo, hi])
            mine = q_idx[a:b] - lo
            method = "none"
            r = np.zeros(0, np.int64); par = np.zeros(0, np.int64)
            if len(mine):
                if len(mine) > (hi - lo) // 8:
                    bq = np.zeros(hi - lo, np.int8); bq[mine] = 1
                    q = Matrix.bitmap_vector(np.zeros(hi - lo, np.int64), bq)
                else:
                    q = Matrix.sparse_vector(hi - lo, mine, np.zeros(len(mine), np.int64))
                dq = q.to_device()
                dpi = Matrix.bitmap_vector(np.zeros(n, np.int64), pi_b).to_device()
                dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
                g.sync()
                tc = time.perf_counter()
                rep = g.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")     # w<!pi> = q(I)' A(I,:): SECONDI = row of the block
                g.sync()
                spent["product"] += time.perf_counter() - tc
                w = dw.to_host()
                for d in (dq, dpi, dw):
                    d.free()
                r, _, par = w.coo()                                  # whatever form the method left w in
                par = par + lo
                method = g.Method.NAMES[rep.method]
            ti, tp = torch.from_numpy(np.ascontiguousarray(r)).to(R.dev), torch.from_numpy(np.ascontiguousarray(par)).to(R.dev)
            torch.cuda.synchronize()
            tx = time.perf_counter()
            gi, gp = fx.sparse(ti, tp)
            torch.cuda.synchronize()
            spent["exchange"] += time.perf_counter() - tx
            found, first = np.unique(gi.cpu().numpy(), return_index=True)   # a vertex reached from several blocks keeps one parent
            parents = gp.cpu().numpy()[first]
            per_level.append(dict(level=level, frontier=int(len(q_idx)), mine=int(len(mine)), found=int(len(found)), method=method))
            if len(found) == 0:
                break
            level_of[found] = level; parent[found] = parents; pi_b[found] = 1
            q_idx = found
        R.fence()
        total = R.max_over_ranks(time.perf_counter() - t0)
        return level_of, parent, per_level, (total, R.max_over_ranks(spent["product"]), R.max_over_ranks(spent["exchange"]))

    traverse()                                                      # loads the kernels, fills the memory pool
    level_of, parent, per_level, (sec, product_s, exchange_s) = traverse()
    dA.free()
    rec = dict(case="BFS push steps w<!pi> = q(I)' A(I,:) per row block, ANY_SECONDI_INT64, found lists exchanged and merged", n_gpus=R.world,
               n=n, nnz=nnz, source=src, visited=int((level_of >= 0).sum()), total_ms=round(sec * 1e3, 2),
               # slowest rank: time inside gba_vxm, inside the exchange; the rest of total_ms is the host's bookkeeping
               # between levels (frontier and mask vectors built and uploaded, found lists merged)
               product_ms=round(product_s * 1e3, 2), exchange_ms=round(exchange_s * 1e3, 2),
               GTEPS=round(nnz / max(product_s + exchange_s, 1e-9) / 1e9, 3), levels=per_level, collective=bool(R.multi),
               workload=f"RMAT-{args.scale} ef=16 seed=42 symmetrised")
    return (level_of, parent), rec


def sssp(R, args):
    """Bellman-Ford from the highest-degree vertex until nothing changes.  Returns the distances and the record."""
    g, torch = R.g, R.torch
    from graphblas_amd import gen, Matrix, Type
    from graphblas_amd.dist import VectorBlockExchange
    A = gen.rmat(args.scale, 16, seed=42, type=Type.FP64)            # CSR, weights uniform in (0, 1]
    n, nnz = A.nrows, A.nvals
    src = int(np.argmax(np.diff(A.p)))
    T = Matrix.from_scipy(A.to_scipy().T.tocsr(), type=Type.FP64)    # d_new(j) = min_k d(k) + A(k,j): rows of A'
    part = _split(T.p, n, R.world)
    lo, hi = part[R.rank], part[R.rank + 1]
    dT = _row_block(T, lo, hi).to_device()
    d = torch.empty(n, dtype=torch.float64, device=R.dev)
    mine = torch.empty(hi - lo, dtype=torch.float64, device=R.dev)
    nxt = torch.empty(n, dtype=torch.float64, device=R.dev)
    dv, mv = _full_vector_over(d, Type.FP64), _full_vector_over(mine, Type.FP64)
    xchg = VectorBlockExchange(R.dist, torch, part, R.rank, R.dev, torch.float64, collective=R.multi)

    def solve():
        d.fill_(float("inf")); d[src] = 0.0
        its = 0
        R.fence()
        t0 = time.perf_counter()
        for _ in range(64):
            mine.copy_(d[lo:hi])
            torch.cuda.synchronize()
            g.mxv(mv, None, "MIN_FP64", "MIN_PLUS_FP64", dT, dv)     # in place, MIN accumulator (dot4)
            g.sync()
            xchg.gather(mine, nxt)
            its += 1
            same = bool(torch.equal(nxt, d))                         # every rank holds the whole vector
            d.copy_(nxt)
            if same:
                break
        R.fence()
        return its, R.max_over_ranks(time.perf_counter() - t0)

    solve()
    its, sec = solve()
    out = d.cpu().numpy()
    dT.free()
    alg = nnz * 16 + (n + 1) * 8 + 2 * n * 8
    rec = dict(case="SSSP Bellman-Ford, d(I) = min(d(I), A'(I,:) min.+ d), d all-gathered", n_gpus=R.world, n=n, nnz=nnz, source=src,
               iterations=its, total_ms=round(sec * 1e3, 2), ms_per_iteration=round(sec * 1e3 / its, 3),
               algorithmic_bytes_per_iteration=alg, achieved=round(alg / (sec / its) / 1e9, 1), unit="GB/s",
               reached=int(np.isfinite(out).sum()), collective=bool(R.multi),
               workload=f"RMAT-{args.scale} ef=16 seed=42, weights uniform (0,1]")
    return out, rec


def tricount(R, args):
    """Triangle count: the rows of the mask are sharded, L is whole on every rank.  Returns the count and the record."""
    g, torch = R.g, R.torch
    from graphblas_amd import gen, Type, DeviceMatrix
    from graphblas_amd.dist import ScalarReduce
    A = gen.rmat(args.scale, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    L = gen.tril_strict(A)
    n, nnz = L.nrows, L.nvals
    deg = np.diff(L.p)
    rows = np.repeat(np.arange(n), deg)
    work = np.bincount(rows, weights=(deg[rows] + deg[L.i]).astype(np.float64), minlength=n)   # both lists of every mask entry
    part = _split(np.concatenate([[0], np.cumsum(work)]), n, R.world)
    lo, hi = part[R.rank], part[R.rank + 1]
    dL = L.to_device()
    dLblk = _row_block(L, lo, hi).to_device()
    red = ScalarReduce(R.dist, torch, R.world, R.dev, collective=R.multi)

    def count():
        R.fence()
        t0 = time.perf_counter()
        C = DeviceMatrix.empty(hi - lo, n, Type.INT64)
        rep = g.mxm(C, dLblk, None, "PLUS_PAIR_INT64", dLblk, dL, g.descriptor("ST1"))
        g.sync()
        nz = int(C.view.nvals)                                       # sum of the live entries, on the device
        Ci, Cx = _int64_over(torch, C.view.i, nz, R.dev), _int64_over(torch, C.view.x, nz, R.dev)
        mine = int(torch.where(Ci >= 0, Cx, torch.zeros_like(Cx)).sum().item()) if nz else 0
        C.free()
        total = red.sum_int64(mine)
        R.fence()
        return total, rep, R.max_over_ranks(time.perf_counter() - t0)

    count()
    total, rep, sec = count()
    dL.free(); dLblk.free()
    rec = dict(case="C(I,:)<L(I,:),struct> = L(I,:)*L' PLUS_PAIR_INT64, per-rank sums reduced", n_gpus=R.world, n=n, nnz_L=nnz,
               triangles=total, method=g.Method.NAMES[rep.method], total_ms=round(sec * 1e3, 2), device_ms=round(rep.kernel_ms, 3),
               rows_per_rank=[part[r + 1] - part[r] for r in range(R.world)], collective=bool(R.multi),
               workload=f"RMAT-{args.scale} ef=16 seed=42 triangle count")
    return total, rec


PROGRAMS = {"spmv": spmv, "bfs": bfs, "sssp": sssp, "tricount": tricount}


def run(what, args, ranks=None):
    """One program on the ranks that are up; rank 0 returns (result, record), the others (result, None)."""
    R = ranks if ranks is not None else Ranks(args.gpus)
    out, rec = PROGRAMS[what](R, args)
    return out, (rec if R.rank == 0 else None)
