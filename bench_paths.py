#!/usr/bin/env python3
"""Secondary measurements of the multiply path on one MI355X (not the driver's bench line):

  spmv      RMAT-s, A CSR fp32, x full fp32:  y = A*x (dot2, y bitmap), y += A*x (dot4, in place),
            y' += x'*A (saxpy4, scatter) -- PLUS_TIMES_FP32 through gba_mxv / gba_vxm
  tricount  L = tril(A|A', -1), C<L,struct> = L*L' PLUS_PAIR_INT64 through gba_mxm (dot3), sum(C)
  bfs       level-synchronous BFS from the highest-degree vertex, one masked gba_vxm per level
  sssp      Bellman-Ford from the highest-degree vertex, one gba_vxm (MIN_PLUS_FP64, MIN accumulator) per iteration
  complex   C = A*A with PLUS_TIMES_FC64 through gba_mxm (the hipRTC road of complex and user-defined types)

Prints one JSON line per measurement: device time per call (HIP events on the library stream, from
the call's report), algorithmic bytes per call and the HBM rate they give.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timed(fn, iters, warmup=2):
    import graphblas_amd as g
    for _ in range(warmup):
        fn()
    g.sync()
    dev, t0 = [], time.perf_counter()
    for _ in range(iters):
        dev.append(fn().kernel_ms)
    g.sync()
    wall = (time.perf_counter() - t0) * 1e3 / iters
    return float(np.median(dev)), wall


def spmv(args):
    import graphblas_amd as g
    from graphblas_amd import gen, Matrix, Type
    fp = {"fp32": (Type.FP32, np.float32, "FP32", 4), "fp64": (Type.FP64, np.float64, "FP64", 8)}[args.type]
    A = gen.rmat(args.scale, 16, seed=42, type=fp[0])                # CSR
    n, nnz = A.nrows, A.nvals
    rng = np.random.default_rng(1)
    x = Matrix.full_vector((rng.random(n) + 0.5).astype(fp[1]))
    y0 = Matrix.full_vector(np.zeros(n, fp[1]))
    dA, dx, dy = A.to_device(), x.to_device(), y0.to_device()
    bytes_spmv = nnz * (8 + fp[3]) + (n + 1) * 8 + 2 * n * fp[3]    # int64 indices as handed over
    PLUS, SR = "PLUS_" + fp[2], "PLUS_TIMES_" + fp[2]
    base = dict(workload=f"RMAT-{args.scale} ef=16 seed=42 SpMV {SR}, CSR int64", n=n, nnz=nnz,
                algorithmic_bytes=bytes_spmv, unit="GB/s", peak=HBM_PEAK_GBS)

    def run(label, fn, extra_bytes=0):
        ms, wall = timed(fn, args.iters)
        rep = fn()
        gbs = (bytes_spmv + extra_bytes) / (ms * 1e-3) / 1e9
        emit(case=label, method=g.Method.NAMES[rep.method], ms=round(ms, 4), wall_ms=round(wall, 4),
             achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4), GFLOPs=round(2 * nnz / ms / 1e6, 1),
             w_i=int(rep.index_bytes) or 8, **base)

    run("y += A*x (accum PLUS, y full)", lambda: g.mxv(dy, None, PLUS, SR, dA, dx))

    def dot2_call():
        w = g.DeviceMatrix.upload(Matrix.sparse_vector(n, np.zeros(0, np.int64), np.zeros(0, fp[1])))
        rep = g.mxv(w, None, None, SR, dA, dx)
        w.free()
        return rep
    run("y = A*x (no accum, y bitmap)", dot2_call, extra_bytes=n)
    run("y' += x'*A (vxm, scatter)", lambda: g.vxm(dy, None, PLUS, SR, dx, dA))
    for d in (dA, dx, dy):
        d.free()


def tricount(args):
    import graphblas_amd as g
    from graphblas_amd import gen, Type
    A = gen.rmat(args.scale, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    L = gen.tril_strict(A)
    n, nnz = L.nrows, L.nvals
    dL = L.to_device()
    deg = np.diff(L.p)
    # per mask entry (i,j) the list A(:,i) is read; B(:,j) is read once per mask vector (the kernel keeps it in LDS
    # for all entries of the vector); plus the mask and the output.  `both_lists_bytes` is the per-entry model that
    # reads both lists for every entry.
    rows = np.repeat(np.arange(n), deg)
    pair_bytes = int((deg[rows] + deg[L.i]).sum()) * 8
    both_lists = nnz * 8 + (n + 1) * 8 + pair_bytes + nnz * 16
    alg = nnz * 8 + (n + 1) * 8 + int(deg[L.i].sum()) * 8 + nnz * 8 + nnz * 16

    def call():
        C = g.DeviceMatrix.empty(n, n, Type.INT64)
        rep = g.mxm(C, dL, None, "PLUS_PAIR_INT64", dL, dL, g.descriptor("ST1"))
        call.last = C
        return rep
    reps = []
    for _ in range(args.iters):
        rep = call()
        reps.append(rep.kernel_ms)
        C = call.last
        if _ < args.iters - 1:
            C.free()
    ms = float(np.median(reps))
    # outside the timed calls: the total, and every C(i,j) of sampled rows against scipy's integer product
    Ch = call.last.to_host()
    live = Ch.i >= 0
    tri = int(Ch.x[live].sum())
    S = L.to_scipy().astype(np.int64); S.data[:] = 1
    rows = np.arange(0, n, max(1, n // 256))
    R = S[rows]
    ref = (R @ S.T.tocsr()).multiply(R).tocsr(); ref.sort_indices()
    verified = bool(np.array_equal(Ch.p, L.p))
    for q, r in enumerate(rows):
        seg = slice(Ch.p[r], Ch.p[r + 1])
        keep = Ch.i[seg] >= 0
        verified &= bool(np.array_equal(Ch.i[seg][keep], ref.indices[ref.indptr[q]:ref.indptr[q + 1]]) and
                         np.array_equal(Ch.x[seg][keep], ref.data[ref.indptr[q]:ref.indptr[q + 1]]))
    call.last.free(); dL.free()
    gbs = alg / (ms * 1e-3) / 1e9
    emit(case="C<L,struct> = L*L' PLUS_PAIR_INT64", method=g.Method.NAMES[rep.method], n=n, nnz_L=nnz, triangles=tri,
         verified=verified, verified_rows=int(len(rows)),
         ms=round(ms, 3), algorithmic_bytes=alg, both_lists_bytes=both_lists, achieved=round(gbs, 1), unit="GB/s", peak=HBM_PEAK_GBS,
         frac=round(gbs / HBM_PEAK_GBS, 4), workload=f"RMAT-{args.scale} ef=16 seed=42 triangle count")
    if not verified:
        sys.exit("tricount: C differs from the scipy product on the sampled rows")


def bfs(args):
    """Level-synchronous BFS, the LAGraph push step: q<!pi,struct,replace> = q'*A with ANY_SECONDI_INT64,
    pi held bitmap, q sparse or bitmap (bitmap once it holds more than n/8 entries)."""
    import graphblas_amd as g
    from graphblas_amd import gen, Matrix, Type
    A = gen.rmat(args.scale, 16, seed=42, type=Type.INT64, values="ones", symmetric=True, drop_diagonal=True)
    n, nnz = A.nrows, A.nvals
    dA = A.to_device()
    src = int(np.argmax(np.diff(A.p)))
    if not getattr(args, "_warm", False):                   # the first traversal loads kernels and fills the memory pool
        args._warm = True
        saved = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            dA.free()
            bfs(args)
        finally:
            sys.stdout = saved
        dA = A.to_device()
    pi_b = np.zeros(n, np.int8); pi_x = np.zeros(n, np.int64)
    pi_b[src] = 1; pi_x[src] = src
    q = Matrix.sparse_vector(n, [src], np.array([src], np.int64))
    levels, total_ms, visited = [], 0.0, 1
    for level in range(64):
        dq = q.to_device()
        dpi = Matrix.bitmap_vector(pi_x, pi_b).to_device()
        dw = Matrix.sparse_vector(n, [], np.zeros(0, np.int64)).to_device()
        g.sync()
        t0 = time.perf_counter()
        rep = g.vxm(dw, dpi, None, "ANY_SECONDI_INT64", dq, dA, "RSC")
        g.sync()
        ms = (time.perf_counter() - t0) * 1e3
        w = dw.to_host()
        r, _, parents = w.coo()
        for d in (dq, dpi, dw):
            d.free()
        levels.append(dict(level=level, frontier=int(q.nvals), found=int(len(r)), method=g.Method.NAMES[rep.method],
                           ms=round(ms, 3), device_ms=round(rep.kernel_ms, 3), flops=int(rep.flops),
                           phases=[round(x, 2) for x in rep.phase_ms[:7]]))
        total_ms += ms
        if len(r) == 0:
            break
        visited += len(r)
        pi_b[r] = 1; pi_x[r] = parents
        if len(r) > n // 8:
            b = np.zeros(n, np.int8); x = np.zeros(n, np.int64); b[r] = 1; x[r] = parents
            q = Matrix.bitmap_vector(x, b)
        else:
            q = Matrix.sparse_vector(n, r, parents)
    dA.free()
    emit(case="BFS push steps, ANY_SECONDI_INT64, masked by !pi", workload=f"RMAT-{args.scale} ef=16 seed=42 symmetrised",
         n=n, nnz=nnz, source=src, visited=visited, total_ms=round(total_ms, 2),
         GTEPS=round(nnz / total_ms / 1e6, 3), levels=levels)


def sssp(args):
    """Bellman-Ford, every iteration d = min(d, d min.+ A) as one gba_vxm with MIN_PLUS_FP64 and the MIN accumulator
    (C full, in place: saxpy4), from the highest-degree vertex until nothing changes."""
    import graphblas_amd as g
    from graphblas_amd import gen, Matrix, Type
    A = gen.rmat(args.scale, 16, seed=42, type=Type.FP64)            # CSR, weights uniform in (0, 1]
    n, nnz = A.nrows, A.nvals
    src = int(np.argmax(np.diff(A.p)))
    dA = A.to_device()
    alg = nnz * 16 + (n + 1) * 8 + 2 * n * 8
    for trial in range(2):                                   # the second traversal is the warm one
        d = np.full(n, np.inf); d[src] = 0.0
        its = []
        for it in range(64):
            dcur = Matrix.full_vector(d).to_device()
            dnew = Matrix.full_vector(d.copy()).to_device()
            g.sync()
            t0 = time.perf_counter()
            rep = g.vxm(dnew, None, "MIN_FP64", "MIN_PLUS_FP64", dcur, dA)
            g.sync()
            ms = (time.perf_counter() - t0) * 1e3
            got = dnew.to_host().x
            dcur.free(); dnew.free()
            its.append(dict(iteration=it, method=g.Method.NAMES[rep.method], ms=round(ms, 3), device_ms=round(rep.kernel_ms, 3),
                            reached=int(np.isfinite(got).sum())))
            if np.array_equal(got, d):
                break
            d = got
    dA.free()
    total = sum(x["ms"] for x in its)
    emit(case="SSSP Bellman-Ford, MIN_PLUS_FP64 with MIN accumulator", workload=f"RMAT-{args.scale} ef=16 seed=42, weights uniform (0,1]",
         n=n, nnz=nnz, source=src, iterations=len(its), total_ms=round(total, 2), ms_per_iteration=round(total / len(its), 3),
         algorithmic_bytes_per_iteration=alg, achieved=round(alg / (total / len(its) * 1e-3) / 1e9, 1), unit="GB/s", peak=HBM_PEAK_GBS,
         reached=its[-1]["reached"], per_iteration=its)


def complex_spgemm(args):
    """C = A*A with PLUS_TIMES_FC64 through gba_mxm: the road of the user-defined types (pattern by the ANY_PAIR
    kernels, then one compiled dot product per entry of C).  A is the RMAT pattern with complex values."""
    import graphblas_amd as g
    from graphblas_amd import gen, Type, Matrix, DeviceMatrix
    A = gen.rmat(args.scale, 16, seed=42, type=Type.FP64)
    rng = np.random.default_rng(1)
    Ac = Matrix(A.vlen, A.vdim, Type.FC64, A.sparsity, A.is_csc, p=A.p, i=A.i, x=A.x + 1j * rng.random(len(A.x)))
    n, nnz = A.nrows, A.nvals
    dA = Ac.to_device()
    res = []
    for it in range(max(args.iters, 1)):
        dC = DeviceMatrix.empty(n, n, Type.FC64)
        g.sync(); t0 = time.perf_counter()
        rep = g.mxm(dC, None, None, "PLUS_TIMES_FC64", dA, dA, None)
        g.sync(); wall = (time.perf_counter() - t0) * 1e3
        cnz = dC.view.nvals
        dC.free()
        res.append({"wall_ms": round(wall, 2), "kernel_ms": round(rep.kernel_ms, 2)})
    dA.free()
    best = min(r["wall_ms"] for r in res[1:] or res)
    alg = (2 * nnz + cnz) * 24                        # int64 index + 16-byte value, A twice and C once
    emit(case="C = A*A PLUS_TIMES_FC64 (hipRTC road)", workload=f"RMAT-{args.scale} ef=16 seed=42, complex values",
         n=n, nnz_A=nnz, nnz_C=int(cnz), calls=res, best_wall_ms=best, algorithmic_bytes=alg,
         achieved=round(alg / (best * 1e-3) / 1e9, 1), unit="GB/s", peak=HBM_PEAK_GBS)


def launch_ranks(gpus):
    """`--gpus N` with N > 1 from a plain `python bench_paths.py`: start N ranks (one process per GPU, RCCL) and pass
    their exit code on.  Nothing has touched the GPU yet: the ranks are children, not a replacement of this process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["spmv", "tricount", "bfs", "sssp", "complex"])
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--type", choices=["fp32", "fp64"], default="fp32", help="spmv value type")
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); above 1, or with GBA_BENCH_FORCE_DIST=1, the "
                    "row-block programs of bench_paths_dist.py run instead of the one-GPU measurements")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if a.gpus > 1 or os.environ.get("GBA_BENCH_FORCE_DIST"):
        import bench_paths_dist
        if a.what not in bench_paths_dist.PROGRAMS:
            sys.exit(f"bench_paths.py: no sharded program for '{a.what}'")
        _, rec = bench_paths_dist.run(a.what, a)
        if rec is not None:
            emit(**rec)
        sys.exit(0)
    import graphblas_amd as g
    g.init(0)
    {"spmv": spmv, "tricount": tricount, "bfs": bfs, "sssp": sssp, "complex": complex_spgemm}[a.what](a)
