#!/usr/bin/env python3
"""Secondary measurements of the multiply path on one MI355X (not the driver's bench line):

  spmv      RMAT-s, A CSR fp32, x full fp32:  y = A*x (dot2, y bitmap), y += A*x (dot4, in place),
            y' += x'*A (saxpy4, scatter) -- PLUS_TIMES_FP32 through gba_mxv / gba_vxm
  tricount  L = tril(A|A', -1), C<L,struct> = L*L' PLUS_PAIR_INT64 through gba_mxm (dot3), sum(C)

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
    A = gen.rmat(args.scale, 16, seed=42, type=Type.FP32)            # CSR
    n, nnz = A.nrows, A.nvals
    rng = np.random.default_rng(1)
    x = Matrix.full_vector((rng.random(n) + 0.5).astype(np.float32))
    y0 = Matrix.full_vector(np.zeros(n, np.float32))
    dA, dx, dy = A.to_device(), x.to_device(), y0.to_device()
    bytes_spmv = nnz * (8 + 4) + (n + 1) * 8 + n * 4 + n * 4        # int64 indices as handed over
    base = dict(workload=f"RMAT-{args.scale} ef=16 seed=42 SpMV PLUS_TIMES_FP32, CSR int64", n=n, nnz=nnz,
                algorithmic_bytes=bytes_spmv, unit="GB/s", peak=HBM_PEAK_GBS)

    def run(label, fn, extra_bytes=0):
        ms, wall = timed(fn, args.iters)
        rep = fn()
        gbs = (bytes_spmv + extra_bytes) / (ms * 1e-3) / 1e9
        emit(case=label, method=g.Method.NAMES[rep.method], ms=round(ms, 4), wall_ms=round(wall, 4),
             achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4), GFLOPs=round(2 * nnz / ms / 1e6, 1), **base)

    run("y += A*x (accum PLUS, y full)", lambda: g.mxv(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, dx))

    def dot2_call():
        w = g.DeviceMatrix.upload(Matrix.sparse_vector(n, np.zeros(0, np.int64), np.zeros(0, np.float32)))
        rep = g.mxv(w, None, None, "PLUS_TIMES_FP32", dA, dx)
        w.free()
        return rep
    run("y = A*x (no accum, y bitmap)", dot2_call, extra_bytes=n)
    run("y' += x'*A (vxm, scatter)", lambda: g.vxm(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dx, dA))
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
    # per mask entry (i,j): both lists are read
    rows = np.repeat(np.arange(n), deg)
    pair_bytes = int((deg[rows] + deg[L.i]).sum()) * 8
    alg = nnz * 8 + (n + 1) * 8 + pair_bytes + nnz * 16

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
    Ch = call.last.to_host() if args.scale <= 20 else None
    tri = int(Ch.x[Ch.i >= 0].sum()) if Ch is not None else None
    call.last.free(); dL.free()
    gbs = alg / (ms * 1e-3) / 1e9
    emit(case="C<L,struct> = L*L' PLUS_PAIR_INT64", method=g.Method.NAMES[rep.method], n=n, nnz_L=nnz, triangles=tri,
         ms=round(ms, 3), algorithmic_bytes=alg, achieved=round(gbs, 1), unit="GB/s", peak=HBM_PEAK_GBS,
         frac=round(gbs / HBM_PEAK_GBS, 4), workload=f"RMAT-{args.scale} ef=16 seed=42 triangle count")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["spmv", "tricount"])
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import graphblas_amd as g
    g.init(0)
    {"spmv": spmv, "tricount": tricount}[a.what](a)
