#!/usr/bin/env python3
"""Headline benchmark: SpGEMM C = A*A, PLUS_TIMES_FP64, Graph500 RMAT (edge factor 16, seed 42),
through gba_mxm on 1..8 MI355X.

One step = one whole C = A*A.  C of RMAT-22 (~7e10 entries, ~1.2 TB with 64-bit indices) does not
fit any GPU, so a rank computes its rows in flop-balanced row chunks, C(I,:) = A(I,:)*A, and
releases each chunk's C before the next (every entry of C is produced, exactly once per step).
With N > 1 ranks the rows of A and C are 1-D partitioned by flop count; each step starts with the
all-gather of the row blocks of A (RCCL) that every rank needs as the right-hand operand.

Prints ONE JSON line (rank 0).  value = 2 * multiply-adds / time, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=int(os.environ.get("GBA_BENCH_SCALE", "22")))
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--chunk-flops", type=float, default=6e9, help="max multiply-adds per row chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the row-sum check of every chunk in the first warm-up step")
    ap.add_argument("--no-spmv", action="store_true", help="leave out the SpMV leg reported beside the headline")
    ap.add_argument("--cpu-flops", type=float, default=8e9, help="multiply-adds in the CPU baseline sample")
    ap.add_argument("--b-panels", type=int, default=0,
                    help="N > 1: exchange the right-hand operand one column panel at a time (this many panels) "
                         "instead of gathering it whole -- for a B that does not fit on every rank")
    return ap.parse_args()


def launch_ranks(args):
    """`--gpus N` with N > 1 from a plain `python bench.py`: start N ranks (one process per GPU, RCCL) and pass
    their exit code on.  Nothing has touched the GPU yet: the ranks are children, not a replacement of this
    process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("GBA_BENCH_WATCHDOG"):                # a hang prints every thread's stack and ends the process
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GBA_BENCH_WATCHDOG"]), exit=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    import torch
    dist = None
    # GBA_BENCH_FORCE_DIST=1: take the N > 1 code path (process group over RCCL, device-side partition, padded
    # all-gather, reductions of the results) with however many ranks there are, even one -- the way to run that
    # path on a one-GPU box.  The result of such a run with one rank is not a scaling measurement.
    multi = world > 1 or bool(os.environ.get("GBA_BENCH_FORCE_DIST"))
    if multi:
        if "MASTER_ADDR" not in os.environ:            # a plain `python bench.py` with the switch set: one rank
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        import torch.distributed as dist
        # RCCL prints a version banner on the standard output when its communicator comes up; the one line this
        # program owes its caller is the JSON record, so the banner is sent to the standard error instead
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if os.environ.get("GBA_BENCH_REHEARSE"):
            # rehearsal of the N > 1 code path on a box with one GPU: every rank on device 0, gloo
            # instead of RCCL (RCCL refuses two ranks on one device).  Not a measurement.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()                                  # the communicator exists from here on
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    import graphblas_amd as g
    from graphblas_amd import gen, Type, DeviceMatrix, Matrix
    from graphblas_amd.dist import row_flops, split_by_flops, chunks_by_flops, RowBlockExchange, ColumnPanelExchange
    from graphblas_amd.binding import gba_matrix
    g.init(local_rank)

    t0 = time.time()
    A = gen.rmat(args.scale, args.edge_factor, seed=42, type=Type.FP64)      # CSR, values uniform (0,1]
    n, nnzA = A.nrows, A.nvals
    rf = row_flops(A.p, A.i)
    prefix = np.concatenate([[0], np.cumsum(rf)]).astype(np.int64)
    total_madds = int(prefix[-1])
    gen_s = time.time() - t0

    # device copies.  Right-hand operand: full A on every rank, held in torch tensors so that RCCL can fill them.
    dev = torch.device("cuda", local_rank)
    panels = args.b_panels if multi else 0               # > 0: B travels one column panel at a time and is never whole
    Bp = torch.from_numpy(A.p).to(dev)
    Bi = torch.empty(1 if panels else nnzA, dtype=torch.int64, device=dev)
    Bx = torch.empty(1 if panels else nnzA, dtype=torch.float64, device=dev)

    # 1-D partition of rows over ranks, balanced by flops: the library's own split (the per-vector count saxpy3
    # bins by, gba_partition_by_flops), taken once on the pattern before the timed region
    if multi and not panels:
        Bi.copy_(torch.from_numpy(A.i))
        m0 = gba_matrix()
        m0.vlen, m0.vdim, m0.nvec, m0.nvals = n, n, n, nnzA
        m0.p, m0.i, m0.x = Bp.data_ptr(), Bi.data_ptr(), Bx.data_ptr()
        m0.type, m0.sparsity, m0.mem, m0.is_csc = Type.FP64, g.Sparsity.SPARSE, 0, False
        whole = DeviceMatrix(m0, owns=False)
        part = g.partition_by_flops(world, whole, whole)
        Bi.fill_(-1)                                      # the timed steps gather it again from the row blocks
    else:
        part = split_by_flops(prefix, 0, n, world)
    my_lo, my_hi = part[rank], part[rank + 1]
    my_chunks = chunks_by_flops(prefix, my_lo, my_hi, args.chunk_flops)
    e_lo, e_hi = int(A.p[my_lo]), int(A.p[my_hi])
    my_i = torch.from_numpy(A.i[e_lo:e_hi]).to(dev)
    my_x = torch.from_numpy(A.x[e_lo:e_hi]).to(dev)
    xchg = RowBlockExchange(dist, torch, A.p, part, rank, dev, torch.int64, torch.float64, collective=multi)

    def exchange():
        """all-gather of the row blocks of A into the full right-hand operand (a copy on one GPU)."""
        xchg.gather(my_i, my_x, Bi, Bx)

    def view_rows(lo, hi):
        m = gba_matrix()
        m.vlen, m.vdim, m.nvec = n, hi - lo, hi - lo
        m.nvals = int(A.p[hi] - A.p[lo])
        m.p, m.i, m.x = Bp.data_ptr() + 8 * lo, Bi.data_ptr(), Bx.data_ptr()
        m.type, m.sparsity, m.mem, m.is_csc = Type.FP64, g.Sparsity.SPARSE, 0, False
        return DeviceMatrix(m, owns=False)

    Bfull = view_rows(0, n)
    if panels:
        # the rank keeps its own row block twice: whole, as the left operand A(I,:), and cut into column panels, as its
        # share of the right operand; one gathered panel B(:, J_c) is resident at a time
        xc = ColumnPanelExchange(dist, torch, part, rank, n, A.p[my_lo:my_hi + 1], A.i[e_lo:e_hi], panels, dev, collective=True)
        my_i_perm = torch.from_numpy(A.i[e_lo:e_hi][xc.perm]).to(dev)
        my_x_perm = torch.from_numpy(A.x[e_lo:e_hi][xc.perm]).to(dev)
        room = max(max(xc.panel_nnz), 1)
        Pi = torch.empty(room, dtype=torch.int64, device=dev)
        Px = torch.empty(room, dtype=torch.float64, device=dev)
        panel_p = [torch.from_numpy(pp).to(dev) for pp in xc.panel_p]
        my_p = torch.from_numpy((A.p[my_lo:my_hi + 1] - e_lo).astype(np.int64)).to(dev)
        panel_chunks = []
        for c in range(panels):
            pc = np.concatenate([[0], np.cumsum(row_flops(A.p, A.i, xc.panel_p[c]))]).astype(np.int64)
            panel_chunks.append(chunks_by_flops(pc, my_lo, my_hi, args.chunk_flops))

        def own_rows(lo, hi):
            m = gba_matrix()
            m.vlen, m.vdim, m.nvec = n, hi - lo, hi - lo
            m.nvals = int(A.p[hi] - A.p[lo])
            m.p, m.i, m.x = my_p.data_ptr() + 8 * (lo - my_lo), my_i.data_ptr(), my_x.data_ptr()
            m.type, m.sparsity, m.mem, m.is_csc = Type.FP64, g.Sparsity.SPARSE, 0, False
            return DeviceMatrix(m, owns=False)

        def panel_view(c):
            m = gba_matrix()
            m.vlen, m.vdim, m.nvec, m.nvals = n, n, n, xc.panel_nnz[c]
            m.p, m.i, m.x = panel_p[c].data_ptr(), Pi.data_ptr(), Px.data_ptr()
            m.type, m.sparsity, m.mem, m.is_csc = Type.FP64, g.Sparsity.SPARSE, 0, False
            return DeviceMatrix(m, owns=False)
    sr = g.semiring("PLUS_TIMES_FP64")
    stats = {"phase_ms": np.zeros(8), "cnz": 0, "bins": np.zeros(4, np.int64), "bin_cnz": np.zeros(4, np.int64),
             "bin_bnz": np.zeros(4, np.int64), "bin_madds": np.zeros(4, np.int64), "launches": 0}

    # Correctness of what is timed, checked on every row chunk of one untimed step: the row sums of the chunk of C
    # against A_I (A v) computed on the host in fp64, for v = 1 and for a random positive v.  A missing, doubled or
    # misplaced entry, or a wrong value, shows in one of the two; all terms are positive, so the two evaluations of a
    # row sum of m products agree within 2 (m + 1) u relative (tests/util.py::assert_sums_within_bound).
    verify = {"chunks": 0, "rows": 0, "worst_ratio": 0.0, "ok": True, "nnz_ok": True}
    vecs = [np.ones(n), np.random.default_rng(7).random(n) + 0.5]
    checks = None

    def verify_chunk(Cc, lo, hi, cnz, cols=None):
        """cols = (j0, j1): the chunk holds the columns j0:j1 of its rows only (a column panel of B)."""
        nonlocal checks
        if checks is None:
            S = A.to_scipy()
            checks = [(Matrix.full_vector(v).to_device(), S, {None: S @ v}, v) for v in vecs]
        u = np.finfo(np.float64).eps / 2
        if Cc.view.nvals != cnz or (cnz <= 0 and cols is None):
            verify["nnz_ok"] = False
        for dv, S, Avs, v in checks:
            if cols not in Avs:
                Avs.clear()                                  # one panel's product at a time
                Avs[cols] = S[:, cols[0]:cols[1]] @ v[cols[0]:cols[1]]
            Av = Avs[cols]
            y = Matrix.full_vector(np.zeros(hi - lo)).to_device()
            g.mxv(y, None, "PLUS_FP64", sr, Cc, dv)                      # y = C_I v on the device (dot4)
            got = y.to_host().x
            y.free()
            want = S[lo:hi] @ Av
            m = rf[lo:hi].astype(np.float64)
            live = want > 0
            ratio = np.zeros(hi - lo)
            ratio[live] = np.abs(got[live] - want[live]) / (2 * (m[live] + 1) * u * want[live])
            if np.any(got[~live] != 0) or (live.any() and ratio.max() > 1.0):
                verify["ok"] = False
            verify["worst_ratio"] = max(verify["worst_ratio"], float(ratio.max()) if live.any() else 0.0)
        verify["chunks"] += 1
        verify["rows"] += hi - lo

    def step(collect, check=False):
        if not panels:
            exchange()
            torch.cuda.synchronize()
            for lo, hi in my_chunks:
                one_product(collect, check, None, lo, hi)
        else:
            for c in range(panels):                          # every rank walks the panels in the same order
                xc.gather(c, my_i_perm, my_x_perm, Pi, Px)
                torch.cuda.synchronize()
                for lo, hi in panel_chunks[c]:
                    one_product(collect, check, c, lo, hi)
        g.sync()

    def one_product(collect, check, c, lo, hi):
        Cc = DeviceMatrix.empty(hi - lo, n, Type.FP64)
        tw = time.perf_counter()
        if c is None:
            rep = g.mxm(Cc, None, None, sr, view_rows(lo, hi), Bfull, None)
        else:
            rep = g.mxm(Cc, None, None, sr, own_rows(lo, hi), panel_view(c), None)
        if check:
            verify_chunk(Cc, lo, hi, rep.cnz, None if c is None else (xc.cuts[c], xc.cuts[c + 1]))
        if os.environ.get("GBA_BENCH_DEBUG"):
            tw1 = time.perf_counter()
            Cc.free(); g.sync()
            print(f"[bench] rows {lo}:{hi} mxm wall {1e3 * (tw1 - tw):.1f} ms kernel_ms {rep.kernel_ms:.1f} "
                  f"free {1e3 * (time.perf_counter() - tw1):.1f} ms cnz {rep.cnz} phases "
                  f"{[round(x, 1) for x in rep.phase_ms[:7]]} bins {list(rep.bin_vectors)}", file=sys.stderr)
        if collect:
            stats["phase_ms"] += np.array(rep.phase_ms[:])
            stats["cnz"] += rep.cnz
            stats["bins"] += np.array(rep.bin_vectors[:])
            stats["bin_cnz"] += np.array(rep.bin_cnz[:])
            stats["bin_bnz"] += np.array(rep.bin_bnz[:])
            stats["bin_madds"] += np.array(rep.bin_madds[:])
            stats["launches"] += 1
            stats["panels"] = stats.get("panels", 0) + rep.panels
            stats["panels_redone"] = stats.get("panels_redone", 0) + rep.panels_redone
        Cc.free()

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(False, check=(w == 0 and not args.no_verify))
    checked = args.warmup > 0 and not args.no_verify
    if checks is not None:
        for dv, *_ in checks:
            dv.free()
        checks = None
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t1
    if multi:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")      # gloo reduces host tensors
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        agg = torch.tensor([stats["cnz"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(agg)
        cnz_all = int(agg.item()) // max(args.steps, 1)
    else:
        cnz_all = stats["cnz"] // max(args.steps, 1)
    failed = checked and not (verify["ok"] and verify["nnz_ok"])
    if multi:                                            # one failing rank fails the job
        bad = torch.tensor([1 if failed else 0], dtype=torch.int64, device=rdev)
        dist.all_reduce(bad)
        failed = bool(bad.item())

    if rank == 0:
        sec_per_step = elapsed / args.steps
        gflops = 2.0 * total_madds / sec_per_step / 1e9
        w_i, w_v = 8, 8                                  # int64 indices and fp64 values, as handed over by the API
        compulsory = (2 * nnzA + cnz_all) * (w_i + w_v) + 3 * (n + 1) * 8
        # dominant kernel of rank 0: the numeric kernel with the largest accumulated time
        names = ["num_wave_kernel", "num_block_kernel<4096,256>", "num_block_kernel<8192,512>", "num_panel_kernel"]
        k = int(np.argmax(stats["phase_ms"][3:7]))
        launches = max(stats["launches"], 1)
        k_ms = stats["phase_ms"][3 + k] / launches
        # algorithmic bytes of that kernel per launch: its B vectors and its C entries once, A once
        k_bytes = (stats["bin_bnz"][k] * (w_i + w_v) + stats["bin_cnz"][k] * (w_i + w_v)) / launches + nnzA * (w_i + w_v)
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # secondary model: every product streams its A entry (no reuse of A out of cache)
        k_stream = ((stats["bin_bnz"][k] + stats["bin_madds"][k] + stats["bin_cnz"][k]) * (w_i + w_v)) / launches
        streamed = k_stream / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        counters = measured_counters(f"RMAT-{args.scale} ef={args.edge_factor}", names[k])
        out = {
            "metric": "SpGEMM semiring GFLOP/s, RMAT C=A*A PLUS_TIMES_FP64",
            "value": round(gflops, 3), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"RMAT-{args.scale} ef={args.edge_factor} seed=42 C=A*A via gba_mxm (CSR, int64 indices), "
                                   f"{sum(len(pc) for pc in panel_chunks) if panels else len(my_chunks)} row chunks on rank 0, C released per chunk",
                       "n": n, "nnz_A": nnzA, "multiply_adds": total_madds, "nnz_C": cnz_all,
                       "parallelism": f"rowblock{world}" + ((f"+{panels}columnpanelsB" if panels else "+allgatherB") if multi else "")},
            "hbm_GBps_compulsory": round(compulsory / sec_per_step / 1e9, 2),
            "phase_ms_per_step_rank0": [round(float(x) / args.steps, 3) for x in stats["phase_ms"]],
            "numeric_bin_vectors_rank0": [int(x) // max(args.steps, 1) for x in stats["bins"]],
            "roofline": {"bound": "hbm", "kernel": names[k], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         **{kk: vv for kk, vv in counters.items() if not kk.startswith("insts_") and not kk.startswith("wave_")},
                         # what holds the kernel below the HBM line, worded from this run's phases and panel counts and
                         # from the committed counter passes of the same workload
                         "limiter": limiter_text(names[k], k_ms, stats["bin_madds"][k] / launches,
                                                 stats.get("panels", 0) / launches, counters),
                         "avg_launch_ms": round(float(k_ms), 3), "algorithmic_bytes_per_launch": int(k_bytes),
                         "streaming_model_GBps": round(streamed, 2),
                         "kernel_multiply_adds_per_launch": int(stats["bin_madds"][k] // launches)},
            "panels_rank0": [int(stats.get("panels", 0)) // max(args.steps, 1), int(stats.get("panels_redone", 0)) // max(args.steps, 1)],
            "generate_s": round(gen_s, 1),
            # every row chunk of the first warm-up step, on this rank: row sums C_I v against A_I (A v) in fp64 on the
            # host for v = 1 and a random v, within 2 (m + 1) u of each other for a row of m products; null = not run
            "verified": ((not failed) if checked else None),
            "verification": {"chunks": verify["chunks"], "rows": verify["rows"], "vectors": len(vecs),
                             "worst_error_over_bound": round(verify["worst_ratio"], 4)},
        }
        if world == 1 and not args.no_spmv and not panels:                   # second workload of the path, outside the timed region
            out["spmv"] = spmv_leg(g, torch, A, Bp, Bi, Bx, dev, args)
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed on the one-GPU run only
            out["cpu_baseline"] = cpu_baseline(A, rf, args.cpu_flops)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: the product failed its row-sum check (see \"verification\")")


def spmv_leg(g, torch, A, Bp, Bi, Bx, dev, args, iters=20):
    """y += A*x and y' += x'*A with PLUS_TIMES_FP32 on the same RMAT pattern (values narrowed to fp32), A by row
    with int64 indices, x and y full: the PageRank-iteration shape through gba_mxv / gba_vxm.  Device time per
    call from the library's HIP events (median of `iters` calls after two warm-up calls); algorithmic bytes =
    every entry of A once (index + value), the row pointers, x once and y once.  `achieved` prices the indices at the
    8 bytes the API hands over; `w_i` says what the kernel actually streamed (after the first call on a pattern the
    library keeps a 32-bit copy of the indices) and `achieved_at_w_i` prices them at that."""
    from graphblas_amd import Matrix, Type, DeviceMatrix
    from graphblas_amd.binding import gba_matrix
    n, nnz = A.nrows, A.nvals
    Ax32 = Bx.to(torch.float32)
    m = gba_matrix()
    m.vlen, m.vdim, m.nvec, m.nvals = n, n, n, nnz
    m.p, m.i, m.x = Bp.data_ptr(), Bi.data_ptr(), Ax32.data_ptr()
    m.type, m.sparsity, m.mem, m.is_csc = Type.FP32, g.Sparsity.SPARSE, 0, False
    dA = DeviceMatrix(m, owns=False)
    rng = np.random.default_rng(1)
    dx = Matrix.full_vector((rng.random(n) + 0.5).astype(np.float32)).to_device()
    dy = Matrix.full_vector(np.zeros(n, np.float32)).to_device()
    alg = nnz * (8 + 4) + (n + 1) * 8 + 2 * n * 4

    def run(fn):
        for _ in range(2):
            fn()
        ms = float(np.median([fn().kernel_ms for _ in range(iters)]))
        rep = fn()
        gbs = alg / (ms * 1e-3) / 1e9
        w_i = int(rep.index_bytes) or 8            # bytes per index the kernel streamed (4: the remembered 32-bit copy)
        streamed = alg - nnz * (8 - w_i)
        return {"method": g.Method.NAMES[rep.method], "ms": round(ms, 4), "achieved": round(gbs, 1),
                "frac": round(gbs / HBM_PEAK_GBS, 4), "GFLOPs": round(2 * nnz / ms / 1e6, 1),
                "w_i": w_i, "achieved_at_w_i": round(streamed / (ms * 1e-3) / 1e9, 1),
                "frac_at_w_i": round(streamed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    mxv = run(lambda: g.mxv(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dA, dx))
    vxm = run(lambda: g.vxm(dy, None, "PLUS_FP32", "PLUS_TIMES_FP32", dx, dA))
    dx.free(); dy.free()
    out = {"workload": f"RMAT-{args.scale} ef={args.edge_factor} seed=42 y += A*x PLUS_TIMES_FP32 via gba_mxv (CSR, int64 indices)",
           "algorithmic_bytes": int(alg), "unit": "GB/s", "peak": HBM_PEAK_GBS, "bound": "hbm"}
    out.update(mxv)
    out["vxm"] = vxm                                        # y' += x'*A on the same matrix (scatter form)
    try:                                                     # HBM bytes per launch of the gather kernel, from the committed PMC passes
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rec = json.load(f).get(f"RMAT-{args.scale} ef={args.edge_factor} spmv")
        for name, k in (rec or {}).get("kernels", {}).items():
            if name.startswith("dot_tile_kernel") and name.rstrip().endswith(", 0>"):
                out["traffic"] = k.get("traffic")
                out["lds_bank_conflict_frac"] = k.get("lds_bank_conflict_frac")
                out["counters_from"] = "profiles/" + str(rec.get("round", "")) + "_rmat22_counters_per_kernel.csv"
    except (OSError, ValueError, KeyError):
        pass
    return out


def measured_counters(workload, kernel):
    """Per launch of `kernel`, from the committed rocprofv3 PMC passes of this same command (profiles/hbm_traffic.json,
    written by profiles/summarize_round.py): HBM bytes (`traffic`), the share of LDS cycles lost to bank conflicts and
    the waves resident per busy CU.  `traffic` is None when this workload has not been profiled."""
    out = {"traffic": None}
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rec = json.load(f).get(workload)
        if rec and rec["kernel"].split("<")[0] == kernel.split("<")[0]:
            out["traffic"] = int(rec["traffic"])
            for key in ("lds_bank_conflict_frac", "waves_per_busy_cu", "insts_valu", "insts_salu", "insts_lds",
                        "wave_issue_frac", "wave_wait_frac"):
                if key in rec:
                    out[key] = rec[key]
            out["counters_from"] = "profiles/" + str(rec.get("round", "")) + "_rmat22_counters_per_kernel.csv"
    except (OSError, ValueError, KeyError):
        pass
    return out


def limiter_text(kernel, ms, madds, panels, counters):
    """One sentence on what bounds the dominant kernel, put together from numbers: this run's time, multiply-adds and
    LDS panels per launch, and (when the committed counter passes cover this workload) the instructions issued per
    64 products and the share of cycles in which the resident waves issue or wait."""
    if ms <= 0 or madds <= 0:
        return None
    parts = [f"{kernel}: {ms:.0f} ms per launch for {madds:.3g} multiply-adds = {ms * 1e9 / madds:.1f} ps per product"]
    if kernel == "num_panel_kernel" and panels > 0:
        parts.append(f"{panels:.0f} LDS panels per launch, each a chain of dependent phases (cursor search, run queue, windows "
                     f"of products, emit) on one 1024-thread workgroup per CU: {ms * 1e3 / panels * 256:.1f} us of CU time and "
                     f"{madds / panels:.0f} products per panel")
    if counters.get("insts_valu"):
        per64 = lambda key: counters.get(key, 0) * 64.0 / madds
        parts.append(f"issue-bound, not HBM-bound: {per64('insts_valu'):.0f} vector + {per64('insts_salu'):.0f} scalar + "
                     f"{per64('insts_lds'):.0f} LDS instructions per 64 products; the resident waves issue in "
                     f"{100 * counters.get('wave_issue_frac', 0):.0f} % and wait in {100 * counters.get('wave_wait_frac', 0):.0f} % "
                     f"of their cycles ({counters.get('counters_from', 'profiles/').strip()})")
    return "; ".join(parts)


def cpu_baseline(A, rf, target_madds):
    """The CPU oracle (restatement of the reference's coarse Gustavson saxpy3, OpenMP) on a bounded
    sample of the same workload: every stride-th row of A times the whole A."""
    from graphblas_amd import Matrix, Type
    from oracle import oracle
    n = A.nrows
    total = float(rf.sum())
    stride = max(int(round(total / max(target_madds, 1.0))), 1)
    rows = np.arange(0, n, stride)
    lens = np.diff(A.p)[rows]
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([np.arange(A.p[r], A.p[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
    As = Matrix(n, len(rows), Type.FP64, A.sparsity, False, p=p, i=A.i[idx], x=A.x[idx])
    madds = int(rf[rows].sum())
    t = time.perf_counter()
    C = oracle.saxpy3(None, False, False, A, As, "PLUS_TIMES_FP64")      # internal operands: (A_I, B_I) = (A, sample rows)
    dt = time.perf_counter() - t
    return {"value": round(2.0 * madds / dt / 1e9, 4), "unit": "GFLOP/s", "cores": oracle.num_threads(), "kind": "port",
            # what pins the port: the reference's printed results for its two user-type demos, replayed by the oracle,
            # and the recorded results of the reference library on RMAT inputs (PLUS_TIMES_FP64 saxpy3: this very kernel)
            "validated_by": "oracle vs Demo/Output/gauss_demo.out and wildtype_demo.out (tests/test_oracle.py) and vs "
                            "recorded runs of the reference library: nnz(C), sum(C), multiply count of C=A*A at RMAT-10/16 "
                            "(tests/test_reference_runs.py); the reference library itself does not build here",
            "sample": f"rows 0,{stride},{2 * stride},... of A ({len(rows)} rows, {madds} multiply-adds, "
                      f"{100.0 * madds / total:.2f}% of the job, nnz(C sample)={C.nvals}) times the whole A; {dt:.1f} s",
            "seconds": round(dt, 2)}


if __name__ == "__main__":
    main()
