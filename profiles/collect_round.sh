#!/bin/bash
# One round's evidence from the GPU box:  profiles/collect_round.sh <tag>   (run from the repository root)
# Leaves under gpurun_out/<tag>/: the default bench line, rocprofv3 kernel statistics of one bench step, the
# counter passes (HBM bytes, LDS bank conflicts, occupancy) summed per kernel, the same for the SpMV kernels,
# and the RMAT-22 triangle-count and BFS lines.  Copy what is to be kept into profiles/.
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
step() { echo "[collect] $*"; }
step bench && timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
step kernel-trace && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_under_kernel_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
python3 $R/profiles/kernel_times.py $OUT/trace > $OUT/kernel_times.txt
step counters && cd $R && GBA_PMC_PASSES="fetch write sq1 sq2 occ" bash profiles/pmc_passes.sh $TAG/pmc --steps 1 --warmup 0 --no-cpu-baseline --no-spmv || exit 1
python3 profiles/pmc_summary.py $OUT/pmc > $OUT/counters_per_kernel.csv
step spmv && cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/spmv_trace --output-format csv -- python3 $R/bench_paths.py spmv > $OUT/spmv.json 2> $OUT/spmv.err || { tail -5 $OUT/spmv.err; exit 1; }
python3 $R/profiles/kernel_times.py $OUT/spmv_trace dot_tile bin_ count_ones fill > $OUT/spmv_kernel_times.txt
cd $R && GBA_PMC_PROG=bench_paths.py GBA_PMC_PASSES="fetch write sq1 sq2 occ" bash profiles/pmc_passes.sh $TAG/spmv_pmc spmv --iters 3 || exit 1
python3 profiles/pmc_summary.py $OUT/spmv_pmc dot_tile bin_ > $OUT/spmv_counters_per_kernel.csv
step tricount && timeout -k 10 400 python3 bench_paths.py tricount --iters 3 > $OUT/tricount.json 2> $OUT/tricount.err || { tail -5 $OUT/tricount.err; exit 1; }
step bfs && timeout -k 10 400 python3 bench_paths.py bfs > $OUT/bfs.json 2> $OUT/bfs.err || { tail -5 $OUT/bfs.err; exit 1; }
step done
