set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3_dot3}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/tri_trace --output-format csv -- python3 $R/bench_paths.py tricount --iters 2 > $OUT/tricount.json 2> $OUT/tri.err || { tail -3 $OUT/tri.err; exit 1; }
python3 $R/profiles/kernel_times.py $OUT/tri_trace > $OUT/tricount_kernel_times.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bfs_trace --output-format csv -- python3 $R/bench_paths.py bfs > $OUT/bfs.json 2> $OUT/bfs.err || { tail -3 $OUT/bfs.err; exit 1; }
python3 $R/profiles/kernel_times.py $OUT/bfs_trace > $OUT/bfs_kernel_times.txt
cd $R && GBA_PMC_PROG=bench_paths.py GBA_PMC_PASSES="fetch write sq1 sq2" bash profiles/pmc_passes.sh ${1:-r3_dot3}/tri_pmc tricount --iters 2 || exit 1
python3 profiles/pmc_summary.py $OUT/tri_pmc dot3 expand > $OUT/tricount_counters_per_kernel.csv
head -8 $OUT/tricount_kernel_times.txt | cut -c1-150
