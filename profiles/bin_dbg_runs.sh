cd /tmp && export TMPDIR=/tmp
for d in 0 1 4 8 16 24; do
  echo "== GBA_BIN_DBG=$d"
  GBA_BIN_DBG=$d GBA_LIB_PATH=$GRAFT_REPO_ROOT/graphblas_amd/libgba_ticks.so timeout -k 10 120 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2_dbg_$d --output-format csv -- python3 $GRAFT_REPO_ROOT/bench_paths.py spmv --type fp64 --iters 3 > /dev/null 2>&1 || { echo "run failed"; exit 1; }
  python3 $GRAFT_REPO_ROOT/profiles/kernel_times.py $GRAFT_REPO_ROOT/gpurun_out/r2_dbg_$d bin_scatter bin_reduce bin_count | tail -3 | cut -c1-40,88-130
done
