#!/bin/bash
# A/B of tuning builds on the GPU box:  profiles/ab.sh <tag> <scale> name1 name2 ...   (libraries graphblas_amd/ab/libgba_<name>.so)
# One bench line per variant under gpurun_out/<tag>/, and a one-line summary of each on stdout.
TAG=$1; SCALE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$TAG
for n in "$@"; do
    GBA_LIB_PATH=$R/graphblas_amd/ab/libgba_$n.so timeout -k 10 300 python3 $R/bench.py --scale $SCALE --steps 1 --warmup 1 --no-cpu-baseline --no-spmv $AB_EXTRA \
        > $R/gpurun_out/$TAG/$n.json 2> $R/gpurun_out/$TAG/$n.err || { echo "$n FAILED"; tail -3 $R/gpurun_out/$TAG/$n.err; exit 1; }
    python3 - $R/gpurun_out/$TAG/$n.json $n <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms", b["ms_per_step"], "GF", b["value"], "phases", [round(x) for x in b["phase_ms_per_step_rank0"]], "panels", b["panels_rank0"], "nnzC", b["config"]["nnz_C"], "verified", b.get("verified"), b.get("verification", {}).get("worst_error_over_bound"), flush=True)
PY
done
