#!/bin/bash
# Hardware-counter passes over one bench command on the GPU box (one rocprofv3 run per pass: the
# counter blocks hold 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 GRBM counters at a time).
#   profiles/pmc_passes.sh <out-dir under gpurun_out> <bench.py arguments...>     (GBA_PMC_PROG=bench_paths.py runs that instead)
# Each pass leaves <out>/<pass>/*counter_collection.csv; profiles/pmc_summary.py sums them per kernel.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
PASS[sq2]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC"
PASS[mem]="TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TA_BUSY_sum"
PASS[occ]="SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
PASS[ta]="TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
PASS[tcc]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
PASS[icache]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
for p in ${GBA_PMC_PASSES:-sq1 sq2 mem occ fetch write}; do
    echo "[pmc] pass $p: ${PASS[$p]}"
    timeout -s KILL 240 rocprofv3 --pmc ${PASS[$p]} --kernel-trace -d "$OUT/$p" --output-format csv -- \
        python3 "$GRAFT_REPO_ROOT/${GBA_PMC_PROG:-bench.py}" "$@" > "$OUT/$p.json" 2> "$OUT/$p.err" || { echo "[pmc] pass $p failed"; tail -5 "$OUT/$p.err"; exit 1; }
done
