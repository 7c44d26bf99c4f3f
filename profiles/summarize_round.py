#!/usr/bin/env python3
"""Turns one round's counter passes into the record bench.py reads (profiles/hbm_traffic.json):

    summarize_round.py <gpurun_out/TAG> <workload key> <kernel substring> <round tag>

For the named kernel of the bench run: HBM bytes per launch (FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes, so the bytes fetched are 2 * 1024 * FETCH_SIZE), the share
of LDS cycles lost to bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) and the waves resident per busy
CU (4 * SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES: SQ_WAVE_CYCLES counts in units of four cycles), and the instruction
issue figures bench.py words its `limiter` from.  The same figures for the SpMV kernels go under "<workload> spmv"."""
import csv
import json
import os
import sys

out_dir, workload, kernel, tag = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def figures(r):
    n = float(r["launches"])
    f = lambda k: float(r[k]) if r.get(k) not in (None, "") else None
    fetch, write = f("FETCH_SIZE"), f("WRITE_SIZE")
    rec = {"kernel": r["kernel"].split("<")[0].replace("void gba::", "").replace("gba::", ""), "launches_profiled": int(n)}
    if fetch is not None and write is not None:
        rec.update(fetch_bytes_reported=int(1024 * fetch / n), fetch_bytes_corrected=int(2 * 1024 * fetch / n),
                   write_bytes=int(1024 * write / n), traffic=int((2 * 1024 * fetch + 1024 * write) / n))
    if f("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = round(f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE"), 4)
    if f("SQ_WAVE_CYCLES") and f("SQ_BUSY_CU_CYCLES"):
        rec["waves_per_busy_cu"] = round(4 * f("SQ_WAVE_CYCLES") / f("SQ_BUSY_CU_CYCLES"), 2)
    # instruction issue: wave-level instructions per launch, and the share of their resident cycles in which the waves
    # issue (SQ_ACTIVE_INST_ANY) or wait on a counter (SQ_WAIT_ANY); both tally in the unit of SQ_WAVE_CYCLES
    for key, name in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_SALU", "insts_salu"), ("SQ_INSTS_LDS", "insts_lds")):
        if f(key):
            rec[name] = int(f(key) / n)
    if f("SQ_WAVE_CYCLES"):
        if f("SQ_ACTIVE_INST_ANY"):
            rec["wave_issue_frac"] = round(f("SQ_ACTIVE_INST_ANY") / f("SQ_WAVE_CYCLES"), 4)
        if f("SQ_WAIT_ANY"):
            rec["wave_wait_frac"] = round(f("SQ_WAIT_ANY") / f("SQ_WAVE_CYCLES"), 4)
    return rec


path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
db = json.load(open(path))
main = [r for r in rows(os.path.join(out_dir, "counters_per_kernel.csv")) if kernel in r["kernel"]]
if main:
    rec = figures(main[0]); rec["round"] = tag
    db[workload] = rec
spmv = {}
for r in rows(os.path.join(out_dir, "spmv_counters_per_kernel.csv")):
    rec = figures(r)
    spmv[r["kernel"].replace("void gba::", "").replace("gba::", "")[:60]] = rec
if spmv:
    db[workload + " spmv"] = {"round": tag, "kernels": spmv}
json.dump(db, open(path, "w"), indent=2)
print(json.dumps(db.get(workload), indent=1))
