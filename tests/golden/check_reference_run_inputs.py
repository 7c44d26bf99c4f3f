"""Shows that graphblas_amd.gen.rmat_lcg rebuilds the inputs of the recorded runs of the reference library:
prints the input sizes beside the ones in reference_run_checksums.json.  (The fixture itself was not made by a
script: it is transcribed from the record of the reference's own runs.)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from graphblas_amd import gen, Type  # noqa: E402

G = json.load(open(os.path.join(os.path.dirname(__file__), "reference_run_checksums.json")))
for case in G["spgemm_plus_times_fp64"]:
    A = gen.rmat_lcg(case["scale"])
    print(f"RMAT-{case['scale']}: nnz(A) = {A.nvals}, recorded {case['nnz_A']}")
for case in G["triangle_count_plus_pair_int64"]:
    S = gen.rmat_lcg(case["scale"], symmetric=True, drop_diagonal=True, type=Type.INT64)
    L = gen.tril_strict(S)
    print(f"RMAT-{case['scale']} symmetrised: nnz(L) = {L.nvals}, recorded {case['nnz_L']}")
