import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblas_amd as g
from oracle import oracle
from util import dense_of
import test_gpu_fuzz as F
g.init(0)
seed = int(sys.argv[1])
sr, A, B, C0, M, accum, desc, vector = F._case(seed)
def run(M_, accum_, desc_):
    dA, dB, dC = A.to_device(), B.to_device(), C0.to_device()
    dM = M_.to_device() if M_ is not None else None
    f = g.mxv if vector else g.mxm
    try:
        rep = f(dC, dM, accum_, sr, dA, dB, desc_)
    except Exception as e:
        return "ERR " + str(e)
    p, v = oracle.mxm_spec(C0, M_, accum_, sr, A, B, desc_)
    gp, gv = dense_of(dC.to_host())
    ok = np.array_equal(gp, p) and np.array_equal(gv[p], v[p])
    return f"{'OK ' if ok else 'BAD'} method={g.Method.NAMES[rep.method]} swapped={rep.swapped} applied={rep.mask_applied} inplace={rep.done_in_place} aT={rep.a_transposed} bT={rep.b_transposed} got={gp.sum()} want={p.sum()}"
print("case", sr, accum, desc, "A", A.sparsity, A.is_csc, "B", B.sparsity, B.is_csc, "C", C0.sparsity, C0.is_csc, C0.nvals, "M", None if M is None else (M.sparsity, M.is_csc))
print("full     ", run(M, accum, desc))
print("no accum ", run(M, None, desc))
nd = desc.replace("C", "").replace("S", "")
print("no mask  ", run(None, accum, nd))
print("plain    ", run(None, None, nd))
print("struct   ", run(M, accum, desc + ("" if "S" in desc else "S")))
print("no comp  ", run(M, accum, desc.replace("C", "")))
