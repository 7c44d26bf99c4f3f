"""A*(A*A) at RMAT-16: the right operand is the jumbled result of the first product.  GBA_SAXPY3_KEEP_JUMBLED=1
multiplies it as it is (memory-side accumulators for the heavy vectors), the default sorts a copy first."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import graphblas_amd as g
from graphblas_amd import gen, Type, DeviceMatrix
g.init(0)
A = gen.rmat(16, 8, seed=42, type=Type.FP64)
dA = A.to_device()
dC = DeviceMatrix.empty(A.nrows, A.ncols, Type.FP64)
g.mxm(dC, None, None, "PLUS_TIMES_FP64", dA, dA, None)
print("C nvals", dC.view.nvals, "jumbled", dC.view.jumbled)
for it in range(3):
    dD = DeviceMatrix.empty(A.nrows, A.ncols, Type.FP64)
    g.sync(); t=time.perf_counter()
    rep = g.mxm(dD, None, None, "PLUS_TIMES_FP64", dA, dC, None)
    g.sync(); print("A*C wall ms", round((time.perf_counter()-t)*1e3,1), "kernel ms", round(rep.kernel_ms,1), "nvals", dD.view.nvals, "method", g.Method.NAMES[rep.method], "flops", rep.flops)
    dD.free()
