"""graphblas_amd -- MI355X (gfx950) back end for the GrB_mxm / GrB_mxv / GrB_vxm
hot path of SuiteSparse:GraphBLAS.

The product is the C-ABI library ``libgraphblas_amd.so`` (include/graphblas_amd.h).
This package is a thin ctypes harness over it for tests and benchmarks: it
mirrors the reference's call shapes (GrB_mxm(C, Mask, accum, semiring, A, B, desc),
Include/GraphBLAS.h:4640-4671) and its GrB_Info error codes.  There is no CPU
fallback: importing the bindings without the built library raises.
"""
from .binding import (  # noqa: F401
    lib, GrbError, Info, Type, Op, Sparsity, Desc, Method,
    gba_matrix, gba_semiring, gba_binop, gba_descriptor, gba_report, gba_user_semiring,
    semiring, binop, descriptor, init, finalize, NP_TYPES, TYPE_OF_NP,
)
from .host import (  # noqa: F401
    Matrix, DeviceMatrix, mxm, mxv, vxm, mxm_select, mxm_udt, vxm_udt, register_udt, user_semiring,
    AxB_saxpy3, AxB_saxpy3_flopcount, AxB_dot3, AxB_dot2, AxB_dot4, AxB_saxpy4, AxB_saxpy5, AxB_saxbit,
    cumsum, matrix_wait, matrix_forget, transpose, partition_by_flops, sync, timer_start, timer_stop,
)
