"""TEST INFRASTRUCTURE ONLY -- ctypes access to oracle/liboracle.so, the CPU
restatement of the reference's multiply path (oracle/grb_oracle.c).  Imported
by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py; never
by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from graphblas_amd.binding import gba_matrix, gba_semiring, gba_binop, NP_TYPES, Sparsity, semiring as _semiring, binop as _binop
from graphblas_amd.host import Matrix, _host_from_view

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class orc_semiring(C.Structure):
    _fields_ = [("fadd", C.c_void_p), ("fmult", C.c_void_p), ("add_op", C.c_int), ("mult_op", C.c_int),
                ("positional", C.c_int), ("xtype", C.c_int), ("ytype", C.c_int), ("ztype", C.c_int),
                ("xsize", C.c_size_t), ("ysize", C.c_size_t), ("zsize", C.c_size_t),
                ("a_pattern", C.c_bool), ("b_pattern", C.c_bool)]


_dll = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            build()
        d = C.CDLL(LIB_PATH)
        P = C.POINTER
        M, S = P(gba_matrix), P(orc_semiring)
        d.orc_semiring_builtin.argtypes = [S, P(gba_semiring), C.c_bool]
        d.orc_saxpy3_flopcount.argtypes = [P(C.c_int64), C.c_void_p, M, C.c_bool, M, M]
        d.orc_saxpy3.argtypes = [M, M, C.c_bool, C.c_bool, M, M, S]
        d.orc_dot3.argtypes = [M, M, C.c_bool, M, M, S]
        d.orc_dot2.argtypes = [M, M, C.c_bool, C.c_bool, M, M, S]
        d.orc_dot4.argtypes = [M, M, M, S]
        d.orc_saxpy4.argtypes = [M, M, M, S]
        d.orc_saxpy5.argtypes = [M, M, M, S]
        d.orc_saxbit.argtypes = [M, M, C.c_bool, C.c_bool, M, M, S]
        d.orc_mxm_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, M, M, P(gba_binop),
                                   P(gba_semiring), M, C.c_bool, M, C.c_bool, C.c_bool, C.c_bool, C.c_bool, C.c_bool]
        d.orc_cast.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        d.orc_cast.restype = None
        d.orc_matrix_wait.argtypes = [M]
        d.orc_matrix_free_content.argtypes = [M]
        d.orc_gauss_replay.argtypes = [C.c_char_p, C.c_size_t]
        d.orc_wildtype_replay.argtypes = [C.c_char_p, C.c_size_t]
        d.orc_num_threads.restype = C.c_int
        _dll = d
    return _dll


def _sr(semiring, flipxy=False):
    s = semiring if isinstance(semiring, gba_semiring) else _semiring(semiring)
    o = orc_semiring()
    if dll().orc_semiring_builtin(C.byref(o), C.byref(s), flipxy) != 0:
        raise ValueError("semiring is not in the oracle's built-in set")
    return o


def _take(view):
    """Copy an oracle-allocated result to numpy and free the C side."""
    def reader(ptr, nbytes):
        return C.string_at(ptr, nbytes)
    out = _host_from_view(view, reader)
    dll().orc_matrix_free_content(C.byref(view))
    return out


def _v(M, keep):
    return None if M is None else C.byref(_keepview(M, keep))


def _keepview(M, keep):
    v = M.view(keep)
    keep.append(v)
    return v


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed ({rc})")


def saxpy3(M, comp, struct, A, B, semiring, wait=False):
    keep, out = [], gba_matrix()
    _chk(dll().orc_saxpy3(C.byref(out), _v(M, keep), comp, struct, _v(A, keep), _v(B, keep), C.byref(_sr(semiring))), "saxpy3")
    if wait:
        _chk(dll().orc_matrix_wait(C.byref(out)), "wait")
    return _take(out)


def saxpy3_flopcount(M, comp, A, B):
    keep = []
    fl = np.zeros(B.vdim + 1, np.int64)
    mw = C.c_int64(0)
    _chk(dll().orc_saxpy3_flopcount(C.byref(mw), fl.ctypes.data, _v(M, keep), comp, _v(A, keep), _v(B, keep)), "flopcount")
    return mw.value, fl


def dot3(M, struct, A, B, semiring):
    keep, out = [], gba_matrix()
    _chk(dll().orc_dot3(C.byref(out), _v(M, keep), struct, _v(A, keep), _v(B, keep), C.byref(_sr(semiring))), "dot3")
    return _take(out)


def dot2(M, comp, struct, A, B, semiring):
    keep, out = [], gba_matrix()
    _chk(dll().orc_dot2(C.byref(out), _v(M, keep), comp, struct, _v(A, keep), _v(B, keep), C.byref(_sr(semiring))), "dot2")
    return _take(out)


def saxbit(M, comp, struct, A, B, semiring):
    keep, out = [], gba_matrix()
    _chk(dll().orc_saxbit(C.byref(out), _v(M, keep), comp, struct, _v(A, keep), _v(B, keep), C.byref(_sr(semiring))), "saxbit")
    return _take(out)


def _inplace(fn, what, Cm, A, B, semiring):
    """C += ... on a copy of the full matrix Cm; returns the updated Matrix."""
    out = Matrix(Cm.vlen, Cm.vdim, Cm.type, Cm.sparsity, Cm.is_csc, x=Cm.x.copy())
    keep = []
    _chk(fn(_v(out, keep), _v(A, keep), _v(B, keep), C.byref(_sr(semiring))), what)
    return out


def dot4(Cm, A, B, semiring):
    return _inplace(dll().orc_dot4, "dot4", Cm, A, B, semiring)


def saxpy4(Cm, A, B, semiring):
    return _inplace(dll().orc_saxpy4, "saxpy4", Cm, A, B, semiring)


def saxpy5(Cm, A, B, semiring):
    return _inplace(dll().orc_saxpy5, "saxpy5", Cm, A, B, semiring)


def mxm_spec(Cm, Mask, accum, semiring, A, B, desc="", flipxy=False, A_tran=None, B_tran=None):
    """C<Mask> = accum(C, A*B) from the definition (Test/GB_spec_mxm.m).  Returns
    (present, value) as nrows-by-ncols numpy arrays."""
    s = semiring if isinstance(semiring, gba_semiring) else _semiring(semiring)
    a = accum if (accum is None or isinstance(accum, gba_binop)) else _binop(accum)
    d = desc.upper() if isinstance(desc, str) else ""
    if A_tran is None:
        A_tran = "T0" in d
    if B_tran is None:
        B_tran = "T1" in d
    nrows, ncols = Cm.nrows, Cm.ncols
    inner = A.nrows if A_tran else A.ncols
    present = np.zeros(nrows * ncols, np.int8)
    value = np.zeros(nrows * ncols, NP_TYPES[Cm.type])
    keep = []
    _chk(dll().orc_mxm_spec(present.ctypes.data, value.ctypes.data, nrows, ncols, inner, _v(Cm, keep), _v(Mask, keep),
                            C.byref(a) if a is not None else None, C.byref(s), _v(A, keep), A_tran, _v(B, keep), B_tran,
                            flipxy, "R" in d, "C" in d, "S" in d), "mxm_spec")
    return present.reshape(nrows, ncols).astype(bool), value.reshape(nrows, ncols)


def cast(values, from_type, to_type):
    """Typecast an array of a built-in type entry by entry (Source/GB_casting.h)."""
    src = np.ascontiguousarray(values, NP_TYPES[from_type])
    out = np.zeros(src.shape, NP_TYPES[to_type])
    f, t, d = src.dtype.itemsize, out.dtype.itemsize, dll()
    for q in range(src.size):
        d.orc_cast(out.ctypes.data + q * t, int(to_type), src.ctypes.data + q * f, int(from_type))
    return out


def gauss_replay():
    buf = C.create_string_buffer(1 << 16)
    _chk(dll().orc_gauss_replay(buf, len(buf)), "gauss_replay")
    return buf.value.decode()


def wildtype_replay():
    buf = C.create_string_buffer(1 << 16)
    _chk(dll().orc_wildtype_replay(buf, len(buf)), "wildtype_replay")
    return buf.value.decode()


def num_threads():
    return dll().orc_num_threads()
