"""ctypes declarations for include/graphblas_amd.h.

Numeric codes are the reference's own (GrB_Info Include/GraphBLAS.h:318-351,
GB_Type_code / GB_Opcode Source/Shared/GB_opaque.h:25-44, 209-284,
GrB_Desc_Value Include/GraphBLAS.h:512-538).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GBA_LIB_PATH") or os.path.join(_HERE, "libgraphblas_amd.so")   # override: kernel A/B runs


class Info:
    SUCCESS = 0
    NO_VALUE = 1
    UNINITIALIZED_OBJECT = -1
    NULL_POINTER = -2
    INVALID_VALUE = -3
    INVALID_INDEX = -4
    DOMAIN_MISMATCH = -5
    DIMENSION_MISMATCH = -6
    OUTPUT_NOT_EMPTY = -7
    NOT_IMPLEMENTED = -8
    PANIC = -101
    OUT_OF_MEMORY = -102
    INSUFFICIENT_SPACE = -103
    INVALID_OBJECT = -104
    INDEX_OUT_OF_BOUNDS = -105
    EMPTY_OBJECT = -106
    _names = None

    @classmethod
    def name(cls, code):
        if cls._names is None:
            cls._names = {v: k for k, v in vars(cls).items() if isinstance(v, int)}
        return "GrB_" + cls._names.get(code, str(code))


class Type:
    BOOL, INT8, UINT8, INT16, UINT16, INT32, UINT32, INT64, UINT64, FP32, FP64, FC32, FC64, UDT = range(1, 15)


NP_TYPES = {
    Type.BOOL: np.bool_, Type.INT8: np.int8, Type.UINT8: np.uint8, Type.INT16: np.int16, Type.UINT16: np.uint16,
    Type.INT32: np.int32, Type.UINT32: np.uint32, Type.INT64: np.int64, Type.UINT64: np.uint64,
    Type.FP32: np.float32, Type.FP64: np.float64, Type.FC32: np.complex64, Type.FC64: np.complex128,
}
TYPE_OF_NP = {np.dtype(v): k for k, v in NP_TYPES.items()}
TYPE_NAMES = {"BOOL": 1, "INT8": 2, "UINT8": 3, "INT16": 4, "UINT16": 5, "INT32": 6, "UINT32": 7, "INT64": 8,
              "UINT64": 9, "FP32": 10, "FP64": 11, "FC32": 12, "FC64": 13}


class Op:
    NOP = 0
    FIRST, SECOND, ANY, PAIR, MIN, MAX, PLUS, MINUS, RMINUS, TIMES, DIV, RDIV, POW = range(72, 85)
    ISEQ, ISNE, ISGT, ISLT, ISGE, ISLE = range(85, 91)
    LOR, LAND, LXOR = 91, 92, 93
    BOR, BAND, BXOR, BXNOR = 94, 95, 96, 97
    EQ, NE, GT, LT, GE, LE = range(102, 108)
    FIRSTI, FIRSTI1, FIRSTJ, FIRSTJ1, SECONDI, SECONDI1, SECONDJ, SECONDJ1 = range(115, 123)
    USER = 123


class Sparsity:
    HYPERSPARSE, SPARSE, BITMAP, FULL = 1, 2, 4, 8


class Desc:
    DEFAULT, REPLACE, COMP, TRAN, STRUCTURE, COMP_STRUCTURE = 0, 1, 2, 3, 4, 6
    AxB_GUSTAVSON, AxB_DOT, AxB_HASH, AxB_SAXPY = 7081, 7083, 7084, 7085


class Method:
    NONE, SAXPY3, SAXBIT, SAXPY4, SAXPY5, DOT2, DOT3, DOT4, ROWSCALE, COLSCALE, ISO_FULL = range(11)
    NAMES = ["none", "saxpy3", "saxbit", "saxpy4", "saxpy5", "dot2", "dot3", "dot4", "rowscale", "colscale", "iso_full"]


class gba_matrix(C.Structure):
    _fields_ = [
        ("vlen", C.c_int64), ("vdim", C.c_int64), ("nvec", C.c_int64), ("nvals", C.c_int64),
        ("p", C.c_void_p), ("h", C.c_void_p), ("i", C.c_void_p), ("x", C.c_void_p), ("b", C.c_void_p),
        ("nzombies", C.c_int64), ("type", C.c_int32), ("sparsity", C.c_int32), ("mem", C.c_int32),
        ("is_csc", C.c_bool), ("iso", C.c_bool), ("jumbled", C.c_bool), ("hyper_switch", C.c_float),
    ]


class gba_semiring(C.Structure):
    _fields_ = [("add", C.c_int32), ("mult", C.c_int32), ("xtype", C.c_int32), ("ytype", C.c_int32), ("ztype", C.c_int32)]


class gba_binop(C.Structure):
    _fields_ = [("op", C.c_int32), ("type", C.c_int32)]


class gba_descriptor(C.Structure):
    _fields_ = [("out", C.c_int32), ("mask", C.c_int32), ("in0", C.c_int32), ("in1", C.c_int32),
                ("axb", C.c_int32), ("do_sort", C.c_int32)]


class gba_report(C.Structure):
    _fields_ = [("method", C.c_int32), ("swapped", C.c_int32), ("mask_applied", C.c_int32),
                ("done_in_place", C.c_int32), ("a_transposed", C.c_int32), ("b_transposed", C.c_int32),
                ("flops", C.c_int64), ("kernel_ms", C.c_double), ("phase_ms", C.c_double * 8),
                ("bin_vectors", C.c_int64 * 4), ("bin_cnz", C.c_int64 * 4), ("bin_bnz", C.c_int64 * 4),
                ("bin_madds", C.c_int64 * 4),
                ("panels", C.c_int64), ("panels_redone", C.c_int64), ("cnz", C.c_int64),
                ("typecast", C.c_int32), ("index_bytes", C.c_int32)]


class gba_user_semiring(C.Structure):
    _fields_ = [("type_name", C.c_char_p), ("type_defn", C.c_char_p), ("type_size", C.c_size_t),
                ("add_name", C.c_char_p), ("add_defn", C.c_char_p), ("mult_name", C.c_char_p), ("mult_defn", C.c_char_p)]


class GrbError(RuntimeError):
    def __init__(self, info, msg):
        super().__init__(f"{Info.name(info)} ({info}): {msg}")
        self.info = info


_POSITIONAL = {"FIRSTI", "FIRSTI1", "FIRSTJ", "FIRSTJ1", "SECONDI", "SECONDI1", "SECONDJ", "SECONDJ1"}
_COMPARATORS = {"EQ", "NE", "GT", "LT", "GE", "LE"}


def semiring(name):
    """'PLUS_TIMES_FP64' -> gba_semiring, with the naming rules of the built-in
    GrB_/GxB_ semirings: the trailing type is the multiplier's input type,
    comparators produce BOOL, positional multipliers produce the named int type."""
    add, mult, tname = name.upper().split("_")
    t = TYPE_NAMES[tname]
    s = gba_semiring()
    s.add, s.mult = getattr(Op, add), getattr(Op, mult)
    if mult in _POSITIONAL:
        s.xtype = s.ytype = t
        s.ztype = t
    elif mult in _COMPARATORS and t != Type.BOOL:
        s.xtype = s.ytype = t
        s.ztype = Type.BOOL
    else:
        s.xtype = s.ytype = s.ztype = t
    return s


def binop(name):
    """'PLUS_FP32' -> gba_binop; None -> no accumulator."""
    if name is None:
        return None
    op, tname = name.upper().split("_")
    b = gba_binop()
    b.op, b.type = getattr(Op, op), TYPE_NAMES[tname]
    return b


def descriptor(name=None, axb=Desc.DEFAULT, do_sort=0):
    """'RSC', 'T0', 'ST1', ... -> gba_descriptor, following the letters of the
    predefined GrB_DESC_* (Include/GraphBLAS.h:567-610)."""
    d = gba_descriptor()
    d.axb, d.do_sort = axb, do_sort
    s = (name or "").upper()
    if "R" in s:
        d.out = Desc.REPLACE
    comp, struct = "C" in s, "S" in s
    d.mask = (Desc.COMP if comp else 0) + (Desc.STRUCTURE if struct else 0)
    if "T0" in s:
        d.in0 = Desc.TRAN
    if "T1" in s:
        d.in1 = Desc.TRAN
    return d


class _Lib:
    """Lazy handle on libgraphblas_amd.so; fails loudly when it is not built."""

    def __init__(self):
        self._dll = None

    def _load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C graphblas_amd/csrc` "
                "(or __graft_entry__.build()).  graphblas_amd has no CPU fallback.")
        dll = C.CDLL(LIB_PATH)
        P = C.POINTER
        M, S, B, D, R = P(gba_matrix), P(gba_semiring), P(gba_binop), P(gba_descriptor), P(gba_report)
        i32, i64, vp, b = C.c_int32, C.c_int64, C.c_void_p, C.c_bool
        sig = {
            "gba_init": (i32, [C.c_int]), "gba_finalize": (i32, []),
            "gba_last_error": (C.c_char_p, []), "gba_version": (C.c_char_p, []),
            "gba_malloc": (vp, [C.c_size_t]), "gba_free": (None, [vp]),
            "gba_pool_set_limit": (i32, [C.c_size_t]), "gba_pool_release": (i32, []),
            "gba_pool_stats": (i32, [P(C.c_size_t), P(C.c_size_t)]),
            "gba_host_malloc": (vp, [C.c_size_t]), "gba_host_calloc": (vp, [C.c_size_t, C.c_size_t]),
            "gba_host_realloc": (vp, [vp, C.c_size_t]), "gba_host_free": (None, [vp]),
            "gba_is_device_visible": (i32, [vp]), "gba_matrix_forget": (i32, [M]),
            "gba_memcpy_h2d": (i32, [vp, vp, C.c_size_t]), "gba_memcpy_d2h": (i32, [vp, vp, C.c_size_t]),
            "gba_memset": (i32, [vp, C.c_int, C.c_size_t]), "gba_sync": (i32, []), "gba_stream": (vp, []),
            "gba_timer_start": (i32, []), "gba_timer_stop": (i32, [P(C.c_double)]),
            "gba_matrix_free_content": (None, [M]),
            "gba_mxm": (i32, [M, M, B, S, M, M, D, R]),
            "gba_mxv": (i32, [M, M, B, S, M, M, D, R]),
            "gba_vxm": (i32, [M, M, B, S, M, M, D, R]),
            "gba_mxm_select": (i32, [M, M, B, S, M, M, D, R]),
            "gba_mxm_udt": (i32, [M, M, C.c_char_p, C.c_char_p, P(gba_user_semiring), M, M, D, R]),
            "gba_jit_stats": (i32, [P(C.c_uint64), P(C.c_uint64)]),
            "gba_vxm_udt": (i32, [M, M, C.c_char_p, C.c_char_p, P(gba_user_semiring), M, M, D, R]),
            "gba_AxB_saxpy3": (i32, [M, M, b, b, M, M, S, i32, i32, P(b), R]),
            "gba_AxB_saxpy3_flopcount": (i32, [P(i64), vp, M, b, M, M]),
            "gba_AxB_dot3": (i32, [M, M, b, M, M, S, R]),
            "gba_AxB_dot2": (i32, [M, M, b, b, M, M, S, R]),
            "gba_AxB_dot4": (i32, [M, M, M, S, R]),
            "gba_AxB_saxpy4": (i32, [M, M, M, S, R]),
            "gba_AxB_saxpy5": (i32, [M, M, M, S, R]),
            "gba_AxB_saxbit": (i32, [M, M, b, b, M, M, S, R]),
            "gba_cumsum": (i32, [vp, i64]),
            "gba_cumsum_noalloc": (i32, [vp, i64]),
            "gba_matrix_wait": (i32, [M]),
            "gba_transpose": (i32, [M, M]),
            "gba_partition_by_flops": (i32, [P(i64), i32, M, M]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(dll, name)       # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        self._dll = dll
        self.symbols = sorted(sig)
        return dll

    def __getattr__(self, name):
        return getattr(self._load(), name)

    def check(self, info):
        if info != Info.SUCCESS:
            raise GrbError(info, self._load().gba_last_error().decode())


lib = _Lib()


def init(device=0):
    lib.check(lib.gba_init(int(device)))


def finalize():
    lib.check(lib.gba_finalize())
