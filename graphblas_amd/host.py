"""Host-side mirror of the reference's matrix objects and of GrB_mxm/mxv/vxm.

``Matrix`` holds the arrays of one GraphBLAS matrix on the host in exactly the
layout GxB_Matrix_unpack_CSR/CSC/BitmapR/FullR hands out
(Include/GraphBLAS.h:9869-9881): p/i/x with int64 indices, int8 bitmap.
``DeviceMatrix`` is the same matrix in HBM (arrays from gba_malloc) plus the
gba_matrix view the C ABI takes.  The multiply functions keep the reference's
argument order: mxm(C, Mask, accum, semiring, A, B, desc).
"""
import ctypes as C
import numpy as np

from .binding import (lib, Type, Sparsity, NP_TYPES, TYPE_OF_NP, gba_matrix, gba_report, gba_semiring, gba_user_semiring,
                      gba_binop, gba_descriptor, semiring as _semiring, binop as _binop, descriptor as _descriptor)

MEM_DEVICE, MEM_HOST = 0, 1


def _as(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Matrix:
    """A matrix or vector on the host.  vlen/vdim/is_csc as in Source/Shared/GB_matrix.h:213-234."""

    def __init__(self, vlen, vdim, type, sparsity, is_csc, p=None, i=None, x=None, b=None,
                 iso=False, jumbled=False, nzombies=0, nvals=None, h=None):
        self.vlen, self.vdim, self.type, self.sparsity, self.is_csc = int(vlen), int(vdim), int(type), int(sparsity), bool(is_csc)
        self.p, self.i, self.b, self.h = _as(p, np.int64), _as(i, np.int64), _as(b, np.int8), _as(h, np.int64)
        self.x = None if x is None else np.ascontiguousarray(x, dtype=NP_TYPES[self.type])
        self.iso, self.jumbled, self.nzombies = bool(iso), bool(jumbled), int(nzombies)
        if nvals is None:
            if sparsity in (Sparsity.SPARSE, Sparsity.HYPERSPARSE):
                nvals = 0 if self.i is None else len(self.i)
            elif sparsity == Sparsity.BITMAP:
                nvals = int(np.count_nonzero(self.b))
            else:
                nvals = self.vlen * self.vdim
        self.nvals = int(nvals)

    # ---- constructors ------------------------------------------------------
    @property
    def nvec(self):
        return len(self.h) if self.sparsity == Sparsity.HYPERSPARSE else self.vdim

    def to_hyper(self):
        """The same sparse matrix in hypersparse form: only the non-empty vectors are listed
        (Source/Shared/GB_matrix.h:306-360)."""
        assert self.sparsity == Sparsity.SPARSE
        keep = np.nonzero(np.diff(self.p))[0]
        p = np.concatenate([self.p[keep], self.p[-1:]]) if len(keep) else np.zeros(1, np.int64)
        return Matrix(self.vlen, self.vdim, self.type, Sparsity.HYPERSPARSE, self.is_csc, p=p, i=self.i, x=self.x,
                      iso=self.iso, jumbled=self.jumbled, nzombies=self.nzombies, nvals=self.nvals, h=keep)

    @property
    def nrows(self):
        return self.vlen if self.is_csc else self.vdim

    @property
    def ncols(self):
        return self.vdim if self.is_csc else self.vlen

    @staticmethod
    def from_scipy(S, type=None, by_row=True):
        """scipy.sparse matrix -> sparse Matrix, CSR (the reference's default, Source/GB_init.c:173) or CSC."""
        import scipy.sparse as sp
        S = sp.csr_matrix(S) if by_row else sp.csc_matrix(S)
        S.sort_indices()
        t = type if type is not None else TYPE_OF_NP[np.dtype(S.dtype)]
        vdim, vlen = (S.shape[0], S.shape[1]) if by_row else (S.shape[1], S.shape[0])
        return Matrix(vlen, vdim, t, Sparsity.SPARSE, not by_row, p=S.indptr, i=S.indices, x=S.data.astype(NP_TYPES[t]))

    @staticmethod
    def sparse_empty(nrows, ncols, type, by_row=True):
        vdim, vlen = (nrows, ncols) if by_row else (ncols, nrows)
        return Matrix(vlen, vdim, type, Sparsity.SPARSE, not by_row, p=np.zeros(vdim + 1, np.int64),
                      i=np.zeros(0, np.int64), x=np.zeros(0, NP_TYPES[type]))

    @staticmethod
    def full_vector(x, type=None):
        x = np.asarray(x)
        t = type if type is not None else TYPE_OF_NP[np.dtype(x.dtype)]
        return Matrix(len(x), 1, t, Sparsity.FULL, True, x=x)

    @staticmethod
    def bitmap_vector(x, b, type=None):
        x = np.asarray(x)
        t = type if type is not None else TYPE_OF_NP[np.dtype(x.dtype)]
        return Matrix(len(x), 1, t, Sparsity.BITMAP, True, x=x, b=np.asarray(b, np.int8))

    @staticmethod
    def sparse_vector(n, idx, x, type=None):
        x = np.asarray(x)
        t = type if type is not None else TYPE_OF_NP[np.dtype(x.dtype)]
        idx = np.asarray(idx, np.int64)
        o = np.argsort(idx, kind="stable")
        return Matrix(n, 1, t, Sparsity.SPARSE, True, p=np.array([0, len(idx)], np.int64), i=idx[o], x=x[o])

    @staticmethod
    def dense(arr, present=None, by_row=True, type=None):
        """2-D numpy array -> full (present is None) or bitmap Matrix."""
        arr = np.asarray(arr)
        t = type if type is not None else TYPE_OF_NP[np.dtype(arr.dtype)]
        nrows, ncols = arr.shape
        lay = arr if by_row else arr.T           # vector-major: vector j contiguous
        vdim, vlen = (nrows, ncols) if by_row else (ncols, nrows)
        x = np.ascontiguousarray(lay).reshape(-1)
        if present is None:
            return Matrix(vlen, vdim, t, Sparsity.FULL, not by_row, x=x)
        pb = np.asarray(present, bool)
        pb = pb if by_row else pb.T
        return Matrix(vlen, vdim, t, Sparsity.BITMAP, not by_row, x=x, b=np.ascontiguousarray(pb).reshape(-1).astype(np.int8))

    # ---- conversions -------------------------------------------------------
    def coo(self):
        """Canonical (row, col, value) triples sorted by (row, col); zombies dropped."""
        if self.sparsity in (Sparsity.SPARSE, Sparsity.HYPERSPARSE):
            counts = np.diff(self.p)
            names = np.arange(self.vdim, dtype=np.int64) if self.sparsity == Sparsity.SPARSE else self.h
            vec = np.repeat(names, counts)
            idx = self.i[: self.p[-1]]
            live = idx >= 0
            vec, idx = vec[live], idx[live]
            vals = np.broadcast_to(self.x[:1], idx.shape) if self.iso else self.x[: self.p[-1]][live]
        else:
            n = self.vlen * self.vdim
            pres = np.ones(n, bool) if self.sparsity == Sparsity.FULL else (self.b[:n] != 0)
            pos = np.nonzero(pres)[0]
            vec, idx = pos // max(self.vlen, 1), pos % max(self.vlen, 1)
            vals = np.broadcast_to(self.x[:1], pos.shape) if self.iso else self.x[pos]
        rows, cols = (idx, vec) if self.is_csc else (vec, idx)
        o = np.lexsort((cols, rows))
        return rows[o], cols[o], np.asarray(vals)[o]

    def to_scipy(self):
        import scipy.sparse as sp
        r, c, v = self.coo()
        return sp.csr_matrix((v, (r, c)), shape=(self.nrows, self.ncols))

    def view(self, keep):
        """gba_matrix over the host arrays (mem = host); keep collects references."""
        m = gba_matrix()
        m.vlen, m.vdim, m.nvec, m.nvals = self.vlen, self.vdim, self.nvec, self.nvals
        for name in ("p", "i", "x", "b", "h"):
            a = getattr(self, name)
            if a is not None:
                if a.size == 0:
                    a = np.zeros(1, a.dtype)
                keep.append(a)
                setattr(m, name, a.ctypes.data)
        m.nzombies, m.type, m.sparsity, m.mem = self.nzombies, self.type, self.sparsity, MEM_HOST
        m.is_csc, m.iso, m.jumbled = self.is_csc, self.iso, self.jumbled
        return m

    def to_device(self):
        return DeviceMatrix.upload(self)


def _host_from_view(m, reader):
    """Build a Matrix from a gba_matrix whose arrays are read through reader(ptr, nbytes)."""
    t = m.type
    dt = np.dtype(NP_TYPES[t])

    def arr(ptr, count, dtype):
        if not ptr or count <= 0:
            return np.zeros(0, dtype)
        return np.frombuffer(reader(ptr, int(count) * np.dtype(dtype).itemsize), dtype=dtype).copy()

    dense = m.vlen * m.vdim
    if m.sparsity in (Sparsity.SPARSE, Sparsity.HYPERSPARSE):
        p = arr(m.p, m.nvec + 1, np.int64)
        nz = int(p[-1]) if len(p) else 0
        h = arr(m.h, m.nvec, np.int64) if m.sparsity == Sparsity.HYPERSPARSE else None
        return Matrix(m.vlen, m.vdim, t, m.sparsity, m.is_csc, p=p, i=arr(m.i, nz, np.int64), h=h,
                      x=arr(m.x, 1 if m.iso else nz, dt), iso=m.iso, jumbled=m.jumbled, nzombies=m.nzombies, nvals=nz)
    b = arr(m.b, dense, np.int8) if m.sparsity == Sparsity.BITMAP else None
    return Matrix(m.vlen, m.vdim, t, m.sparsity, m.is_csc, x=arr(m.x, 1 if m.iso else dense, dt), b=b,
                  iso=m.iso, nvals=m.nvals)


class DeviceMatrix:
    """A matrix resident in HBM; owns its arrays unless constructed with owns=False."""

    def __init__(self, view, owns=True):
        self.view, self.owns = view, owns

    @staticmethod
    def upload(M):
        m = gba_matrix()
        m.vlen, m.vdim, m.nvec, m.nvals = M.vlen, M.vdim, M.nvec, M.nvals
        for name in ("p", "i", "x", "b", "h"):
            a = getattr(M, name)
            if a is None:
                continue
            nbytes = max(a.nbytes, 1)
            d = lib.gba_malloc(nbytes)
            if not d:
                raise MemoryError(f"gba_malloc({nbytes}) failed: {lib.gba_last_error().decode()}")
            if a.nbytes:
                lib.check(lib.gba_memcpy_h2d(d, a.ctypes.data, a.nbytes))
            setattr(m, name, d)
        m.nzombies, m.type, m.sparsity, m.mem = M.nzombies, M.type, M.sparsity, MEM_DEVICE
        m.is_csc, m.iso, m.jumbled = M.is_csc, M.iso, M.jumbled
        return DeviceMatrix(m)

    @staticmethod
    def empty(nrows, ncols, type, by_row=True):
        """Empty sparse result holder (GrB_Matrix_new analogue)."""
        return Matrix.sparse_empty(nrows, ncols, type, by_row).to_device()

    def to_host(self):
        def reader(ptr, nbytes):
            buf = (C.c_char * nbytes)()
            lib.check(lib.gba_memcpy_d2h(C.cast(buf, C.c_void_p), ptr, nbytes))
            return bytes(buf)
        return _host_from_view(self.view, reader)

    def row_slice(self, v0, v1):
        """Zero-copy view of vectors [v0, v1) of a sparse matrix: the row block a rank owns."""
        m = gba_matrix.from_buffer_copy(self.view)
        assert m.sparsity == Sparsity.SPARSE
        m.p = self.view.p + 8 * v0
        m.vdim = m.nvec = v1 - v0
        ends = np.zeros(2, np.int64)     # p is not rebased: entries keep their absolute positions in i/x
        lib.check(lib.gba_memcpy_d2h(ends.ctypes.data, self.view.p + 8 * v0, 8))
        lib.check(lib.gba_memcpy_d2h(ends.ctypes.data + 8, self.view.p + 8 * v1, 8))
        m.nvals = int(ends[1] - ends[0])
        return DeviceMatrix(m, owns=False)

    def free(self):
        if self.owns and self.view is not None:
            lib.gba_matrix_free_content(C.byref(self.view))
        self.view = None

    def ref(self):
        return C.byref(self.view)


def _ref(x):
    return None if x is None else x.ref()


def _sr(s):
    return s if isinstance(s, gba_semiring) else _semiring(s)


def _acc(a):
    if a is None or isinstance(a, gba_binop):
        return a
    return _binop(a)


def _desc(d):
    if d is None or isinstance(d, gba_descriptor):
        return d
    return _descriptor(d)


def _call3(fn, C_, Mask, accum, semiring, A, B, desc):
    rep = gba_report()
    s, a, d = _sr(semiring), _acc(accum), _desc(desc)
    info = fn(C_.ref(), _ref(Mask), C.byref(a) if a is not None else None, C.byref(s), A.ref(), B.ref(),
              C.byref(d) if d is not None else None, C.byref(rep))
    lib.check(info)
    return rep


def mxm(C_, Mask, accum, semiring, A, B, desc=None):
    """C<Mask> = accum (C, A*B) -- GrB_mxm (Include/GraphBLAS.h:4640-4649)."""
    return _call3(lib.gba_mxm, C_, Mask, accum, semiring, A, B, desc)


def mxv(w, mask, accum, semiring, A, u, desc=None):
    """w<mask> = accum (w, A*u) -- GrB_mxv (Include/GraphBLAS.h:4662-4671)."""
    return _call3(lib.gba_mxv, w, mask, accum, semiring, A, u, desc)


def vxm(w, mask, accum, semiring, u, A, desc=None):
    """w'<mask'> = accum (w', u'*A) -- GrB_vxm (Include/GraphBLAS.h:4651-4660)."""
    return _call3(lib.gba_vxm, w, mask, accum, semiring, u, A, desc)


def register_udt(dtype):
    """numpy dtype that host copies of GBA_UDT matrices use (one user-defined type at a time)."""
    dt = np.dtype(dtype)
    NP_TYPES[Type.UDT] = dt
    TYPE_OF_NP[dt] = Type.UDT
    return dt


def user_semiring(type_name, type_defn, type_size, add_name, add_defn, mult_name, mult_defn):
    s = gba_user_semiring()
    s.type_name, s.type_defn, s.type_size = type_name.encode(), type_defn.encode(), type_size
    s.add_name, s.add_defn = add_name.encode(), add_defn.encode()
    s.mult_name, s.mult_defn = mult_name.encode(), mult_defn.encode()
    return s


def mxm_udt(C_, Mask, accum, semiring, A, B, desc=None):
    """C<Mask> = accum (C, A*B) over a user-defined semiring; accum is None or (name, defn)."""
    rep = gba_report()
    d = _desc(desc)
    an, ad = (accum[0].encode(), accum[1].encode()) if accum else (None, None)
    lib.check(lib.gba_mxm_udt(C_.ref(), _ref(Mask), an, ad, C.byref(semiring), A.ref(), B.ref(),
                              C.byref(d) if d is not None else None, C.byref(rep)))
    return rep


def vxm_udt(w, mask, accum, semiring, u, A, desc=None):
    """w'<mask'> = accum (w', u'*A) over a user-defined semiring (the multiplier sees its operands as z = f (u(k), A(k,j)))."""
    rep = gba_report()
    d = _desc(desc)
    an, ad = (accum[0].encode(), accum[1].encode()) if accum else (None, None)
    lib.check(lib.gba_vxm_udt(w.ref(), _ref(mask), an, ad, C.byref(semiring), u.ref(), A.ref(),
                              C.byref(d) if d is not None else None, C.byref(rep)))
    return rep


def mxm_select(C_, Mask, accum, semiring, A, B, desc=None):
    """Method the dispatcher would take for this GrB_mxm call (no compute)."""
    return _call3(lib.gba_mxm_select, C_, Mask, accum, semiring, A, B, desc)


def _new_result(type):
    m = gba_matrix()
    m.type, m.mem = type, MEM_DEVICE
    return DeviceMatrix(m)


def AxB_saxpy3(M, Mask_comp, Mask_struct, A, B, semiring, axb_method=0, do_sort=0):
    s = _sr(semiring)
    out, rep, applied = _new_result(s.ztype), gba_report(), C.c_bool(False)
    lib.check(lib.gba_AxB_saxpy3(out.ref(), _ref(M), Mask_comp, Mask_struct, A.ref(), B.ref(), C.byref(s),
                                 axb_method, do_sort, C.byref(applied), C.byref(rep)))
    return out, rep, applied.value


def AxB_saxpy3_flopcount(M, Mask_comp, A, B):
    n = B.view.nvec
    d = lib.gba_malloc(8 * (n + 1))
    mwork = C.c_int64(0)
    try:
        lib.check(lib.gba_AxB_saxpy3_flopcount(C.byref(mwork), d, _ref(M), Mask_comp, A.ref(), B.ref()))
        out = np.zeros(n + 1, np.int64)
        lib.check(lib.gba_memcpy_d2h(out.ctypes.data, d, out.nbytes))
    finally:
        lib.gba_free(d)
    return mwork.value, out


def AxB_dot3(M, Mask_struct, A, B, semiring):
    s = _sr(semiring)
    out, rep = _new_result(s.ztype), gba_report()
    lib.check(lib.gba_AxB_dot3(out.ref(), M.ref(), Mask_struct, A.ref(), B.ref(), C.byref(s), C.byref(rep)))
    return out, rep


def AxB_dot2(M, Mask_comp, Mask_struct, A, B, semiring):
    s = _sr(semiring)
    out, rep = _new_result(s.ztype), gba_report()
    lib.check(lib.gba_AxB_dot2(out.ref(), _ref(M), Mask_comp, Mask_struct, A.ref(), B.ref(), C.byref(s), C.byref(rep)))
    return out, rep


def _inplace(fn, C_, A, B, semiring):
    s, rep = _sr(semiring), gba_report()
    info = fn(C_.ref(), A.ref(), B.ref(), C.byref(s), C.byref(rep))
    if info == 1:
        return None                       # GrB_NO_VALUE: method declined
    lib.check(info)
    return rep


def AxB_dot4(C_, A, B, semiring):
    return _inplace(lib.gba_AxB_dot4, C_, A, B, semiring)


def AxB_saxpy4(C_, A, B, semiring):
    return _inplace(lib.gba_AxB_saxpy4, C_, A, B, semiring)


def AxB_saxpy5(C_, A, B, semiring):
    return _inplace(lib.gba_AxB_saxpy5, C_, A, B, semiring)


def AxB_saxbit(M, Mask_comp, Mask_struct, A, B, semiring):
    s = _sr(semiring)
    out, rep = _new_result(s.ztype), gba_report()
    lib.check(lib.gba_AxB_saxbit(out.ref(), _ref(M), Mask_comp, Mask_struct, A.ref(), B.ref(), C.byref(s), C.byref(rep)))
    return out, rep


def cumsum(count):
    """GB_cumsum on the device: returns the n+1 exclusive sums of count[0..n)."""
    a = np.ascontiguousarray(count, np.int64)
    n = len(a)
    buf = np.concatenate([a, np.zeros(1, np.int64)])
    d = lib.gba_malloc(buf.nbytes)
    try:
        lib.check(lib.gba_memcpy_h2d(d, buf.ctypes.data, buf.nbytes))
        lib.check(lib.gba_cumsum(d, n))
        lib.check(lib.gba_memcpy_d2h(buf.ctypes.data, d, buf.nbytes))
    finally:
        lib.gba_free(d)
    return buf


def matrix_wait(A):
    lib.check(lib.gba_matrix_wait(A.ref()))


def matrix_forget(A):
    """Drop what the library remembers about the pattern of A (to be called before p or i of A are rewritten in place)."""
    lib.check(lib.gba_matrix_forget(A.ref()))


def transpose(A):
    out = _new_result(A.view.type)
    lib.check(lib.gba_transpose(out.ref(), A.ref()))
    return out


def partition_by_flops(nparts, A, B):
    part = (C.c_int64 * (nparts + 1))()
    lib.check(lib.gba_partition_by_flops(part, nparts, A.ref(), B.ref()))
    return list(part)


def sync():
    lib.check(lib.gba_sync())


def timer_start():
    lib.check(lib.gba_timer_start())


def timer_stop():
    ms = C.c_double(0)
    lib.check(lib.gba_timer_stop(C.byref(ms)))
    return ms.value
