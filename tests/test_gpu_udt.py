"""User-defined type and semiring, compiled at run time: the seven GrB_mxm calls of
Demo/Program/gauss_demo.c:402-470 replayed through gba_mxm_udt on the device and compared with
the reference's own printed results (Demo/Output/gauss_demo.out, fixture
tests/golden/gauss_demo_mxm.json)."""
import json
import os

import numpy as np
import pytest

import graphblas_amd as g
from graphblas_amd import Matrix, Type, Sparsity

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "gauss_demo_mxm.json")

# a Gaussian integer, its sum and its product: what the application hands to GxB_Type_new /
# GxB_BinaryOp_new as strings (Demo/Program/gauss_demo.c:22-26, 60-66, 88-96 declare the same type
# and operators)
GAUSS_DEFN = "typedef struct { int32_t real ; int32_t imag ; } gauss ;"
ADD_DEFN = """void addgauss (gauss *z, const gauss *x, const gauss *y)
{
    z->real = x->real + y->real ;
    z->imag = x->imag + y->imag ;
}"""
MULT_DEFN = """void multgauss (gauss *z, const gauss *x, const gauss *y)
{
    int32_t zr = x->real * y->real - x->imag * y->imag ;
    int32_t zi = x->real * y->imag + x->imag * y->real ;
    z->real = zr ;
    z->imag = zi ;
}"""


def _gauss_matrix(dt, entries, by_row=True):
    """entries: dict (i,j) -> (real, imag); a 4-by-4 sparse matrix held by row."""
    keys = sorted(entries)
    rows = np.array([k[0] for k in keys], np.int64)
    cols = np.array([k[1] for k in keys], np.int64)
    x = np.array([entries[k] for k in keys], dtype=dt)
    p = np.zeros(5, np.int64)
    np.add.at(p, rows + 1, 1)
    return Matrix(4, 4, Type.UDT, Sparsity.SPARSE, False, p=np.cumsum(p), i=cols, x=x)


def _as_table(M):
    r, c, v = M.coo()
    out = [[None] * 4 for _ in range(4)]
    for i, j, e in zip(r, c, v):
        out[int(i)][int(j)] = [int(e["real"]), int(e["imag"])]
    return out


def test_gauss_demo_on_the_device(gba):
    want = json.load(open(GOLDEN))
    dt = g.register_udt(np.dtype([("real", "<i4"), ("imag", "<i4")]))
    sr = g.user_semiring("gauss", GAUSS_DEFN, dt.itemsize, "addgauss", ADD_DEFN, "multgauss", MULT_DEFN)
    add = ("addgauss", ADD_DEFN)

    A = _gauss_matrix(dt, {(i, j): (i + 1, 2 - j) for i in range(4) for j in range(4) if (i, j) != (0, 0)})
    dA = A.to_device()

    # A = A*A
    g.mxm_udt(dA, None, None, sr, dA, dA)
    A2 = dA.to_host()
    assert _as_table(A2) == want[0]["matrix"], want[0]["name"]

    # C<D> = A*A', D the boolean identity
    D = Matrix(4, 4, Type.BOOL, Sparsity.SPARSE, False, p=np.arange(5), i=np.arange(4), x=np.ones(4, bool))
    dD = D.to_device()
    dC = Matrix.sparse_empty(4, 4, Type.UDT).to_device()
    g.mxm_udt(dC, dD, None, sr, dA, dA, "T1")
    assert _as_table(dC.to_host()) == want[1]["matrix"], want[1]["name"]
    dD.free()

    # D = diag (A), then C = D*A and C = A*D
    r, c, v = A2.coo()
    Dg = _gauss_matrix(dt, {(int(i), int(j)): (int(e["real"]), int(e["imag"])) for i, j, e in zip(r, c, v) if i == j})
    dDg = Dg.to_device()
    g.mxm_udt(dC, None, None, sr, dDg, dA)
    assert _as_table(dC.to_host()) == want[2]["matrix"], want[2]["name"]
    g.mxm_udt(dC, None, None, sr, dA, dDg)
    assert _as_table(dC.to_host()) == want[3]["matrix"], want[3]["name"]
    dC.free(); dDg.free()

    # C = (1,-2) everywhere (full), then C += A*A'
    ones = np.empty((4, 4), dt)
    ones["real"], ones["imag"] = 1, -2
    dC = Matrix.dense(ones).to_device()
    g.mxm_udt(dC, None, add, sr, dA, dA, "T1")
    assert _as_table(dC.to_host()) == want[4]["matrix"], want[4]["name"]

    # C += B*A and C += A*B with B = (1,-2) everywhere (full)
    dB = Matrix.dense(ones).to_device()
    g.mxm_udt(dC, None, add, sr, dB, dA)
    assert _as_table(dC.to_host()) == want[5]["matrix"], want[5]["name"]
    g.mxm_udt(dC, None, add, sr, dA, dB)
    assert _as_table(dC.to_host()) == want[6]["matrix"], want[6]["name"]
    for d in (dA, dB, dC):
        d.free()


def test_udt_compile_error_is_reported(gba):
    dt = g.register_udt(np.dtype([("real", "<i4"), ("imag", "<i4")]))
    bad = g.user_semiring("gauss", GAUSS_DEFN, dt.itemsize, "addgauss", ADD_DEFN, "multgauss", "void multgauss (gauss *z) { nonsense ; }")
    A = _gauss_matrix(dt, {(0, 1): (1, 1), (1, 0): (2, 2)})
    dA = A.to_device()
    dC = Matrix.sparse_empty(4, 4, Type.UDT).to_device()
    with pytest.raises(g.GrbError):
        g.mxm_udt(dC, None, None, bad, dA, dA)
    dA.free(); dC.free()


# ---------------------------------------------------------------- wildtype ----
# the 192-byte type and the two operators of Demo/Program/wildtype_demo.c:31-46, 112-190, as the strings
# the application hands to GxB_Type_new / GxB_BinaryOp_new
WILD_DEFN = "typedef struct { double stuff [4][4] ; char whatstuff [64] ; } wildtype ;"
WILDADD_DEFN = """void wildadd (wildtype *z, const wildtype *x, const wildtype *y)
{
   for (int i = 0 ; i < 4 ; i++)
   {
       for (int j = 0 ; j < 4 ; j++)
       {
           z->stuff [i][j] = x->stuff [i][j] + y->stuff [i][j] ;
       }
   }
   const char *psrc = "this was added" ;
   char *pdst = z->whatstuff ;
   while ((*pdst++ = *psrc++)) ;
}"""
WILDMULT_DEFN = """void wildmult (wildtype *z, const wildtype *x, const wildtype *y)
{
   for (int i = 0 ; i < 4 ; i++)
   {
       for (int j = 0 ; j < 4 ; j++)
       {
           z->stuff [i][j] = 0 ;
           for (int k = 0 ; k < 4 ; k++)
           {
               z->stuff [i][j] += (x->stuff [i][k] * y->stuff [k][j]) ;
           }
       }
   }
   const char *psrc = "this was multiplied" ;
   char *pdst = z->whatstuff ;
   while ((*pdst++ = *psrc++)) ;
}"""
WILD_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "wildtype_demo_mxm.json")


def _wild_matrix(dt, entries, n=10):
    """entries: the fixture's list of {i, j, text, stuff}; an n-by-n sparse matrix held by row."""
    es = sorted(entries, key=lambda e: (e["i"], e["j"]))
    x = np.zeros(len(es), dtype=dt)
    for q, e in enumerate(es):
        x[q]["stuff"] = np.array(e["stuff"])
        x[q]["whatstuff"] = e["text"].encode()
    p = np.zeros(n + 1, np.int64)
    np.add.at(p, np.array([e["i"] for e in es], np.int64) + 1, 1)
    return Matrix(n, n, Type.UDT, Sparsity.SPARSE, False, p=np.cumsum(p), i=np.array([e["j"] for e in es], np.int64), x=x)


def _wild_check(M, want):
    r, c, v = M.coo()
    got = sorted(zip(r.tolist(), c.tolist(), range(len(r))))
    assert [(i, j) for i, j, _ in got] == [(e["i"], e["j"]) for e in want["entries"]], want["name"]
    for (i, j, q), e in zip(got, want["entries"]):
        assert bytes(v[q]["whatstuff"]).split(b"\0")[0].decode() == e["text"], (want["name"], i, j)
        flat = [x for row in e["stuff"] for x in row]
        assert ["%.1f" % x for x in v[q]["stuff"].ravel()] == ["%.1f" % x for x in flat], (want["name"], i, j)


def test_wildtype_demo_on_the_device(gba):
    """Demo/Program/wildtype_demo.c:366 and :381 through gba_mxm_udt: C = A*B, then C<M,struct,replace> = C*C'
    with C its own operands, against the entries the reference prints (Demo/Output/wildtype_demo.out:142-330).
    The inputs printed there are exact (integers and halves); results are compared at the printed precision."""
    want = json.load(open(WILD_GOLDEN))
    dt = g.register_udt(np.dtype([("stuff", "<f8", (4, 4)), ("whatstuff", "S64")]))
    assert dt.itemsize == 192
    sr = g.user_semiring("wildtype", WILD_DEFN, dt.itemsize, "wildadd", WILDADD_DEFN, "wildmult", WILDMULT_DEFN)
    dA, dB = _wild_matrix(dt, want[0]["entries"]).to_device(), _wild_matrix(dt, want[1]["entries"]).to_device()
    dC = Matrix.sparse_empty(10, 10, Type.UDT).to_device()
    g.mxm_udt(dC, None, None, sr, dA, dB)
    _wild_check(dC.to_host(), want[2])
    Mk = Matrix(10, 10, Type.BOOL, Sparsity.SPARSE, False, p=np.array([0, 0, 0, 2, 4, 4, 4, 4, 4, 4, 4], np.int64),
                i=np.array([2, 3, 2, 3], np.int64), x=np.ones(4, bool))
    dM = Mk.to_device()
    g.mxm_udt(dC, dM, None, sr, dC, dC, "RST1")
    _wild_check(dC.to_host(), want[3])
    for d in (dA, dB, dC, dM):
        d.free()


@pytest.mark.parametrize("desc", ["", "R", "S", "RSC", "C"])
@pytest.mark.parametrize("with_accum", [False, True])
def test_gauss_into_a_sparse_C_that_holds_entries(gba, desc, with_accum):
    """C<M> = accum (C, A*B) on the user-defined type with a sparse, non-empty C: the values ride through the
    write-back as positions (Source/GB_accum_mask.c semantics).  Expected result: the same call on Gaussian
    integers held as numpy complex numbers."""
    import scipy.sparse as sp
    n = 40
    dt = g.register_udt(np.dtype([("real", "<i4"), ("imag", "<i4")]))
    sr = g.user_semiring("gauss", GAUSS_DEFN, dt.itemsize, "addgauss", ADD_DEFN, "multgauss", MULT_DEFN)
    rng = np.random.default_rng(7)

    def rand(seed, density):
        S = sp.random(n, n, density=density, random_state=np.random.default_rng(seed), format="csr").astype(np.complex128)
        S.data = rng.integers(1, 5, len(S.data)) + 1j * rng.integers(-4, 5, len(S.data))
        S.sort_indices()
        x = np.zeros(len(S.data), dt)
        x["real"], x["imag"] = S.data.real.astype(np.int32), S.data.imag.astype(np.int32)
        return S, Matrix(n, n, Type.UDT, Sparsity.SPARSE, False, p=S.indptr, i=S.indices, x=x)

    (Sa, A), (Sb, B), (Sc, C0) = rand(1, 0.1), rand(2, 0.1), rand(3, 0.2)
    Sm = sp.random(n, n, density=0.4, random_state=np.random.default_rng(4), format="csr")
    Sm.data = rng.integers(0, 2, len(Sm.data)).astype(np.float64)
    M = Matrix.from_scipy(Sm, type=Type.INT8)
    dA, dB, dC, dM = A.to_device(), B.to_device(), C0.to_device(), M.to_device()
    g.mxm_udt(dC, dM, ("addgauss", ADD_DEFN) if with_accum else None, sr, dA, dB, desc)
    r, c, v = dC.to_host().coo()
    got = np.zeros((n, n), np.complex128); gp = np.zeros((n, n), bool)
    got[r, c] = v["real"] + 1j * v["imag"]; gp[r, c] = True
    # the definition, densely
    a, b, cc = Sa.toarray(), Sb.toarray(), Sc.toarray()
    tp = ((a != 0).astype(int) @ (b != 0).astype(int)) > 0
    t = a @ b
    cp = cc != 0
    mp = (Sm.toarray() != 0) if "S" not in desc else np.zeros((n, n), bool)
    if "S" in desc:                                  # structural: every stored entry counts, zeros included
        mp[np.repeat(np.arange(n), np.diff(Sm.indptr)), Sm.indices] = True
    if "C" in desc:
        mp = ~mp
    if with_accum:
        zp, z = cp | tp, np.where(cp & tp, cc + t, np.where(cp, cc, t))
    else:
        zp, z = tp, t
    keep_c = cp & ~mp & ("R" not in desc)
    wp = (mp & zp) | keep_c
    w = np.where(mp, z, cc)
    assert np.array_equal(gp, wp), desc
    assert np.array_equal(got[wp], w[wp]), desc


AFFINE_DEFN = "typedef struct { int32_t a ; int32_t b ; } affine ;"
AFFINE_ADD = """void addaffine (affine *z, const affine *x, const affine *y)
{
    z->a = x->a + y->a ;
    z->b = x->b + y->b ;
}"""
# composition of x -> a x + b maps: not commutative, so the order of the operands shows
AFFINE_MULT = """void compose (affine *z, const affine *x, const affine *y)
{
    int32_t a = x->a * y->a ;
    int32_t b = x->a * y->b + x->b ;
    z->a = a ;
    z->b = b ;
}"""


@pytest.mark.parametrize("tran", [False, True])
def test_vxm_over_a_user_type_hands_the_multiplier_its_operands_in_vxm_order(gba, tran):
    """w' = u'*A with a multiplier that does not commute: every term is compose (u(k), A(k,j)), the vector's entry
    first (Source/GrB_vxm.c:63-72 computes A'*u with the operands exchanged, Source/GB_flip_binop.c:355-375)."""
    dt = g.register_udt(np.dtype([("a", "<i4"), ("b", "<i4")]))
    sr = g.user_semiring("affine", AFFINE_DEFN, dt.itemsize, "addaffine", AFFINE_ADD, "compose", AFFINE_MULT)
    n = 40
    rng = np.random.default_rng(11)
    dense = rng.random((n, n)) < 0.2
    rows, cols = np.nonzero(dense)
    ax = np.zeros(len(rows), dtype=dt); ax["a"] = rng.integers(-3, 4, len(rows)); ax["b"] = rng.integers(-3, 4, len(rows))
    p = np.zeros(n + 1, np.int64); np.add.at(p, rows + 1, 1)
    A = Matrix(n, n, Type.UDT, Sparsity.SPARSE, False, p=np.cumsum(p), i=cols, x=ax)          # by row
    uk = np.flatnonzero(rng.random(n) < 0.5)
    ux = np.zeros(len(uk), dtype=dt); ux["a"] = rng.integers(-3, 4, len(uk)); ux["b"] = rng.integers(-3, 4, len(uk))
    u = Matrix(n, 1, Type.UDT, Sparsity.SPARSE, True, p=np.array([0, len(uk)], np.int64), i=uk, x=ux)
    w = Matrix(n, 1, Type.UDT, Sparsity.SPARSE, True, p=np.zeros(2, np.int64), i=np.zeros(0, np.int64), x=np.zeros(0, dt))
    dA, du, dw = A.to_device(), u.to_device(), w.to_device()
    g.vxm_udt(dw, None, None, sr, du, dA, "T1" if tran else None)
    r, _, v = dw.to_host().coo()
    want = {}
    uval = {int(k): (int(x["a"]), int(x["b"])) for k, x in zip(uk, ux)}
    for i, j, x in zip(rows, cols, ax):
        k, out = (int(j), int(i)) if tran else (int(i), int(j))         # A'(k,out) = A(out,k)
        if k in uval:
            xa, xb = uval[k]
            ta, tb = xa * int(x["a"]), xa * int(x["b"]) + xb
            sa, sb = want.get(out, (0, 0))
            want[out] = (sa + ta, sb + tb)
    assert sorted(want) == [int(q) for q in r]
    for q, e in zip(r, v):
        assert (int(e["a"]), int(e["b"])) == want[int(q)], int(q)
    for d in (dA, du, dw):
        d.free()


_DISK_WORKER = r"""
import ctypes, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import graphblas_amd as g
from graphblas_amd import Matrix, Type, Sparsity
from graphblas_amd.binding import lib
sys.path.insert(0, sys.argv[1] + "/tests")
import test_gpu_udt as t
g.init(0)
dt = g.register_udt(np.dtype([("real", "<i4"), ("imag", "<i4")]))
sr = g.user_semiring("gauss", t.GAUSS_DEFN, dt.itemsize, "addgauss", t.ADD_DEFN, "multgauss", t.MULT_DEFN)
A = t._gauss_matrix(dt, {(i, j): (i + 1, 2 - j) for i in range(4) for j in range(4) if (i, j) != (0, 0)})
dA = A.to_device()
dC = Matrix(4, 4, Type.UDT, Sparsity.SPARSE, False, p=np.zeros(5, np.int64), i=np.zeros(0, np.int64), x=np.zeros(0, dt)).to_device()
g.mxm_udt(dC, None, None, sr, dA, dA)
a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
lib.gba_jit_stats(ctypes.byref(a), ctypes.byref(b))
print(json.dumps({"compiled": a.value, "from_disk": b.value, "C": t._as_table(dC.to_host())}))
"""


def test_compiled_semirings_are_kept_on_disk(gba, tmp_path):
    """The first process compiles the semiring and leaves the code object in the cache folder; the second process
    loads it from there without compiling and computes the same product."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GBA_JIT_CACHE=str(tmp_path / "jit"))
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _DISK_WORKER, root], capture_output=True, text=True, timeout=240, env=env)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        runs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert runs[0]["compiled"] >= 1 and runs[0]["from_disk"] == 0
    assert runs[1]["compiled"] == 0 and runs[1]["from_disk"] >= 1
    assert runs[0]["C"] == runs[1]["C"]
    assert any(f.suffix == ".co" for f in (tmp_path / "jit").iterdir())
    off = subprocess.run([sys.executable, "-c", _DISK_WORKER, root], capture_output=True, text=True, timeout=240,
                         env=dict(env, GBA_JIT_CACHE="0"))
    assert off.returncode == 0 and json.loads(off.stdout.splitlines()[-1])["compiled"] >= 1
