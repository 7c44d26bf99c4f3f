"""CPU checks of the drop-in boundary: libgraphblas_amd.so loads without a GPU, exports every
function include/graphblas_amd.h declares, the ctypes mirror has the C layout, and the library
refuses to compute when there is no device (it has no CPU path)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import graphblas_amd as g
from graphblas_amd import binding as b

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "graphblas_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gba_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared_functions()
    for must in ("gba_mxm", "gba_mxv", "gba_vxm", "gba_AxB_saxpy3", "gba_AxB_saxpy3_flopcount", "gba_AxB_dot3",
                 "gba_AxB_dot2", "gba_AxB_dot4", "gba_AxB_saxpy4", "gba_AxB_saxpy5", "gba_AxB_saxbit", "gba_cumsum",
                 "gba_matrix_wait", "gba_transpose", "gba_init", "gba_malloc", "gba_free"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(b.LIB_PATH), "build with `make -C graphblas_amd/csrc` or __graft_entry__.build()"
    dll = C.CDLL(b.LIB_PATH)
    missing = [n for n in _declared_functions() if not hasattr(dll, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_covers_the_header():
    b.lib._load()
    assert sorted(b.lib.symbols) == _declared_functions()


def test_struct_layout_matches_c(tmp_path):
    fields = {
        "gba_matrix": [f for f, _ in b.gba_matrix._fields_],
        "gba_semiring": [f for f, _ in b.gba_semiring._fields_],
        "gba_binop": [f for f, _ in b.gba_binop._fields_],
        "gba_descriptor": [f for f, _ in b.gba_descriptor._fields_],
        "gba_report": [f for f, _ in b.gba_report._fields_],
    }
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for s, fs in fields.items():
        src.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fs:
            src.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    src.append("return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", str(c), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for s in fields:
        cls = getattr(b, s)
        assert int(out[s]) == C.sizeof(cls), s
        for f in fields[s]:
            assert int(out[f"{s}.{f}"]) == getattr(cls, f).offset, f"{s}.{f}"


def test_enum_values_are_the_reference_codes():
    text = open(HEADER).read()
    def val(name):
        m = re.search(rf"\b{name}\s*=\s*(-?\d+)", text)
        assert m, name
        return int(m.group(1))
    assert val("GBA_SUCCESS") == 0 and val("GBA_NO_VALUE") == 1 and val("GBA_DIMENSION_MISMATCH") == -6
    assert val("GBA_PANIC") == -101 and val("GBA_OUT_OF_MEMORY") == -102 and val("GBA_NOT_IMPLEMENTED") == -8
    assert val("GBA_FP64") == b.Type.FP64 == 11 and val("GBA_BOOL") == 1 and val("GBA_UDT") == 14
    assert val("GBA_FIRST") == b.Op.FIRST == 72 and val("GBA_PLUS") == b.Op.PLUS == 78 and val("GBA_TIMES") == 81
    assert val("GBA_FIRSTI") == b.Op.FIRSTI == 115 and val("GBA_SECONDJ1") == 122
    assert val("GBA_SPARSE") == 2 and val("GBA_BITMAP") == 4 and val("GBA_FULL") == 8 and val("GBA_HYPERSPARSE") == 1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_a_cpu_path():
    """Runs in a child process: the HIP runtime may abort rather than return when no device node exists."""
    code = (
        "import graphblas_amd as g, numpy as np\n"
        "from graphblas_amd import binding as b\n"
        "rc = b.lib.gba_init(0)\n"
        "assert rc != 0, 'gba_init succeeded without a GPU'\n"
        "assert b.lib.gba_malloc(64) is None\n"
        "n = np.zeros(4, np.int64)\n"
        "rc2 = b.lib.gba_cumsum(n.ctypes.data, 3)\n"
        "assert rc2 != 0, 'compute entry point ran without gba_init'\n"
        "print('LOUD', rc, b.lib.gba_last_error().decode())\n"
    )
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 or "LOUD" in r.stdout, r.stdout + r.stderr
    assert "Traceback" not in r.stderr or "AssertionError" not in r.stderr, r.stderr


def test_semiring_and_descriptor_parsing():
    s = g.semiring("PLUS_TIMES_FP64")
    assert (s.add, s.mult, s.xtype, s.ytype, s.ztype) == (78, 81, 11, 11, 11)
    s = g.semiring("LOR_GT_INT32")                     # comparator: inputs INT32, result BOOL
    assert (s.add, s.mult, s.xtype, s.ztype) == (b.Op.LOR, b.Op.GT, b.Type.INT32, b.Type.BOOL)
    s = g.semiring("ANY_SECONDI_INT64")
    assert s.mult == b.Op.SECONDI and s.ztype == b.Type.INT64
    d = g.descriptor("RSC", axb=b.Desc.AxB_HASH)
    assert (d.out, d.mask, d.in0, d.in1, d.axb) == (b.Desc.REPLACE, b.Desc.COMP_STRUCTURE, 0, 0, 7084)
    d = g.descriptor("T0T1")
    assert (d.in0, d.in1, d.mask) == (b.Desc.TRAN, b.Desc.TRAN, 0)
    assert g.binop(None) is None and g.binop("PLUS_FP32").type == b.Type.FP32


def test_generators_are_deterministic():
    from graphblas_amd import gen
    A1, A2 = gen.rmat(8, 16, seed=42), gen.rmat(8, 16, seed=42)
    assert np.array_equal(A1.p, A2.p) and np.array_equal(A1.i, A2.i) and np.array_equal(A1.x, A2.x)
    assert A1.nrows == 256 and 0 < A1.nvals <= 16 * 256 and not A1.is_csc
    assert np.all(np.diff(A1.p) >= 0) and A1.p[-1] == A1.nvals
    for r in range(A1.nrows):                                  # sorted, duplicate-free rows
        seg = A1.i[A1.p[r]:A1.p[r + 1]]
        assert np.all(np.diff(seg) > 0)
    B = gen.rmat(8, 16, seed=43)
    assert not (np.array_equal(A1.p, B.p) and np.array_equal(A1.i, B.i))
