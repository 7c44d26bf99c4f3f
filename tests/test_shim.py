"""The drop-in layer, shim/grb_shim.c: GrB_mxm / GrB_mxv / GrB_vxm with the reference's own prototypes, in front of an
unchanged libgraphblas.  It is compiled here against the reference's public header only (Include/GraphBLAS.h; the
reference library itself is not built): a prototype that differs from the header's is a compile error (the file
assigns the three functions to pointers of the header's types), and the object may need nothing beyond the public
GrB_/GxB_ interface, the gba_ C ABI and libc."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INCLUDE = "/root/reference/Include"
SHIM = os.path.join(ROOT, "shim", "grb_shim.c")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_INCLUDE, "GraphBLAS.h")),
                                reason="the reference's public header is not on this machine")


@pytest.fixture(scope="module")
def shim_object(tmp_path_factory):
    obj = str(tmp_path_factory.mktemp("shim") / "grb_shim.o")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-fPIC", "-Wall", "-Wextra", "-Werror", "-c", "-I" + REF_INCLUDE,
                        "-I" + os.path.join(ROOT, "include"), SHIM, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return obj


def _symbols(obj, flag):
    out = subprocess.run(["nm", flag, obj], capture_output=True, text=True, check=True).stdout
    return [line.split()[-1] for line in out.splitlines() if line.strip()]


def test_shim_defines_the_three_entry_points_with_the_header_prototypes(shim_object):
    defined = set(_symbols(shim_object, "--defined-only"))
    assert {"GrB_mxm", "GrB_mxv", "GrB_vxm"} <= defined
    text = open(SHIM).read()
    for f in ("mxm", "vxm", "mxv"):                      # the compile-time type check is in the file
        assert re.search(rf"static const shim_{f}_f shim_check_{f} = GrB_{f} ;", text)


def test_shim_needs_only_public_interfaces(shim_object):
    libc = {"dlsym", "getenv", "strcmp", "strncmp", "strncpy", "strrchr", "strtol", "atoi", "memset", "memcpy", "malloc", "free", "strlen",
            "pthread_mutex_lock", "pthread_mutex_unlock",
            "__stack_chk_fail", "_GLOBAL_OFFSET_TABLE_"}
    stray = [s for s in _symbols(shim_object, "-u") if not (s.startswith(("GrB_", "GxB_", "gba_")) or s in libc)]
    assert not stray, stray
    # every gba_ symbol it calls is declared in the C ABI header
    header = open(os.path.join(ROOT, "include", "graphblas_amd.h")).read()
    for s in _symbols(shim_object, "-u"):
        if s.startswith("gba_"):
            assert re.search(rf"\b{s}\s*\(", header), s
    # and every GrB_/GxB_ symbol in the reference's public header
    ref = open(os.path.join(REF_INCLUDE, "GraphBLAS.h")).read()
    for s in _symbols(shim_object, "-u"):
        if s.startswith(("GrB_", "GxB_")):
            assert re.search(rf"\b{s}\b", ref), s


def test_shim_links_as_a_shared_library(shim_object, tmp_path):
    so = str(tmp_path / "libgrb_shim.so")
    r = subprocess.run(["gcc", "-shared", shim_object, "-o", so, "-ldl", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_the_test_double_compiles_against_the_header_and_serves_the_shim(shim_object, tmp_path):
    """tests/fake_grb.c stands in for the CPU library when the shim is executed (tests/test_gpu_shim_run.py).  Its
    definitions take their prototypes from the reference's header (a mismatch does not compile), and every GrB_ / GxB_
    function the shim calls is defined there, so the shim links against it with nothing unresolved but the gba_ C ABI."""
    fake = str(tmp_path / "libfake_grb.so")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-fPIC", "-Wall", "-Wextra", "-Werror", "-shared", "-I" + REF_INCLUDE,
                        os.path.join(ROOT, "tests", "fake_grb.c"), "-o", fake], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    provided = set(_symbols(fake, "--defined-only"))
    needed = [s for s in _symbols(shim_object, "-u") if s.startswith(("GrB_", "GxB_"))]
    assert not [s for s in needed if s not in provided]
