"""GPU test of the one entry point that allocates nothing: gba_cumsum_noalloc, the prefix sum for callers that
cannot call hipMalloc (HIP graph capture, serving loops).  Called through the C ABI and compared, bit for bit,
with gba_cumsum on the same input: two calls in a row, the second on a smaller input (the built-in scratch
array keeps the first call's tile sums), then the size limit."""
import ctypes

import numpy as np
import pytest

from graphblas_amd.binding import lib, Info

pytestmark = pytest.mark.gpu


def _scan_both(count):
    """(gba_cumsum, gba_cumsum_noalloc) of the same counts, each on its own device buffer."""
    a = np.ascontiguousarray(count, np.int64)
    n = len(a)
    out = []
    for fn in (lib.gba_cumsum, lib.gba_cumsum_noalloc):
        buf = np.concatenate([a, np.full(1, -7, np.int64)])          # slot n is overwritten by the total
        d = lib.gba_malloc(buf.nbytes)
        try:
            lib.check(lib.gba_memcpy_h2d(d, buf.ctypes.data, buf.nbytes))
            lib.check(fn(d, n))
            lib.check(lib.gba_sync())                                   # the allocation-free form only queues its kernels
            lib.check(lib.gba_memcpy_d2h(buf.ctypes.data, d, buf.nbytes))
        finally:
            lib.gba_free(d)
        out.append(buf)
    return out


def test_cumsum_noalloc_matches_cumsum_on_two_calls_in_a_row(gba):
    rng = np.random.default_rng(7)
    for n in (5_000_001, 70_000, 1500, 0, 2047, 2048, 4_200_000):      # three scan levels, two, one; tile edges; again large
        c = rng.integers(0, 1 << 20, n)
        want, got = _scan_both(c)
        assert np.array_equal(want, np.concatenate([[0], np.cumsum(c)]).astype(np.int64)), n
        assert np.array_equal(got, want), n


def test_cumsum_noalloc_rejects_sizes_beyond_its_scratch(gba):
    d = lib.gba_malloc(64)
    try:
        rc = lib.gba_cumsum_noalloc(d, 65535 * 2048)                    # one slot more than the scratch array serves
        assert rc == Info.INVALID_VALUE
        assert b"exceeds the limit" in lib.gba_last_error()
    finally:
        lib.gba_free(d)
