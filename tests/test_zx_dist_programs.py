"""The sharded programs of bench_paths_dist.py (spmv, bfs, sssp, tricount by row blocks with the exchanges of
graphblas_amd.dist) run with one rank over RCCL on the GPU: same results as without a process group, and as scipy.
Who sends what with more than one rank is covered by the 2-rank gloo tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sharded_programs_over_rccl_equal_the_plain_programs_and_scipy():
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_programs_worker.py")
    r = subprocess.run([sys.executable, "-u", worker, "14"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "one rank over RCCL equals the plain run" in r.stdout and "programs equal scipy" in r.stdout, r.stdout
