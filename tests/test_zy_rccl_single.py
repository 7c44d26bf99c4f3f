"""The collectives of graphblas_amd.dist on device tensors through RCCL itself (backend "nccl"), with the one
rank a one-GPU box allows: communicator set-up, all_gather_into_tensor on the padded flat buffers, all_reduce
MIN / MAX / SUM, barrier, and a gathered right-hand operand handed to gba_mxm.  The multi-rank logic (who sends
what, where the blocks land) is covered by the 2-rank gloo tests; this shows that the calls the N > 1 bench makes
are accepted by RCCL with these dtypes and shapes on the MI355X, and that the library comes up after RCCL.
The work runs in one child process (tests/rccl_single_worker.py): torch has to be imported before the library is
loaded, as bench.py does, so that both use the HIP runtime that torch ships."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_collectives_and_library_in_one_rank():
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_worker.py")
    r = subprocess.run([sys.executable, "-u", worker], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "collectives ok" in r.stdout and "gather ok" in r.stdout, r.stdout
