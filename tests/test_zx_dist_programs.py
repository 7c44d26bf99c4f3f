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


def _record(what, gpus, env_extra):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench_paths.py"), what, "--gpus", str(gpus), "--scale", "14", "--iters", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("what,keys", [("tricount", ["triangles"]), ("bfs", ["visited"]), ("sssp", ["iterations", "reached"])])
def test_two_ranks_on_one_device_agree_with_one_rank(what, keys):
    """Two ranks share device 0 and talk over gloo (RCCL refuses two ranks on one device): the row blocks, the
    library calls on them and the exchanges are those of a 2-GPU run; only the transport differs."""
    one = _record(what, 1, {"GBA_BENCH_FORCE_DIST": "1"})
    two = _record(what, 2, {"GBA_BENCH_REHEARSE": "1"})
    assert two["n_gpus"] == 2 and two["collective"]
    for k in keys:
        assert one[k] == two[k], (k, one[k], two[k])
    if what == "bfs":
        assert [lv["found"] for lv in one["levels"]] == [lv["found"] for lv in two["levels"]]


def _spgemm_record(gpus, extra, env_extra):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--scale", "14", "--steps", "1", "--warmup", "1",
                        "--chunk-flops", "2e7", "--no-cpu-baseline", "--no-spmv"] + extra,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_column_panels_of_B_give_the_product_of_the_whole_B():
    """C = A*A with the right-hand operand exchanged one column panel at a time (bench.py --b-panels): every panel
    product passes the row-sum check against the host, and the panels together hold as many entries as the product
    against the gathered B -- with one rank over RCCL, and with two ranks on device 0 over gloo."""
    whole = _spgemm_record(1, [], {"GBA_BENCH_FORCE_DIST": "1"})
    assert whole["verified"] is True
    for gpus, env in ((1, {"GBA_BENCH_FORCE_DIST": "1"}), (2, {"GBA_BENCH_REHEARSE": "1"})):
        cut = _spgemm_record(gpus, ["--b-panels", "3"], env)
        assert cut["verified"] is True and cut["n_gpus"] == gpus
        assert cut["config"]["nnz_C"] == whole["config"]["nnz_C"]
        assert "3columnpanelsB" in cut["config"]["parallelism"]
        assert cut["verification"]["chunks"] >= 3
