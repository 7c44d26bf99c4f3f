"""Diagnostic: libgraphblas_amd.so next to PyTorch + RCCL in one process.  `nccl_first` (default) brings RCCL up and then
the library, `gba_first` the other way round; torch is imported first in both, so one HIP runtime serves all."""
import os, socket, ctypes, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'rccl' in l or 'hsa-runtime' in l})
print("before:", maps())
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl_first"
import graphblas_amd as g
if mode == "gba_first":
    g.init(0); print("gba init ok (first)")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("nccl ok", t.tolist())
print("env:", {k: v for k, v in os.environ.items() if "VISIBLE" in k or k.startswith("HSA") or k.startswith("HIP") or k.startswith("ROC")})
print("after nccl:", maps())
try:
    if mode != "gba_first":
        g.init(0); print("gba init ok (after nccl)")
    from graphblas_amd import Matrix, Type
    import numpy as np
    y = Matrix.full_vector(np.ones(8)).to_device(); print("to_device ok", y.to_host().x.sum())
except Exception as e:
    print("FAILED:", e)
dist.destroy_process_group()
