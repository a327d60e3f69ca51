"""RCCL on the GPU box, once: torch.distributed with the `nccl` backend (= RCCL on ROCm) at world size 1 on cuda:0, the
collective branch of shard.gather_blocks on GPU tensors (force_collective), and the MAX all-reduce bench.py uses for its
timing.  One GPU cannot show a second rank; it can show that librccl loads, that a communicator comes up on this driver
(dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0), and that the padding / ids-column / scatter logic of the exchange runs on
device tensors.  Runs in a child process: a process group must not outlive the test in the pytest process.
Reference role: the final gather of the SLURM fan-out, slurmscripts/base_estimation.sh:5,17."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
import hmc_jl_amd
from hmc_jl_amd import device as hdev, shard, synth

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
K, W, T, nrun = 3, 12, 300, 40
Y, Tw, fut = synth.generate_panel(W, T, K, window_base=100)
panel = hdev.DevicePanel(Y, Tw, K, nrun, (12,), fut[:, 11:12], device=0)
panel.run(burnin=5, window_base=100)
ids = [100 + 2 * i for i in range(W)]                      # scattered global ids: the gather places rows, it does not append
out = shard.gather_blocks(panel.summary, ids, 100 + 2 * W, force_collective=True)
assert out.is_cuda and out.shape == (100 + 2 * W, panel.summary.shape[1])
assert torch.equal(out[ids], panel.summary) and float(out.abs().sum()) == float(panel.summary.abs().sum())
ref = shard.gather_blocks(panel.summary, ids, 100 + 2 * W)        # the world-size-1 shortcut
assert torch.equal(out, ref)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.25
dist.destroy_process_group()
print("rccl ok", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_loads_and_the_gather_runs_on_gpu_tensors(hmclib):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
