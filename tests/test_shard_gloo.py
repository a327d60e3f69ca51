"""N>1 path on CPU: world_size-2 gloo processes shard a panel, each computes its windows
(the ORACLE stands in for the GPU here, as the checker's compute), and the gather to rank 0
must reproduce the single-process table row for row."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Tw, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import hmc_jl_amd  # noqa: F401
    from hmc_jl_amd import shard, synth
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = shard.partition_windows(Tw, world)
    mine = parts[rank]
    rows = []
    for gid in mine:
        y, _ = synth.generate_window(int(Tw[gid]) + 12, 3, 20240000 + gid)
        r = oracle.estimate_window(y[:Tw[gid]], 3, 2, 8, yreal=[y[Tw[gid] + 11]], window_id=gid)
        rows.append(r["summary"])
    table = shard.gather_blocks(torch.tensor(np.array(rows)), mine, len(Tw))
    if rank == 0:
        q.put(table.numpy())
    else:
        assert table is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_gather_equals_unsharded(oracle):
    import torch.multiprocessing as mp
    from hmc_jl_amd import synth
    Tw = np.array([60, 200, 90, 150, 75, 120, 64], dtype=np.int32)      # ragged, odd count
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, Tw, q)) for r in range(2)]
    for p in procs:
        p.start()
    table = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = []
    for gid in range(len(Tw)):
        y, _ = synth.generate_window(int(Tw[gid]) + 12, 3, 20240000 + gid)
        ref.append(oracle.estimate_window(y[:Tw[gid]], 3, 2, 8, yreal=[y[Tw[gid] + 11]], window_id=gid)["summary"])
    assert np.array_equal(table, np.array(ref))


def test_gather_single_process():
    import torch
    from hmc_jl_amd import shard
    blk = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    out = shard.gather_blocks(blk, [3, 0, 2, 1], 4)
    assert torch.equal(out[3], blk[0]) and torch.equal(out[0], blk[1]) and torch.equal(out[1], blk[3])


def test_bench_parent_spawns_ranks_without_touching_the_gpu():
    """`python bench.py --gpus 2` outside torchrun: the parent starts two fresh ranks (torch.distributed.run, rendezvous
    on 127.0.0.1) and relays their outcome.  Without a GPU the ranks stop with bench.py's own message -- which proves
    the spawn path ran end to end; the parent itself never imports torch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert "needs a GPU" in (r.stderr + r.stdout)
