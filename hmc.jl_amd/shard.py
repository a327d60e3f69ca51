"""Window sharding over the GPUs of one node.

Windows are independent chains -- exactly how the reference fans out today, one SLURM
array task per end-date index (slurmscripts/base_estimation.sh:5) -- so the data path has
no collective: each rank (one process per GPU) samples its own windows, keyed by GLOBAL
window ids so the sharded run reproduces the unsharded one draw for draw.  The only
exchange is the final gather of per-window result blocks to rank 0 (RCCL over xGMI when
the tensors are on the GPU; gloo in the CPU tests).
"""
import numpy as np


def partition_windows(T, world_size):
    """Static LPT partition: windows sorted by length (descending, stable), dealt to the
    currently lightest rank.  Returns a list (one per rank) of ascending global ids.
    Equal-length panels reduce to near-contiguous equal blocks."""
    T = np.asarray(T, dtype=np.int64)
    order = np.argsort(-T, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    count = np.zeros(world_size, dtype=np.int64)
    parts = [[] for _ in range(world_size)]
    cap = -(-len(T) // world_size)        # balance counts too: per-window overheads are not all O(T)
    for w in order:
        cand = [r for r in range(world_size) if count[r] < cap]
        r = min(cand, key=lambda q: (load[q], q))
        parts[r].append(int(w))
        load[r] += T[w]
        count[r] += 1
    return [sorted(p) for p in parts]


def contiguous_blocks(W, world_size):
    """Equal-T panels: rank r takes the contiguous block [r*W/G, (r+1)*W/G)."""
    edges = [(W * r) // world_size for r in range(world_size + 1)]
    return [list(range(edges[r], edges[r + 1])) for r in range(world_size)]


def gather_blocks(local_block, local_ids, W_total, group=None, dst=0, force_collective=False):
    """Gather per-window rows to rank `dst`, placed at their global ids.

    local_block: torch tensor (n_local, C) on this rank (cuda -> RCCL, cpu -> gloo).
    local_ids:   the global window ids of its rows.
    Returns a (W_total, C) tensor on rank dst, None elsewhere.  Blocks are padded to a
    common row count so one all_gather moves everything (ranks differ by at most one
    window under partition_windows).
    force_collective: take the collective branch (padding, ids column, dist.gather) at world size 1 too -- lets a one-GPU
    box run RCCL's load, communicator creation and this function's whole exchange path (tests/test_gpu_rccl.py)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ids = torch.as_tensor(list(local_ids), dtype=torch.int64, device=local_block.device)
    if world == 1 and not (force_collective and dist.is_initialized()):
        out = torch.zeros((W_total, local_block.shape[1]), dtype=local_block.dtype, device=local_block.device)
        out[ids] = local_block
        return out
    rank = dist.get_rank(group)
    if local_block.is_cuda and dist.get_backend(group) == "gloo":
        local_block = local_block.cpu()          # gloo gathers host tensors (CPU tests, one-GPU rehearsals)
        ids = ids.cpu()
    cap = -(-W_total // world)
    C = local_block.shape[1]
    pad = torch.zeros((cap, C + 1), dtype=local_block.dtype, device=local_block.device)
    pad[:, C] = -1.0
    n = local_block.shape[0]
    pad[:n, :C] = local_block
    pad[:n, C] = ids.to(local_block.dtype)           # ids ride along as an exact small integer column
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    # place every row at its global id in one indexed store; padding rows (id -1) go to a spare row that is cut off again --
    # no boolean-mask indexing, which would make the host wait for the device once per rank
    allb = torch.cat(bufs, dim=0)
    idx = allb[:, C].to(torch.int64)
    idx = torch.where(idx >= 0, idx, torch.full_like(idx, W_total))
    out = torch.zeros((W_total + 1, C), dtype=local_block.dtype, device=local_block.device)
    out[idx] = allb[:, :C]
    return out[:W_total]
