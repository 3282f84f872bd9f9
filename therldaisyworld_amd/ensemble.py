"""Ensemble sharding: independent worlds split over the GPUs of a node, one process per GPU.

Worlds never interact (every operation of the path is per world; the only shared quantity is the
scalar luminosity, a deterministic function of the step count — SURVEY.md §8e), so the data path has
NO collective: each rank owns a contiguous block of worlds, keyed for the device RNG by its global
world offset, and steps it independently.  The only communication is a gather of small per-world
results (lifespans, reductions) at the end of a run, over ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
"""
from __future__ import annotations

import os

import numpy as np


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_worlds(total_worlds: int, rank: int, world_size: int):
    """Contiguous block [offset, offset+count) of `total_worlds` for `rank`; the first
    ``total_worlds % world_size`` ranks get one extra world.  Blocks tile the range exactly."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(int(total_worlds), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def init_process_group(backend: str | None = None, device: int | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).  `device`: the
    GPU of this rank when it is not LOCAL_RANK (rehearsals with several ranks on one GPU)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = rank_info()
    if world == 1:
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank if device is None else device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def gather_per_world(local: np.ndarray, counts=None, device=None) -> np.ndarray:
    """Concatenate per-world arrays (leading axis = this rank's worlds) from all ranks, in rank
    order, on every rank.  Ragged shards are handled by padding to the largest shard."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.copy()
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(ns, n_local)
    ns = [int(n.item()) for n in ns]
    if counts is not None and list(counts) != ns:
        raise RuntimeError(f"shard sizes {ns} do not match the expected {list(counts)}")
    nmax = max(ns)
    flat = local.reshape(local.shape[0], -1)
    as_bytes = np.ascontiguousarray(flat).view(np.uint8).reshape(local.shape[0], -1)
    pad = np.zeros((nmax, as_bytes.shape[1]), dtype=np.uint8)
    pad[: local.shape[0]] = as_bytes
    t = torch.from_numpy(pad).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    parts = []
    for n, o in zip(ns, outs):
        b = o.cpu().numpy()[:n]
        parts.append(np.ascontiguousarray(b).view(local.dtype).reshape((n,) + local.shape[1:]))
    return np.concatenate(parts, axis=0)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
