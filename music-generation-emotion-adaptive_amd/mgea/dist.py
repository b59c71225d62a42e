"""Multi-GPU plumbing: one process per GPU, prompts sharded data-parallel, ONE RCCL broadcast of
the packed weight arena over xGMI at start-up and no cross-GPU traffic in decode.
The reference has no inference-time communication at all (its only collective is a training-time
broadcast_object_list, train/train_large.py:82-86); this is new surface, not a replacement."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  backend 'nccl' is RCCL on ROCm; 'gloo' for CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(be, rank=rank, world_size=world)
    return rank, world, local


def broadcast_arena(arena: torch.Tensor, src: int = 0) -> torch.Tensor:
    """One collective for every weight: the arena is a single contiguous fp32 buffer."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if arena.is_cuda and dist.get_backend() == "gloo":
            # rehearsal of the N > 1 path on a 1-GPU box (several ranks share the card, RCCL refuses
            # that): stage through the host.  Production is backend "nccl" = RCCL over xGMI.
            host = arena.cpu()
            dist.broadcast(host, src=src)
            arena.copy_(host)
        else:
            dist.broadcast(arena, src=src)
    return arena


def all_reduce_max(value: float, device) -> float:
    """max over ranks of a host scalar (bench.py's timing reduction)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(value: float, device) -> List[float]:
    """every rank's host scalar, in rank order, on every rank (bench.py: per-rank tokens/s, so a straggler shows in the line)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [float(b.item()) for b in bufs]


def shard_rows(n_rows: int, rank: int, world: int) -> range:
    """Contiguous batch slice of rank `rank` (prompts are independent: no data-path collective).
    Remainder rows go to the lowest ranks."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_ids(local_ids: torch.Tensor, n_rows_total: int) -> Optional[torch.Tensor]:
    """Gather per-rank id matrices [rows_r, T] to rank 0 (the only exchange after generation).
    Ragged row counts are padded to the maximum and trimmed."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local_ids
    world, rank = dist.get_world_size(), dist.get_rank()
    T = local_ids.shape[1]
    max_rows = -(-n_rows_total // world)
    pad = torch.full((max_rows, T), -1, dtype=local_ids.dtype, device=local_ids.device)
    pad[: local_ids.shape[0]] = local_ids
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = [bufs[r][: len(shard_rows(n_rows_total, r, world))] for r in range(world)]
    return torch.cat(out, 0) if rank == 0 else None
