"""torch.distributed helpers used on the hot path (reference common/dist_utils.py:131-167).
One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node), "gloo" on CPU."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def all_gather(tensors):
    """Gather each tensor from every rank and concatenate along dim 0 (rank order).
    Like the reference (:131-149) the result carries NO autograd history: gradients do not flow
    back into the gathered features."""
    world = get_world_size()
    out = []
    for t in tensors:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.detach().contiguous())
        out.append(torch.cat(parts, dim=0))
    return out


def all_reduce(tensors, average=True):
    world = get_world_size()
    for t in tensors:
        dist.all_reduce(t)
        if average:
            t.mul_(1.0 / world)
    return tensors


def broadcast(obj):
    if isinstance(obj, torch.Tensor):
        dist.broadcast(obj, src=0)
        return obj
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([obj], dtype=torch.float64, device=dev)
    dist.broadcast(t, src=0)
    return t[0].item()
