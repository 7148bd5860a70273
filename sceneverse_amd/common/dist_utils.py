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


def bf16_wire_fp32_acc_hook(state, bucket):
    """DDP communication hook: the mean of a gradient bucket with bf16 on the wire and fp32 accumulation.

    The reference all-reduces fp32 buckets (trainer/build.py:66-75; C1, 491 MB per step).  torch's
    `bf16_compress_hook` halves the bytes but all-reduces IN bf16, so the cross-rank sum itself is rounded at every
    ring step.  Here each rank rounds its own contribution to bf16 once, slice r of every rank's bucket goes to rank
    r (all-to-all: on MI355X the 7 xGMI links of a GPU are point-to-point to its 7 peers, which is exactly this
    pattern), rank r sums its world_size slices in fp32, divides by world_size, rounds the mean to bf16 once and the
    slices are all-gathered.  Bytes on the wire = those of a bf16 all-reduce; error = two bf16 roundings per element
    independent of world size.  `state` = process group or None (default group)."""
    group = state if state is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    n = buf.numel()
    if world == 1:
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut
    chunk = (n + world - 1) // world
    send = torch.empty(world * chunk, dtype=torch.bfloat16, device=buf.device)
    send[:n].copy_(buf)
    if world * chunk > n:
        send[n:].zero_()
    recv = torch.empty_like(send)
    mean16 = torch.empty(chunk, dtype=torch.bfloat16, device=buf.device)
    out = torch.empty_like(send)

    if not buf.is_cuda:
        # gloo (CPU tests): done-callbacks would run on gloo's worker threads in COMPLETION order, which may differ
        # between ranks and would interleave the collectives of different buckets differently -> issue both
        # collectives synchronously, in the (rank-consistent) order DDP hands out the buckets
        dist.all_to_all_single(recv, send, group=group)
        mean16.copy_(recv.view(world, chunk).sum(dim=0, dtype=torch.float32).mul_(1.0 / world))
        dist.all_gather_into_tensor(out, mean16, group=group)
        buf.copy_(out[:n])
        done = torch.futures.Future()
        done.set_result(buf)
        return done
    # RCCL: everything is issued in program order (same on every rank) on a SIDE stream, so the backward pass that is
    # still running on the current stream is not held up; `work.wait()` only orders the side stream after the
    # collective's stream.  The returned future carries the side stream's event (set_result records it), which is
    # what DDP synchronises with before the optimizer reads the bucket.
    cur = torch.cuda.current_stream(buf.device)
    side = _hook_stream(buf.device)
    side.wait_stream(cur)
    send.record_stream(side), recv.record_stream(side), mean16.record_stream(side), out.record_stream(side)
    with torch.cuda.stream(side):
        dist.all_to_all_single(recv, send, group=group, async_op=True).wait()
        mean16.copy_(recv.view(world, chunk).sum(dim=0, dtype=torch.float32).mul_(1.0 / world))
        dist.all_gather_into_tensor(out, mean16, group=group, async_op=True).wait()
        buf.copy_(out[:n])
        done = torch.futures.Future(devices=[buf.device])
        done.set_result(buf)
    return done


_HOOK_STREAMS = {}


def _hook_stream(device) -> "torch.cuda.Stream":
    st = _HOOK_STREAMS.get(device)
    if st is None:
        st = _HOOK_STREAMS[device] = torch.cuda.Stream(device=device)
    return st
