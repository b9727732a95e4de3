"""One process per GPU: contiguous shot blocks per rank and a single all-reduce of the fused gradient
buffer per operator call (north star: "a single RCCL all-reduce of the gradient over xGMI per FWI
iteration").  The reference instead sums per-GPU host tensors serially (Src/Torch_Fwi.cpp:96-101).

backend "nccl" is RCCL on ROCm; CPU tests use "gloo" with world_size 2.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def active() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank() -> int:
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def world_size() -> int:
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def local_device_index() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def block_bounds(n_shots: int, world: int):
    """Start offsets of each rank's contiguous block -- the reference's split rule
    (Src/Torch_Fwi.cpp:59-60,78-80) with ranks in place of GPUs of one process."""
    if world > n_shots:
        raise RuntimeError("The number of GPUs should be smaller than the number of shots!")
    return torch.linspace(0, n_shots, world + 1, dtype=torch.float32).to(torch.int32).tolist()


def my_block(n_shots: int):
    b = block_bounds(n_shots, world_size())
    return b[rank()], b[rank() + 1]


def allreduce_gradients(misfit, gL, gM, gD):
    """Sum [gLambda | gMu | gDen | misfit] over ranks with ONE collective.  Tensors may live on the
    host (gloo) or on the rank's GPU (RCCL); results come back in place, same shapes."""
    n = gL.numel()
    dev = gL.device
    backend = td.get_backend()
    if backend == "nccl":     # RCCL reduces device buffers
        use_dev = dev if dev.type == "cuda" else torch.device("cuda", local_device_index())
    else:                     # gloo (CPU tests, or several ranks sharing one GPU): reduce on the host
        use_dev = torch.device("cpu")
    fused = torch.empty(3 * n + 1, dtype=torch.float32, device=use_dev)
    fused[0:n] = gL.reshape(-1).to(use_dev)
    fused[n:2 * n] = gM.reshape(-1).to(use_dev)
    fused[2 * n:3 * n] = gD.reshape(-1).to(use_dev)
    fused[3 * n] = misfit.reshape(-1)[0].to(use_dev)
    td.all_reduce(fused, op=td.ReduceOp.SUM)
    gL.copy_(fused[0:n].view_as(gL))
    gM.copy_(fused[n:2 * n].view_as(gM))
    gD.copy_(fused[2 * n:3 * n].view_as(gD))
    misfit = fused[3 * n:3 * n + 1].to(misfit.device).clone()
    return misfit, gL, gM, gD


def barrier():
    if active():
        td.barrier()
