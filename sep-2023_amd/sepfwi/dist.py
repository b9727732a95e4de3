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


def fused_view(misfit, gL, gM, gD):
    """The 1-D tensor [gLambda | gMu | gDen | misfit] when the four already ARE one contiguous buffer in that order
    (ops._cufd allocates them that way and the session writes gradients and misfit straight into it), else None."""
    n = gL.numel()
    try:
        base = gL.untyped_storage().data_ptr()
        same = all(t.untyped_storage().data_ptr() == base and t.device == gL.device and t.dtype == torch.float32 and t.is_contiguous()
                   for t in (gM, gD, misfit))
    except RuntimeError:
        return None
    o = gL.storage_offset()
    if not (same and gL.is_contiguous() and gM.numel() == n and gD.numel() == n and misfit.numel() == 1 and
            gM.storage_offset() == o + n and gD.storage_offset() == o + 2 * n and misfit.storage_offset() == o + 3 * n):
        return None
    return torch.as_strided(gL, (3 * n + 1,), (1,), o)


# Record of the collectives issued by allreduce_gradients since the last reset: what a driver needs to verify that the
# backend really saw N ranks and what the one collective per operator call cost (bench.py prints it as "rccl": {...}).
_coll = {"calls": 0, "bytes": 0, "host_ms": 0.0, "events": [], "staged": 0, "timing": False}


def enable_collective_timing(on=True):
    """Timing of the collective is opt-in (bench.py, tests): an inversion loop that never reads the record should not create two HIP
    events per operator call.  Calls, bytes and staged copies are always counted."""
    _coll["timing"] = bool(on)


def _timed_all_reduce(buf):
    """all_reduce(SUM) of one fused buffer, timed: a HIP-event pair on the current stream around the call for device buffers
    (the stream waits for the collective, so the pair brackets it; resolved later, no synchronisation here), wall time for
    host buffers (gloo is synchronous)."""
    import time
    _coll["calls"] += 1
    _coll["bytes"] = int(buf.numel() * buf.element_size())
    if not _coll["timing"]:
        td.all_reduce(buf, op=td.ReduceOp.SUM)
        return
    if buf.is_cuda:
        with torch.cuda.device(buf.device):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            td.all_reduce(buf, op=td.ReduceOp.SUM)
            e1.record()
        if len(_coll["events"]) < 4096:
            _coll["events"].append((e0, e1))
    else:
        t0 = time.perf_counter()
        td.all_reduce(buf, op=td.ReduceOp.SUM)
        _coll["host_ms"] += 1e3 * (time.perf_counter() - t0)


def collective_stats(reset=False):
    """{"ranks", "backend", "calls", "bytes", "allreduce_ms", "staged"}: calls and mean time of the gradient all-reduce since the last
    reset, bytes of one call's buffer (4 * (3 nz nx + 1)), how many calls had to stage a copy.  Synchronises the device the
    events were recorded on."""
    ms, n_ev = _coll["host_ms"], 0
    for e0, e1 in _coll["events"]:
        e1.synchronize()
        ms += e0.elapsed_time(e1)
        n_ev += 1
    timed = n_ev if n_ev else (_coll["calls"] if _coll["timing"] else 0)
    out = {"ranks": world_size(), "backend": (td.get_backend() if (td.is_available() and td.is_initialized()) else None),
           "calls": _coll["calls"], "bytes": _coll["bytes"], "allreduce_ms": (ms / timed if timed else None), "staged": _coll["staged"]}
    if reset:
        _coll.update(calls=0, bytes=0, host_ms=0.0, events=[], staged=0)
    return out


def allreduce_gradients(misfit, gL, gM, gD):
    """Sum [gLambda | gMu | gDen | misfit] over ranks with ONE collective, in place.  On the production path (RCCL, model in
    HBM) the four tensors are views of the one buffer the session wrote, which is handed to all_reduce as it is: no staging
    copy, the misfit never leaves the device.  Other combinations (gloo with HIP tensors: ranks sharing a GPU in rehearsals;
    RCCL with the reference's CPU tensors; tensors that are not one buffer) stage ONE fused copy each way."""
    n = gL.numel()
    backend = td.get_backend()
    want = "cuda" if backend == "nccl" else "cpu"     # RCCL reduces device buffers, gloo host buffers
    fused = fused_view(misfit, gL, gM, gD)
    if fused is not None and fused.device.type == want:
        _timed_all_reduce(fused)
        return misfit, gL, gM, gD
    _coll["staged"] += 1
    use_dev = torch.device("cuda", local_device_index()) if want == "cuda" else torch.device("cpu")
    if want == "cuda" and gL.is_cuda:
        use_dev = gL.device
    if fused is not None:
        stage = fused.to(use_dev)
        _timed_all_reduce(stage)
        fused.copy_(stage)
        return misfit, gL, gM, gD
    stage = torch.cat([gL.reshape(-1).to(use_dev), gM.reshape(-1).to(use_dev), gD.reshape(-1).to(use_dev), misfit.reshape(-1)[:1].to(use_dev)])
    _timed_all_reduce(stage)
    gL.copy_(stage[0:n].view_as(gL))
    gM.copy_(stage[n:2 * n].view_as(gM))
    gD.copy_(stage[2 * n:3 * n].view_as(gD))
    misfit = stage[3 * n:3 * n + 1].to(misfit.device).clone()
    return misfit, gL, gM, gD


def barrier():
    if active():
        td.barrier()
