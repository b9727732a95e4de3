"""Operator boundary: same surface as the reference's JIT-built extension `fwi_ops` and its autograd
Function (DAS_Waveform_Inversion/Ops/FWI/FWI_ops.py:15-63, Src/Torch_Fwi.cpp:12-142), backed by
libsepfwi.so through ctypes.  No CPU fallback exists.

Multi-GPU:
  * under torch.distributed (one process per GPU, launched by torchrun) every rank works on its
    contiguous block of Shot_ids and ONE all-reduce (RCCL on GPUs, gloo on CPU tests) sums the fused
    buffer [gLambda | gMu | gDen | misfit]  -- see dist.py;
  * without torch.distributed, `ngpu` > 1 drives that many devices from one process with one host
    thread each, the reference's own model (OpenMP, Src/Torch_Fwi.cpp:71-95).
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _native
from . import dist as _dist


def split_shots(group_size: int, ngpu: int):
    """Start offsets of the per-GPU shot blocks: int(float32 linspace(0, n, ngpu+1))
    (Src/Torch_Fwi.cpp:59-60,78-80)."""
    if ngpu > group_size:
        raise RuntimeError("The number of GPUs should be smaller than the number of shots!")  # Torch_Fwi.cpp:49-52
    return torch.linspace(0, group_size, ngpu + 1, dtype=torch.float32).to(torch.int32).tolist()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (the reference reads data_ptr<float>(), Torch_Fwi.cpp:55-58)" % name)
    return t.detach().contiguous()


class _FwiOps:
    """Module object: fwi_ops.forward / backward / obscalc."""

    def __init__(self):
        self.device_override = None   # bench/tests may pin the HIP device index

    # -- one cufd call on one device ------------------------------------------------------
    def _cufd(self, calc_id, gpu_id, Lambda, Mu, Den, Stf, shot_ids, para_fname, out_device=None):
        L = _native.lib()
        Lambda, Mu, Den, Stf = _f32c(Lambda, "Lambda"), _f32c(Mu, "Mu"), _f32c(Den, "Den"), _f32c(Stf, "Stf")
        if Lambda.dim() != 2 or Lambda.shape != Mu.shape or Lambda.shape != Den.shape:
            raise ValueError("Lambda, Mu, Den must be 2-D tensors of one shape (nz_pad, nx_pad)")
        ids = np.ascontiguousarray(np.asarray(shot_ids.cpu() if torch.is_tensor(shot_ids) else shot_ids, dtype=np.int32))
        dev = out_device if out_device is not None else Lambda.device
        misfit = torch.zeros(1, dtype=torch.float32)
        gL = gM = gD = gS = None
        if calc_id == 1:
            gL = torch.zeros(Lambda.shape, dtype=torch.float32, device=dev)
            gM = torch.zeros_like(gL)
            gD = torch.zeros_like(gL)
            gS = torch.zeros((int(ids.size), Stf.shape[1]), dtype=torch.float32)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        stream = None
        if Lambda.is_cuda:
            stream = C.c_void_p(torch.cuda.current_stream(Lambda.device).cuda_stream)
        rc = L.sepfwi_cufd_stream(ptr(misfit), ptr(gL), ptr(gM), ptr(gD), ptr(gS), ptr(Lambda), ptr(Mu), ptr(Den),
                                  ptr(Stf), int(calc_id), int(gpu_id), int(ids.size), C.c_void_p(ids.ctypes.data),
                                  str(para_fname).encode(), stream, 0)
        _native.check(rc)
        return misfit, gL, gM, gD, gS

    def _device_for(self, t: torch.Tensor, i: int) -> int:
        if self.device_override is not None:
            return int(self.device_override)
        if t.is_cuda:
            return t.device.index or 0
        if _dist.active():
            return _dist.local_device_index()
        return i

    # -- reference surface -------------------------------------------------------------------
    def backward(self, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname):
        """-> [misfit(1,), gLambda, gMu, gDen, gStf]   (fwi_backward, Src/Torch_Fwi.cpp:38-104)."""
        ids = torch.as_tensor(Shot_ids, dtype=torch.int32).cpu()
        n = int(ids.numel())
        if _dist.active():
            lo, hi = _dist.my_block(n)
            m, gL, gM, gD, gS_loc = self._cufd(1, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids[lo:hi], para_fname)
            m, gL, gM, gD = _dist.allreduce_gradients(m, gL, gM, gD)
            gS = torch.zeros_like(_f32c(Stf, "Stf").cpu())
            if _dist.rank() == 0:   # the reference returns GPU 0's buffer only (Torch_Fwi.cpp:102-103)
                gS[: gS_loc.shape[0]] = gS_loc
            return [m, gL, gM, gD, gS]
        ngpu = int(ngpu)
        bars = split_shots(n, ngpu)
        if ngpu == 1:
            parts = [self._cufd(1, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids, para_fname)]
        else:
            with ThreadPoolExecutor(max_workers=ngpu) as ex:   # one host thread per GPU, ctypes drops the GIL
                futs = [ex.submit(self._cufd, 1, self._device_for(Lambda, i), Lambda, Mu, Den, Stf,
                                  ids[bars[i]:bars[i + 1]], para_fname, Lambda.device) for i in range(ngpu)]
                parts = [f.result() for f in futs]
        m, gL, gM, gD, gS0 = parts[0]
        for p in parts[1:]:              # host sum of Torch_Fwi.cpp:96-101
            m = m + p[0]
            gL += p[1].to(gL.device)
            gM += p[2].to(gM.device)
            gD += p[3].to(gD.device)
        gS = torch.zeros_like(_f32c(Stf, "Stf").cpu())   # zeros_like(th_stf), rows by local shot position
        gS[: gS0.shape[0]] = gS0
        return [m, gL, gM, gD, gS]

    def forward(self, Lambda, Mu, Den, Stf, gpu_id, Shot_ids, para_fname):
        """-> [misfit(1,)]   (fwi_forward, Src/Torch_Fwi.cpp:12-36; calc_id 0 on device gpu_id)."""
        ids = torch.as_tensor(Shot_ids, dtype=torch.int32).cpu()
        m, *_ = self._cufd(0, int(gpu_id) if self.device_override is None else self.device_override,
                           Lambda, Mu, Den, Stf, ids, para_fname)
        return [m]

    def obscalc(self, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname):
        """Writes Shot_{pr,vx,vz,ett}{id}.bin; returns None   (fwi_obscalc, Src/Torch_Fwi.cpp:106-136)."""
        ids = torch.as_tensor(Shot_ids, dtype=torch.int32).cpu()
        n = int(ids.numel())
        if _dist.active():
            lo, hi = _dist.my_block(n)
            self._cufd(2, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids[lo:hi], para_fname)
            _dist.barrier()
            return None
        ngpu = int(ngpu)
        bars = split_shots(n, ngpu)
        if ngpu == 1:
            self._cufd(2, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids, para_fname)
        else:
            with ThreadPoolExecutor(max_workers=ngpu) as ex:
                futs = [ex.submit(self._cufd, 2, self._device_for(Lambda, i), Lambda, Mu, Den, Stf,
                                  ids[bars[i]:bars[i + 1]], para_fname) for i in range(ngpu)]
                [f.result() for f in futs]
        return None

    # -- extras --------------------------------------------------------------------------------
    def set_observed(self, para_fname, shot_id, ett, gpu_id=0):
        """Observed axial-strain gather of one shot from a tensor ((nrec, nSteps) float32, CPU or HIP) instead of
        Shot_ett{id}.bin: cached in HBM by the session of (para_fname, gpu_id) until release() / invalidation."""
        ett = _f32c(ett, "ett")
        if ett.dim() != 2:
            raise ValueError("ett must be (nrec, nSteps)")
        dev = self.device_override if self.device_override is not None else int(gpu_id)
        _native.check(_native.lib().sepfwi_set_observed(str(para_fname).encode(), dev, int(shot_id), C.c_void_p(ett.data_ptr()),
                                                        int(ett.shape[0]), int(ett.shape[1])))

    def stats(self, para_fname, gpu_id=0):
        st = _native.Stats()
        _native.check(_native.lib().sepfwi_get_stats(str(para_fname).encode(), int(gpu_id), C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def release(self):
        _native.lib().sepfwi_release_all()


fwi_ops = _FwiOps()


class FWIFunction(torch.autograd.Function):
    """forward() runs forward+adjoint and caches the gradients; backward() hands them out and ignores
    grad_misfit -- exactly the reference behaviour (FWI_ops.py:46-63)."""

    @staticmethod
    def forward(ctx, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname):
        from . import ops as _self   # late lookup so tests may swap `fwi_ops`
        outputs = _self.fwi_ops.backward(Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname)
        ctx.outputs = outputs[1:]
        return outputs[0]

    @staticmethod
    def backward(ctx, grad_misfit):
        grad_Lambda, grad_Mu, grad_Den, grad_stf = ctx.outputs
        return grad_Lambda, grad_Mu, grad_Den, grad_stf, None, None, None
