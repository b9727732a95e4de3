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


_DIMS_CACHE = {}


def _para_dims(para_fname):
    """(nz, nx, nSteps) of the one-line parameter JSON (fwi_utils.py:46-83), cached by mtime.  The reference passes raw
    data_ptr<float>()s with no size anywhere (Src/Torch_Fwi.cpp:55-58); here a tensor of the wrong shape is an error
    before the library reads or writes past it."""
    import json
    import os
    try:
        key = (str(para_fname), os.stat(para_fname).st_mtime_ns)
    except OSError:
        return None          # the library reports the missing file (SEPFWI_EIO)
    if key not in _DIMS_CACHE:
        try:
            with open(para_fname) as fp:
                j = json.loads(fp.readline())
            _DIMS_CACHE.clear()
            _DIMS_CACHE[key] = (int(j["nz"]), int(j["nx"]), int(j["nSteps"]))
        except (ValueError, KeyError, TypeError):
            return None      # the library reports the malformed file (SEPFWI_EJSON)
    return _DIMS_CACHE[key]


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
        if Mu.device != Lambda.device or Den.device != Lambda.device:
            raise ValueError("Lambda, Mu, Den must live on one device")
        ids = np.ascontiguousarray(np.asarray(shot_ids.cpu() if torch.is_tensor(shot_ids) else shot_ids, dtype=np.int32))
        dims = _para_dims(para_fname)
        if dims is not None:
            nz, nx, nSteps = dims
            if tuple(Lambda.shape) != (nz, nx):
                raise ValueError("Lambda/Mu/Den are %s but the parameter file says (nz, nx) = (%d, %d)" % (tuple(Lambda.shape), nz, nx))
            if Stf.dim() != 2 or Stf.shape[1] != nSteps:
                raise ValueError("Stf must be (nSrc, nSteps = %d), got %s" % (nSteps, tuple(Stf.shape)))
            if ids.size and (int(ids.min()) < 0 or int(ids.max()) >= Stf.shape[0]):
                raise ValueError("Shot_ids must index rows of Stf (0..%d), got %d..%d" % (Stf.shape[0] - 1, int(ids.min()), int(ids.max())))
        gpu_id = int(gpu_id)
        dev = out_device if out_device is not None else Lambda.device
        # gradients are allocated where the session writes them in place: on ITS GPU when the model lives on a GPU
        # (also another one: single-process ngpu > 1), on the host for the reference's CPU tensors
        gdev = torch.device("cuda", gpu_id) if Lambda.is_cuda else torch.device("cpu")
        # the loss lives where the model lives (the reference: CPU tensors throughout, torch::zeros(1)); calc_id 1 below makes it the
        # last element of the gradient buffer
        misfit = torch.zeros(1, dtype=torch.float32, device=gdev if calc_id == 0 else "cpu")
        gL = gM = gD = gS = None
        if calc_id == 1:
            # ONE buffer [gLambda | gMu | gDen | misfit]: the session writes all four in place, and under torch.distributed
            # this very buffer is what the single all-reduce sums (dist.allreduce_gradients) -- no staging, misfit stays in HBM
            n = Lambda.numel()
            fused = torch.zeros(3 * n + 1, dtype=torch.float32, device=gdev)
            gL, gM, gD = (fused[k * n:(k + 1) * n].view(Lambda.shape) for k in range(3))
            misfit = fused[3 * n:3 * n + 1]
            gS = torch.zeros((int(ids.size), Stf.shape[1]), dtype=torch.float32)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        stream = None        # NULL = the legacy default stream: the library orders itself behind it
        if Lambda.is_cuda and Lambda.device.index == gpu_id:
            cs = torch.cuda.current_stream(Lambda.device).cuda_stream
            stream = C.c_void_p(cs) if cs else None
        elif Lambda.is_cuda:
            torch.cuda.synchronize(Lambda.device)   # the model was produced on another GPU: finished before it is staged
        if Stf.is_cuda:
            torch.cuda.synchronize(Stf.device)      # read with a blocking copy inside the library
        rc = L.sepfwi_cufd_stream(ptr(misfit), ptr(gL), ptr(gM), ptr(gD), ptr(gS), ptr(Lambda), ptr(Mu), ptr(Den),
                                  ptr(Stf), int(calc_id), gpu_id, int(ids.size), C.c_void_p(ids.ctypes.data),
                                  str(para_fname).encode(), stream, 0)
        _native.check(rc)
        if calc_id == 1 and gL.device != dev:   # single-process ngpu > 1: every block's results return to the model's device
            gL, gM, gD, misfit = gL.to(dev), gM.to(dev), gD.to(dev), misfit.to(dev)
        elif calc_id == 0 and misfit.device != dev:   # forward(): the loss follows the model too, whatever gpu_id computed it
            misfit = misfit.to(dev)
        return misfit, gL, gM, gD, gS

    def _device_for(self, t: torch.Tensor, i: int, ngpu: int = 1) -> int:
        """HIP device of shot block `i` of `ngpu`: the pinned one (bench, tests); the tensors' own device for a single block or a
        rank of a torch.distributed job; LOCAL_RANK for a rank that holds the reference's CPU tensors; device i otherwise (the
        reference's omp thread i <-> GPU i, Src/Torch_Fwi.cpp:71-95)."""
        if self.device_override is not None:
            return int(self.device_override)
        if t.is_cuda and (ngpu == 1 or _dist.active()):
            return t.device.index or 0      # the tensors' own device, also under per-rank device masking (every rank sees one GPU)
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
                futs = [ex.submit(self._cufd, 1, self._device_for(Lambda, i, ngpu), Lambda, Mu, Den, Stf,
                                  ids[bars[i]:bars[i + 1]], para_fname, Lambda.device) for i in range(ngpu)]
                parts = [f.result() for f in futs]
        m, gL, gM, gD, gS0 = parts[0]
        m = m.clone()                    # not a view of the fused [gL | gM | gD | misfit] buffer: a kept loss must not pin 3 nz nx floats
        for p in parts[1:]:              # sum of Torch_Fwi.cpp:96-101 (every part already sits on Lambda's device)
            m = m + p[0]
            gL += p[1]
            gM += p[2]
            gD += p[3]
        gS = torch.zeros_like(_f32c(Stf, "Stf").cpu())   # zeros_like(th_stf), rows by local shot position
        gS[: gS0.shape[0]] = gS0
        return [m, gL, gM, gD, gS]

    def forward(self, Lambda, Mu, Den, Stf, gpu_id, Shot_ids, para_fname):
        """-> [misfit(1,)]   (fwi_forward, Src/Torch_Fwi.cpp:12-36; calc_id 0 on device gpu_id)."""
        ids = torch.as_tensor(Shot_ids, dtype=torch.int32).cpu()
        m, *_ = self._cufd(0, int(gpu_id) if self.device_override is None else self.device_override,
                           Lambda, Mu, Den, Stf, ids, para_fname)
        return [m]

    def obscalc(self, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname, to_store=False):
        """Writes Shot_{pr,vx,vz,ett}{id}.bin; returns None   (fwi_obscalc, Src/Torch_Fwi.cpp:106-136).
        Extension `to_store=True` (SEPFWI_CALC_OBSERVE_TO_STORE): no files -- the axial-strain gather of every shot goes straight
        into the HBM store of observed data of the session that will later evaluate that shot (same device, same shot split as
        `backward` with the same ngpu / ranks), bit for bit what the file route leaves there."""
        calc = 3 if to_store else 2
        ids = torch.as_tensor(Shot_ids, dtype=torch.int32).cpu()
        n = int(ids.numel())
        if _dist.active():
            lo, hi = _dist.my_block(n)
            self._cufd(calc, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids[lo:hi], para_fname)
            _dist.barrier()
            return None
        ngpu = int(ngpu)
        bars = split_shots(n, ngpu)
        if ngpu == 1:
            self._cufd(calc, self._device_for(Lambda, 0), Lambda, Mu, Den, Stf, ids, para_fname)
        else:
            with ThreadPoolExecutor(max_workers=ngpu) as ex:
                futs = [ex.submit(self._cufd, calc, self._device_for(Lambda, i, ngpu), Lambda, Mu, Den, Stf,
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
        if ett.is_cuda:     # the library copies on its own stream, ordered only behind the legacy default stream
            torch.cuda.current_stream(ett.device).synchronize()
        dev = self.device_override if self.device_override is not None else int(gpu_id)
        _native.check(_native.lib().sepfwi_set_observed(str(para_fname).encode(), dev, int(shot_id), C.c_void_p(ett.data_ptr()),
                                                        int(ett.shape[0]), int(ett.shape[1])))

    def debug_field(self, para_fname, which, lane=0, gpu_id=0):
        """Test hook (sepfwi_debug_field): wavefield 0..4 (vz, vx, szz, sxx, sxz) / adjoint 5..9 of a forward lane as the
        last call left it, (nz - nPad, nx) float32."""
        import json
        with open(para_fname) as fp:
            j = json.loads(fp.readline())
        out = torch.empty((int(j["nz"]) - int(j["nPad"]), int(j["nx"])), dtype=torch.float32)
        dev = self.device_override if self.device_override is not None else int(gpu_id)
        _native.check(_native.lib().sepfwi_debug_field(str(para_fname).encode(), dev, int(lane), int(which), C.c_void_p(out.data_ptr())))
        return out

    def stats(self, para_fname, gpu_id=0):
        st = _native.Stats()
        _native.check(_native.lib().sepfwi_get_stats(str(para_fname).encode(), int(gpu_id), C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def loop_status(self, para_fname, gpu_id=0):
        """"" while the session's backward passes run in the persistent loop, else why they do not (sepfwi_loop_status)."""
        buf = C.create_string_buffer(512)
        _native.check(_native.lib().sepfwi_loop_status(str(para_fname).encode(), int(gpu_id), buf, 512))
        return buf.value.decode(errors="replace")

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
