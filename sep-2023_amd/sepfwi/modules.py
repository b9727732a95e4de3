"""Parameterisation modules: user parameters -> pad / mask -> (Lambda[MPa], Mu[MPa], Den) -> FWIFunction.

These are the pure-torch CALLERS of the operator boundary (reference: FWI_ops.py:66-330).  They work on CPU
tensors exactly like the reference and, unlike it, on HIP tensors too (device-resident iteration).  Same class
names, constructor signatures, attribute names (parameters `Vp`/`Vs`/`Den` ..., buffers `*_ref`, `Bounds`,
`Mask`) and forward(Shot_ids, ngpu) contract, so obj_wrapper.PyTorchObjective and the experiment scripts
work unchanged.  One generic base replaces the reference's copy-per-parameterisation.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import utils as ft
from .ops import FWIFunction, fwi_ops


class _MaskedTriple(nn.Module):
    """Three user fields, replicate-padded, blended with their initial values outside `Mask`."""

    NAMES = ("A", "B", "C")

    def __init__(self, a, b, c, Stf, opt, Mask=None, bounds=(None, None, None)):
        super().__init__()
        self.nz, self.nx = opt["nz"], opt["nx"]
        self.nz_orig, self.nx_orig = opt["nz_orig"], opt["nx_orig"]
        self.nPml, self.nPad = opt["nPml"], opt["nPad"]
        self.Bounds = {}
        padded = ft.padding(a, b, c, self.nz_orig, self.nx_orig, self.nz, self.nx, self.nPml, self.nPad)
        for name, t, tp, bd in zip(self.NAMES, (a, b, c), padded, bounds):
            self.register_buffer(name + "_ref", tp.clone().detach())
            if t.requires_grad:
                setattr(self, name, nn.Parameter(t))
                if bd is not None:
                    self.Bounds[name] = bd
            else:
                setattr(self, name, t)
        if Mask is None:
            Mask = torch.ones((self.nz + 2 * self.nPml + self.nPad, self.nx + 2 * self.nPml), dtype=torch.float32,
                              device=padded[0].device)
        # a buffer (the reference keeps a plain attribute, FWI_ops.py:98-103) so that module.to(device) moves it with
        # the parameters: with HIP tensors the whole chain pad -> mask -> Lame map -> propagator -> chain rule stays
        # in HBM; only SciPy's flat vector crosses PCIe
        self.register_buffer("Mask", Mask.to(padded[0].device))
        self.Stf = Stf
        self.para_fname = opt["para_fname"]

    def _masked(self):
        cur = [getattr(self, n) for n in self.NAMES]
        pad = ft.padding(*cur, self.nz_orig, self.nx_orig, self.nz, self.nx, self.nPml, self.nPad)
        return [self.Mask * p + (1.0 - self.Mask) * getattr(self, n + "_ref") for n, p in zip(self.NAMES, pad)]

    def lame(self, a, b, c):   # -> Lambda [MPa], Mu [MPa], Den
        raise NotImplementedError

    def forward(self, Shot_ids, ngpu=1):
        Lambda, Mu, Den = self.lame(*self._masked())
        return FWIFunction.apply(Lambda, Mu, Den, self.Stf, ngpu, Shot_ids, self.para_fname)


class FWI(_MaskedTriple):
    """Vp, Vs [m/s], Den [kg/m^3]   (FWI_ops.py:66-127)."""
    NAMES = ("Vp", "Vs", "Den")

    def __init__(self, Vp, Vs, Den, Stf, opt, Mask=None, Vp_bounds=None, Vs_bounds=None, Den_bounds=None):
        super().__init__(Vp, Vs, Den, Stf, opt, Mask, (Vp_bounds, Vs_bounds, Den_bounds))

    def lame(self, vp, vs, den):
        return (vp ** 2 - 2.0 * vs ** 2) * den / 1e6, vs ** 2 * den / 1e6, den   # FWI_ops.py:124-125


class FWI_Lame_Den(_MaskedTriple):
    """Lambda, Mu [MPa], Den   (FWI_ops.py:145-204)."""
    NAMES = ("Lam", "Mu", "Den")

    def __init__(self, Lam, Mu, Den, Stf, opt, Mask=None, Lam_bounds=None, Mu_bounds=None, Den_bounds=None):
        super().__init__(Lam, Mu, Den, Stf, opt, Mask, (Lam_bounds, Mu_bounds, Den_bounds))

    def lame(self, lam, mu, den):
        return lam, mu, den


class FWI_IP_IS_Den(_MaskedTriple):
    """P- and S-impedance [1e3 kg/m^2/s], Den   (FWI_ops.py:208-266)."""
    NAMES = ("IP", "IS", "Den")

    def __init__(self, IP, IS, Den, Stf, opt, Mask=None, IP_bounds=None, IS_bounds=None, Den_bounds=None):
        super().__init__(IP, IS, Den, Stf, opt, Mask, (IP_bounds, IS_bounds, Den_bounds))

    def lame(self, ip, is_, den):
        return (ip ** 2 - 2.0 * is_ ** 2) / den, is_ ** 2 / den, den   # FWI_ops.py:261-262


class FWI_obscalc(nn.Module):
    """Observed-data generation from PADDED Vp, Vs, Den   (FWI_ops.py:130-141)."""

    def __init__(self, Vp, Vs, Den, Stf, para_fname):
        super().__init__()
        self.Lambda = (Vp ** 2 - 2.0 * Vs ** 2) * Den / 1e6
        self.Mu = Vs ** 2 * Den / 1e6
        self.Den = Den
        self.Stf = Stf
        self.para_fname = para_fname

    def forward(self, Shot_ids, ngpu=1):
        from . import ops as _ops
        _ops.fwi_ops.obscalc(self.Lambda, self.Mu, self.Den, self.Stf, ngpu, Shot_ids, self.para_fname)
