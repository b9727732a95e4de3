"""Parameterisation modules: user parameters -> pad / mask -> (Lambda[MPa], Mu[MPa], Den) -> FWIFunction.

These are the CALLERS of the operator boundary (reference: FWI_ops.py:66-619).  They work on CPU tensors exactly like
the reference (pure torch) and, unlike it, on HIP tensors too (device-resident iteration): there the five algebraic
parameterisations and the two rock-physics maps run as ONE fused HIP launch forward and ONE backward
(csrc/param_maps.hip) instead of a dozen (rock physics: about forty) elementwise kernels over 3 x 9 MB.  Same class names, constructor signatures, attribute names (parameters `Vp`/`Vs`/`Den` ..., buffers
`*_ref`, `Bounds`, `Mask`) and forward(Shot_ids, ngpu) contract, so obj_wrapper.PyTorchObjective and the experiment
scripts work unchanged.  One generic base replaces the reference's copy-per-parameterisation.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _native
from . import utils as ft
from .ops import FWIFunction, fwi_ops

USE_FUSED_MAPS = True   # tests switch it off to compare the fused launches with the torch expressions


class _FusedParamMap(torch.autograd.Function):
    """(A, B, C) on the physical grid -> (Lambda, Mu, Den) on the padded grid through sepfwi_param_forward; backward
    through sepfwi_param_backward (padding transpose + mask + Lame derivatives in one launch)."""

    @staticmethod
    def forward(ctx, a, b, c, a_ref, b_ref, c_ref, mask, kind, nPml, nPad):
        L = _native.lib()
        a, b, c = [t.detach().contiguous() for t in (a, b, c)]
        nz, nx = a.shape
        outs = [torch.empty_like(a_ref) for _ in range(3)]
        st = torch.cuda.current_stream(a.device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        _native.check(L.sepfwi_param_forward(kind, nz, nx, nPml, nPad, p(a), p(b), p(c), p(a_ref), p(b_ref), p(c_ref), p(mask),
                                             p(outs[0]), p(outs[1]), p(outs[2]), C.c_void_p(st) if st else None))
        ctx.save_for_backward(a, b, c, a_ref, b_ref, c_ref, mask)
        ctx.meta = (kind, nPml, nPad)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gl, gm, gd):
        L = _native.lib()
        a, b, c, a_ref, b_ref, c_ref, mask = ctx.saved_tensors
        kind, nPml, nPad = ctx.meta
        nz, nx = a.shape
        zero = None
        gs = []
        for g in (gl, gm, gd):   # autograd hands None for an unused output
            if g is None:
                zero = torch.zeros_like(a_ref) if zero is None else zero
                g = zero
            gs.append(g.contiguous())
        outs = [torch.empty_like(a) for _ in range(3)]
        st = torch.cuda.current_stream(a.device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        _native.check(L.sepfwi_param_backward(kind, nz, nx, nPml, nPad, p(a), p(b), p(c), p(a_ref), p(b_ref), p(c_ref), p(mask),
                                              p(gs[0]), p(gs[1]), p(gs[2]), p(outs[0]), p(outs[1]), p(outs[2]),
                                              C.c_void_p(st) if st else None))
        return outs[0], outs[1], outs[2], None, None, None, None, None, None, None


class _MaskedTriple(nn.Module):
    """Three user fields, replicate-padded, blended with their initial values outside `Mask`."""

    NAMES = ("A", "B", "C")
    KIND = None   # ParamKind of csrc/param_maps.hpp when the Lame map has a fused HIP form

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

    def _fusable(self):
        """The one-launch maps take raw pointers and the sizes of the module: every tensor must have exactly the shape the
        kernels index (a Mask that torch would broadcast, or a parameter of another shape, stays on the torch expressions,
        which broadcast or raise)."""
        cur = [getattr(self, n) for n in self.NAMES]
        refs = [getattr(self, n + "_ref") for n in self.NAMES]
        padded_shape = (self.nz + 2 * self.nPml + self.nPad, self.nx + 2 * self.nPml)
        return (USE_FUSED_MAPS and self.KIND is not None and self.nz == self.nz_orig and self.nx == self.nx_orig
                and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.device == self.Mask.device
                        and tuple(t.shape) == (self.nz_orig, self.nx_orig) for t in cur)
                and all(tuple(t.shape) == padded_shape and t.dtype == torch.float32 and t.device == self.Mask.device for t in refs)
                and self.Mask.dtype == torch.float32 and tuple(self.Mask.shape) == padded_shape)

    def lame_padded(self):
        """-> Lambda [MPa], Mu [MPa], Den on the padded grid, differentiable w.r.t. the module's parameters."""
        if self._fusable():
            cur = [getattr(self, n) for n in self.NAMES]
            refs = [getattr(self, n + "_ref").contiguous() for n in self.NAMES]
            return _FusedParamMap.apply(cur[0], cur[1], cur[2], refs[0], refs[1], refs[2], self.Mask.contiguous(),
                                        int(self.KIND), int(self.nPml), int(self.nPad))
        return self.lame(*self._masked())

    def forward(self, Shot_ids, ngpu=1):
        Lambda, Mu, Den = self.lame_padded()
        return FWIFunction.apply(Lambda, Mu, Den, self.Stf, ngpu, Shot_ids, self.para_fname)


class FWI(_MaskedTriple):
    """Vp, Vs [m/s], Den [kg/m^3]   (FWI_ops.py:66-127)."""
    NAMES = ("Vp", "Vs", "Den")
    KIND = 0

    def __init__(self, Vp, Vs, Den, Stf, opt, Mask=None, Vp_bounds=None, Vs_bounds=None, Den_bounds=None):
        super().__init__(Vp, Vs, Den, Stf, opt, Mask, (Vp_bounds, Vs_bounds, Den_bounds))

    def lame(self, vp, vs, den):
        return (vp ** 2 - 2.0 * vs ** 2) * den / 1e6, vs ** 2 * den / 1e6, den   # FWI_ops.py:124-125


class FWI_Lame_Den(_MaskedTriple):
    """Lambda, Mu [MPa], Den   (FWI_ops.py:145-204)."""
    NAMES = ("Lam", "Mu", "Den")
    KIND = 1

    def __init__(self, Lam, Mu, Den, Stf, opt, Mask=None, Lam_bounds=None, Mu_bounds=None, Den_bounds=None):
        super().__init__(Lam, Mu, Den, Stf, opt, Mask, (Lam_bounds, Mu_bounds, Den_bounds))

    def lame(self, lam, mu, den):
        return lam, mu, den


class FWI_IP_IS_Den(_MaskedTriple):
    """P- and S-impedance [1e3 kg/m^2/s], Den   (FWI_ops.py:208-266)."""
    NAMES = ("IP", "IS", "Den")
    KIND = 2

    def __init__(self, IP, IS, Den, Stf, opt, Mask=None, IP_bounds=None, IS_bounds=None, Den_bounds=None):
        super().__init__(IP, IS, Den, Stf, opt, Mask, (IP_bounds, IS_bounds, Den_bounds))

    def lame(self, ip, is_, den):
        return (ip ** 2 - 2.0 * is_ ** 2) / den, is_ ** 2 / den, den   # FWI_ops.py:261-262


class FWI_Vp_Vs_IP(_MaskedTriple):
    """Vp, Vs [m/s] and P-impedance IP = rho Vp   (FWI_ops.py:270-330)."""
    NAMES = ("Vp", "Vs", "IP")
    KIND = 3

    def __init__(self, Vp, Vs, IP, Stf, opt, Mask=None, Vp_bounds=None, Vs_bounds=None, IP_bounds=None):
        super().__init__(Vp, Vs, IP, Stf, opt, Mask, (Vp_bounds, Vs_bounds, IP_bounds))

    def lame(self, vp, vs, ip):
        return ip * vp - 2. * ip / vp * vs ** 2, ip / vp * vs ** 2, ip / vp   # FWI_ops.py:326-328


class FWI_Vp_Vs_IS(_MaskedTriple):
    """Vp, Vs [m/s] and S-impedance IS = rho Vs   (FWI_ops.py:333-393)."""
    NAMES = ("Vp", "Vs", "IS")
    KIND = 4

    def __init__(self, Vp, Vs, IS, Stf, opt, Mask=None, Vp_bounds=None, Vs_bounds=None, IS_bounds=None):
        super().__init__(Vp, Vs, IS, Stf, opt, Mask, (Vp_bounds, Vs_bounds, IS_bounds))

    def lame(self, vp, vs, is_):
        return is_ / vs * vp ** 2 - 2.0 * is_ * vs, is_ * vs, is_ / vs   # FWI_ops.py:389-391


class FWI_Rock_Physics_VRH(_MaskedTriple):
    """Porosity, clay content, water saturation -> elastic moduli by the Voigt-Reuss-Hill average of a quartz / clay
    matrix with a water / hydrocarbon pore fill (FWI_ops.py:401-497; constants ft.ROCK)."""
    NAMES = ("PHI", "CC", "SW")
    KIND = 5

    def __init__(self, PHI, CC, SW, Stf, opt, Mask=None, PHI_bounds=None, CC_bounds=None, SW_bounds=None):
        super().__init__(PHI, CC, SW, Stf, opt, Mask, (PHI_bounds, CC_bounds, SW_bounds))

    def lame(self, phi, cc, sw):
        R = ft.ROCK
        kv = (1 - phi) * (R["k_c"] * cc + R["k_q"] * (1 - cc)) + phi * (R["k_w"] * sw + R["k_h"] * (1 - sw))       # :463
        kr_1 = (1 - phi) * (cc / R["k_c"] + (1 - cc) / R["k_q"]) + phi * (sw / R["k_w"] + (1 - sw) / R["k_h"])   # :464
        k = 0.5 * (kv + 1 / kr_1)
        mu = 0.5 * ((1 - phi) * (R["mu_c"] * cc + R["mu_q"] * (1 - cc)) + 0)      # Reuss shear modulus is zero (:468-472)
        rho_f = R["rho_w"] * sw + R["rho_h"] * (1 - sw)
        rho_s = R["rho_c"] * cc + R["rho_q"] * (1 - cc)
        den = rho_f * phi + rho_s * (1 - phi)
        lam = k - 2. / 3. * mu                                                     # :484
        return lam / 1e6, mu / 1e6, den


class FWI_Rock_Physics_gassmann(_MaskedTriple):
    """Porosity, clay content, water saturation -> elastic moduli by Biot-Gassmann fluid substitution on a consolidation-
    parameter frame (FWI_ops.py:504-619, after PyFWI; constants ft.ROCK, cs = 20)."""
    NAMES = ("PHI", "CC", "SW")
    KIND = 6

    def __init__(self, PHI, CC, SW, Stf, opt, Mask=None, PHI_bounds=None, CC_bounds=None, SW_bounds=None):
        super().__init__(PHI, CC, SW, Stf, opt, Mask, (PHI_bounds, CC_bounds, SW_bounds))

    def lame(self, phi, cc, sw):
        R = ft.ROCK
        cs = R["cs"]
        rho_f = R["rho_w"] * sw + R["rho_h"] * (1 - sw)
        k_f = R["k_w"] * sw + R["k_h"] * (1 - sw)
        k_s = R["k_c"] * cc + R["k_q"] * (1 - cc)
        mu_s = R["mu_c"] * cc + R["mu_q"] * (1 - cc)
        rho_s = R["rho_c"] * cc + R["rho_q"] * (1 - cc)
        k_d = k_s * ((1 - phi) / (1 + cs * phi))                                   # :591
        mu_d = mu_s * ((1 - phi) / (1 + 1.5 * cs * phi))
        delta = ((1 - phi) / phi) * (k_f / k_s) * (1 - (k_d / (k_s - k_s * phi)))  # :594
        denom = phi * (1 + delta)
        k_u = (phi * k_d + (1 - (1 + phi) * (k_d / k_s)) * k_f) / denom            # :598
        rho = rho_f * phi + rho_s * (1 - phi)
        vp = torch.sqrt((k_u + 0.75 * mu_d) / rho)                                 # :603 (0.75, as the reference has it)
        vs = torch.sqrt(mu_d / rho)
        return rho * (vp ** 2 - 2 * vs ** 2) / 1e6, rho * vs ** 2 / 1e6, rho        # :609-611


class FWI_obscalc(nn.Module):
    """Observed-data generation from PADDED Vp, Vs, Den   (FWI_ops.py:130-141)."""

    def __init__(self, Vp, Vs, Den, Stf, para_fname):
        super().__init__()
        self.Lambda = (Vp ** 2 - 2.0 * Vs ** 2) * Den / 1e6
        self.Mu = Vs ** 2 * Den / 1e6
        self.Den = Den
        self.Stf = Stf
        self.para_fname = para_fname

    def forward(self, Shot_ids, ngpu=1, to_store=False):
        """to_store=True: the gathers go into the sessions' HBM store of observed data instead of the Shot_*.bin files (extension)."""
        from . import ops as _ops
        if to_store:
            _ops.fwi_ops.obscalc(self.Lambda, self.Mu, self.Den, self.Stf, ngpu, Shot_ids, self.para_fname, to_store=True)
        else:
            _ops.fwi_ops.obscalc(self.Lambda, self.Mu, self.Den, self.Stf, ngpu, Shot_ids, self.para_fname)
