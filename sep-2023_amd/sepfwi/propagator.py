"""The reference's second, non-autograd caller of the operator (DAS_Waveform_Inversion/Ops/FWI/propagator.py:8-226 and
survey.py:3-38): `Model`, `Survey`, `Propagator`, `ElasticPropagator.apply_forward / apply_gradient`.

Same class and attribute names and the same numbers out: `apply_forward` models every shot of the survey and writes the
Shot_{pr,vx,vz,ett}{id}.bin gathers (propagator.py:80-137); `apply_gradient` returns (misfit, grad_vp, grad_vs, grad_rho,
grad_stf) of a trial model against those gathers, with the chain rule from (lambda, mu, rho) to (vp, vs, rho) written out
by hand (propagator.py:202-216).  The moduli handed to the operator are rho*(vp^2 - 2 vs^2) and rho*vs^2 WITHOUT the 1e-6
of FWI_ops.py:124-125 (propagator.py:103-104): the operator works in MPa, so this caller's velocities are km/s with rho in
kg/m^3 -- kept as the reference has it.

What is different: `device=` (a HIP device) keeps the padded moduli, the gradients and the chain rule in HBM; the default
(None) hands CPU tensors over as the reference does.  The two routes differ in the last step only: the default multiplies the
operator's float32 gradients by the caller's numpy model arrays (float64 unless the caller made them float32), as the reference
does, and returns that dtype; with `device=` the chain rule runs in float32 on the GPU and float32 arrays come back -- the same
numbers to about 1e-7 relative."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from . import utils as ft


class Model:
    """survey.py:3-22: sizes in cells, spacings, time axis, C-PML width, the three (nz, nx) parameter arrays, experiment
    directory."""

    def __init__(self, nx, nz, dx, dz, nt, dt, nPml, vp, vs, rho, exp_name):
        self.nx, self.nz, self.dx, self.dz, self.nt, self.dt, self.nPml = nx, nz, dx, dz, nt, dt, nPml
        self.vp, self.vs, self.rho = vp, vs, rho
        self.exp_name = exp_name


class Survey:
    """survey.py:25-38: Ricker peak frequency, source and receiver positions as grid INDICES of the physical model (the
    operator's survey file wants indices of the padded grid: this class's users add nPml themselves, as the reference's do)."""

    def __init__(self, f0, src_x, src_z, rec_x, rec_z):
        self.f0, self.src_x, self.src_z, self.rec_x, self.rec_z = f0, src_x, src_z, rec_x, rec_z


class Propagator:
    """propagator.py:8-54: the interface."""

    def __init__(self, model, survey):
        self.model, self.survey = model, survey

    def apply_forward(self, data):
        raise NotImplementedError

    def apply_adjoint(self, data):
        raise NotImplementedError

    def apply_gradient(self, gradient):
        raise NotImplementedError


class ElasticPropagator(Propagator):
    """propagator.py:57-226: isotropic elastic velocity-stress propagator behind `fwi_ops.obscalc` / `fwi_ops.backward`."""

    def __init__(self, model, survey, device=None):
        super().__init__(model, survey)
        self.device = None if device is None else torch.device(device)

    # what both entry points of the reference do before they call the operator (propagator.py:84-135 == 151-200)
    def _operands(self, medium):
        m, s = self.model, self.survey
        nPad = ft.nPad_for(m.nz, m.nPml)
        nz_pad, nx_pad = m.nz + 2 * m.nPml + nPad, m.nx + 2 * m.nPml
        vp, vs, rho = (np.asarray(a) for a in (medium.vp, medium.vs, medium.rho))
        lam = rho * (vp ** 2 - 2 * vs ** 2)
        mu = rho * vs ** 2
        n_src = len(s.src_x)
        stf = torch.tensor(ft.sourceGene(s.f0, m.nt, m.dt), dtype=torch.float32).repeat(n_src, 1)
        shot_ids = torch.arange(n_src, dtype=torch.int32)
        padded = [torch.tensor(ft.padding_numpy_array(a, m.nPml, nPad), dtype=torch.float32) for a in (lam, mu, rho)]
        if self.device is not None:
            padded = [t.to(self.device) for t in padded]
        para_fname = os.path.join(m.exp_name, "para_file.json")
        survey_fname = os.path.join(m.exp_name, "survey_file.json")
        ft.paraGen(nz_pad, nx_pad, m.dz, m.dx, m.nt, m.dt, s.f0, m.nPml, nPad, para_fname, survey_fname,
                   os.path.join(m.exp_name, "Data"))
        ft.surveyGen(np.asarray(s.src_z), np.asarray(s.src_x), np.asarray(s.rec_z), np.asarray(s.rec_x), survey_fname)
        return padded, stf, shot_ids, para_fname, nPad

    def apply_forward(self, ngpu=1):
        """Model the survey in `self.model` and leave the gathers in <exp_name>/Data (propagator.py:80-137)."""
        (lam, mu, rho), stf, shot_ids, para_fname, _ = self._operands(self.model)
        ops.fwi_ops.obscalc(lam, mu, rho, stf, ngpu, shot_ids, para_fname)

    def apply_gradient(self, model_init, ngpu=1):
        """Misfit of `model_init` against the gathers on disk and its gradient in (vp, vs, rho) on the physical grid
        (propagator.py:141-216).  -> misfit (1,), grad_vp, grad_vs, grad_rho (nz, nx), grad_stf (n_src, nt), numpy."""
        (lam, mu, rho), stf, shot_ids, para_fname, nPad = self._operands(model_init)
        misfit, g_lam, g_mu, g_rho0, g_stf = ops.fwi_ops.backward(lam, mu, rho, stf, ngpu, shot_ids, para_fname)
        nPml = self.model.nPml
        crop = (slice(nPml, -(nPad + nPml)), slice(nPml, -nPml))
        if self.device is not None:      # chain rule where the gradients are, one read-back of three physical-size arrays
            vp, vs, r = (torch.as_tensor(np.asarray(a), dtype=torch.float32).to(self.device)
                         for a in (model_init.vp, model_init.vs, model_init.rho))
            gl, gm, gr = g_lam[crop], g_mu[crop], g_rho0[crop]
            grads = (2 * r * vp * gl, -4 * r * vs * gl + 2 * r * vs * gm, (vp ** 2 - 2 * vs ** 2) * gl + vs ** 2 * gm + gr)
            grad_vp, grad_vs, grad_rho = (g.cpu().numpy() for g in grads)
        else:
            vp, vs, r = (np.asarray(a) for a in (model_init.vp, model_init.vs, model_init.rho))
            gl, gm, gr = (g.detach().cpu().numpy()[crop] for g in (g_lam, g_mu, g_rho0))
            grad_vp = 2 * r * vp * gl
            grad_vs = -4 * r * vs * gl + 2 * r * vs * gm
            grad_rho = (vp ** 2 - 2 * vs ** 2) * gl + vs ** 2 * gm + gr
        return misfit.detach().cpu().numpy(), grad_vp, grad_vs, grad_rho, g_stf.detach().cpu().numpy()
