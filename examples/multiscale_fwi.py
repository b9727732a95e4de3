#!/usr/bin/env python
"""Multi-scale (frequency-continuation) FWI with the data-conditioning chain: the experiment-001 model inverted band by band with
the low-pass table the reference's scripts define but never use (Main-001-FWI-Anomaly-Vp-Vs-Den.py:46-51, `filter = [[0, 0, 2, 2.5],
[0, 0, 2, 3.5], ...]`; the band-pass call sites are commented out in its driver, Src/libCUFD.cu:370-374,446-448).  Here the
parameter key "filter" is live (csrc/conditioning.hip: zero-phase sin^2 / cos^2 band-pass on hipFFT applied to observed, synthetic
and residual gathers), so a stage is just another parameter file; every stage starts from the model the previous one ended with
and the last one uses the unfiltered data.

    python examples/multiscale_fwi.py --niter 4 [--bands 3]
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sep-2023_amd")]
from sepfwi import modules as M           # noqa: E402
from sepfwi import utils as ft            # noqa: E402
from sepfwi.obj_wrapper import PyTorchObjective, minimize_lbfgsb  # noqa: E402

FILTER_TABLE = [[0.0, 0.0, 2.0, 2.5], [0.0, 0.0, 2.0, 3.5], [0.0, 0.0, 2.0, 4.5], [0.0, 0.0, 2.0, 5.5], [0.0, 0.0, 2.0, 6.5],
                [0.0, 0.0, 2.0, 7.5]]    # Main-001-...py:46-51


def run(niter=4, n_bands=3, workdir=None, device="cuda", verbose=True):
    """-> [(band or None, misfit at the start of the stage, misfit at its end)], (vp, vs, rho) after the last stage."""
    dev = torch.device(device)
    nx, nz, dx, dz, dt, nt, f0, nPml = 201, 101, 20.0, 20.0, 0.002, 1501, 10.0, 32
    vp = np.ones((nz, nx), np.float32) * 4000.0
    vs = vp / 1.732
    rho = np.ones((nz, nx), np.float32) * 2500.0
    cur = [vp.copy(), vs.copy(), rho.copy()]
    vp[42:58, 42:58] += 80.0
    vs[42:58, 92:108] -= 80.0 / 1.732
    rho[42:58, 142:158] += 40.0
    nPad = ft.nPad_for(nz, nPml)
    nz_pad, nx_pad = nz + 2 * nPml + nPad, nx + 2 * nPml
    Mask = np.zeros((nz_pad, nx_pad), np.float32)
    Mask[nPml + 4:nPml + nz, nPml:nPml + nx] = 1.0
    src_x = np.arange(10, nx - 10, 10).astype(int)
    rec_x = np.arange(10, nx - 10).astype(int)
    work = workdir or os.path.join(tempfile.gettempdir(), "sepfwi_example_multiscale")
    os.makedirs(work, exist_ok=True)
    survey_fname, data_dir = os.path.join(work, "survey_file.json"), os.path.join(work, "Data")
    ft.surveyGen(np.ones_like(src_x), src_x, 95 * np.ones_like(rec_x), rec_x, survey_fname)
    Stf = torch.tensor(ft.sourceGene(f0, nt, dt), dtype=torch.float32).repeat(len(src_x), 1).to(dev)
    Shot_ids = torch.arange(len(src_x), dtype=torch.int32)

    def para_for(tag, band):
        fn = os.path.join(work, "para_%s.json" % tag)
        ft.paraGen(nz_pad, nx_pad, dz, dx, nt, dt, f0, nPml, nPad, fn, survey_fname, data_dir, filter_para=band)
        return fn

    pad = lambda m: torch.tensor(ft.padding_numpy_array(m, nPml, nPad), dtype=torch.float32, device=dev)
    M.FWI_obscalc(pad(vp), pad(vs), pad(rho), Stf, para_for("obs", None))(Shot_ids, ngpu=1)   # raw gathers, written once

    log = []
    for k, band in enumerate(FILTER_TABLE[len(FILTER_TABLE) - n_bands:] + [None]):    # the widest n_bands rows: a 10 Hz Ricker has little below 4 Hz
        opt = dict(nz=nz, nx=nx, nz_orig=nz, nx_orig=nx, nPml=nPml, nPad=nPad, para_fname=para_for("stage%d" % k, band))
        T = lambda m: torch.tensor(m, dtype=torch.float32, device=dev, requires_grad=True)
        fwi = M.FWI(T(cur[0]), T(cur[1]), T(cur[2]), Stf, opt, Mask=torch.tensor(Mask, device=dev))
        obj = PyTorchObjective(fwi, lambda: fwi(Shot_ids, ngpu=1))
        fun, jac = obj.fun, obj.jac
        f_start = fun(obj.x0)
        res = minimize_lbfgsb(fun, obj.x0, jac, bounds=obj.bounds, maxiter=niter, maxcor=5, ftol=1e-12, gtol=1e-16, maxfun=1500, maxls=6)
        n = nz * nx
        cur = [res.x[i * n:(i + 1) * n].reshape(nz, nx).astype(np.float32) for i in range(3)]
        log.append((band, float(f_start), float(res.fun)))
        if verbose:
            print("stage %d, %s: misfit %.4e -> %.4e in %d iterations (%d evaluations)" %
                  (k, "low-pass %g-%g Hz" % (band[2], band[3]) if band else "all frequencies", f_start, res.fun, res.nit, res.nfev), flush=True)
    if verbose:
        print("recovered Vp anomaly peak %.1f m/s (true 80), Vs %.1f (true %.1f), density %.1f kg/m^3 (true 40)" %
              ((cur[0] - 4000.0)[42:58, 42:58].max(), (cur[1] - 4000.0 / 1.732)[42:58, 92:108].min(), -80.0 / 1.732,
               (cur[2] - 2500.0)[42:58, 142:158].max()))
    return log, cur


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--niter", type=int, default=4, help="L-BFGS-B iterations per stage")
    ap.add_argument("--bands", type=int, default=3, help="how many rows of the filter table, counted from its wide end (then one stage on all frequencies)")
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()
    run(a.niter, a.bands, a.workdir)
