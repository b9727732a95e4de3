"""The three fully specified inversion experiments of the reference whose printed L-BFGS-B logs are the
only machine-checkable numbers of the repository:

    001  Vp/Vs/Den      notebooks/001-FWI-Anomaly-Vp-Vs-Den.ipynb cells 3,7 + Main-001-...py
    002  Lambda/Mu/Den  notebooks/002-FWI-Anomaly-Lame-Den.ipynb  cells 3,7 + Main-002-...py
    003  IP/IS/Den      notebooks/003-FWI-Anomaly-IP-IS-Den.ipynb cells 3,7 + Main-003-...py

Models are analytic (homogeneous + three 16x16-cell boxes); nz=101, nx=201, dx=dz=20, dt=2e-3, nt=1501,
f0=10, nPml=32, 19 shots at z=1, 181 DAS channels at z=95, mask rows nPml:nPml+4.
`run_iterate0(exp, ops)` evaluates misfit and gradient at the initial model through an operator backend
`ops` (the HIP fwi_ops on the GPU, or the CPU oracle adapter in CPU tests) and the UNCHANGED host chain
(padding, mask, lambda/mu formulas, autograd).
"""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KNOWN = json.load(open(os.path.join(GOLDEN, "known_answers.json")))

nx, nz, dx, dz, dt, nt, f0, nPml = 201, 101, 20.0, 20.0, 0.002, 1501, 10.0, 32


def models(exp):
    """(true vp, vs, rho), (init vp, vs, rho): float32 (nz, nx) arrays as np.loadtxt(...).astype('float32')
    of the notebook's np.savetxt(model.T) would give."""
    vp = np.ones((nx, nz)) * 4000.0
    vs = np.ones((nx, nz)) * 4000.0 / 1.732
    rho = np.ones((nx, nz)) * 2500.0
    vp0, vs0, rho0 = vp.copy(), vs.copy(), rho.copy()
    if exp == "001":
        vp[42:58, 42:58] += 80.0
        vs[92:108, 42:58] -= 80.0 / 1.732
        rho[142:158, 42:58] += 40
    elif exp == "002":
        mu = rho * vs ** 2
        lam = rho * vp ** 2 - 2 * mu
        lam[42:58, 42:58] += lam[0, 0] * 0.025
        mu[92:108, 42:58] -= mu[0, 0] * 0.025
        rho[142:158, 42:58] += rho[0, 0] * 0.020
        vp = np.sqrt((lam + 2 * mu) / rho)
        vs = np.sqrt(mu / rho)
    elif exp == "003":
        IP, IS = vp * rho, vs * rho
        IP[42:58, 42:58] += IP[0, 0] * 0.025
        IS[92:108, 42:58] -= IS[0, 0] * 0.025
        rho[142:158, 42:58] += 50
        vp, vs = IP / rho, IS / rho
    else:
        raise KeyError(exp)
    f = lambda a: np.ascontiguousarray(a.T, dtype=np.float32)
    return (f(vp), f(vs), f(rho)), (f(vp0), f(vs0), f(rho0))


def setup(exp, workdir):
    from sepfwi import utils as ft
    nPad = ft.nPad_for(nz, nPml)
    nz_pad, nx_pad = nz + 2 * nPml + nPad, nx + 2 * nPml
    Mask = np.zeros((nz_pad, nx_pad))
    Mask[nPml:nPml + nz, nPml:nPml + nx] = 1.0
    Mask[nPml:nPml + 4, :] = 0.0
    ind_src_x = np.arange(10, nx - 10, 10).astype(int)
    ind_src_z = np.ones(ind_src_x.shape[0]).astype(int)
    ind_rec_x = np.arange(10, nx - 10).astype(int)
    ind_rec_z = 95 * np.ones(ind_rec_x.shape[0]).astype(int)
    os.makedirs(workdir, exist_ok=True)
    para_fname = os.path.join(workdir, "para_file.json")
    survey_fname = os.path.join(workdir, "survey_file.json")
    ft.paraGen(nz_pad, nx_pad, dz, dx, nt, dt, f0, nPml, nPad, para_fname, survey_fname, os.path.join(workdir, "Data"))
    ft.surveyGen(ind_src_z, ind_src_x, ind_rec_z, ind_rec_x, survey_fname)
    Stf = torch.tensor(ft.sourceGene(f0, nt, dt), dtype=torch.float32).repeat(len(ind_src_x), 1)
    Shot_ids = torch.tensor(np.arange(0, len(ind_src_x)), dtype=torch.int32)
    opt = dict(nz=nz, nx=nx, nz_orig=nz, nx_orig=nx, nPml=nPml, nPad=nPad, para_fname=para_fname)
    return dict(opt=opt, Mask=torch.tensor(Mask, dtype=torch.float32), Stf=Stf, Shot_ids=Shot_ids, nPad=nPad,
                para_fname=para_fname)


def run_iterate0(exp, workdir, ngpu=1, device=None):
    """-> dict(f, ginf, grads{name: array}) at the initial model (what L-BFGS-B prints at iterate 0).
    device: None/"cpu" = the reference's host tensors; "cuda" = every tensor of the chain lives in HBM."""
    from sepfwi import modules as M
    from sepfwi import utils as ft
    from sepfwi.obj_wrapper import PyTorchObjective
    dev = torch.device(device or "cpu")
    su = setup(exp, workdir)
    (vp_t, vs_t, rho_t), (vp_i, vs_i, rho_i) = models(exp)
    pad = lambda a: torch.tensor(ft.padding_numpy_array(a, nPml, su["nPad"]), dtype=torch.float32, device=dev)
    Stf, Mask = su["Stf"].to(dev), su["Mask"].to(dev)
    M.FWI_obscalc(pad(vp_t), pad(vs_t), pad(rho_t), Stf, su["para_fname"])(su["Shot_ids"], ngpu=ngpu)
    T = lambda a: torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=True)
    fwi = _module_for(exp, su, (vp_i, vs_i, rho_i), T, Stf, Mask)
    obj = PyTorchObjective(fwi, lambda: fwi(su["Shot_ids"], ngpu=ngpu))
    jac = obj.jac            # the reference's quirk: cache() shadows .jac with the array (obj_wrapper.py:86)
    f = obj.fun(obj.x0)
    g = jac(obj.x0)
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in fwi.named_parameters()}
    return dict(f=f, ginf=float(np.abs(g).max()), grads=grads, N=int(obj.x0.size))


def _module_for(exp, su, models_i, T, Stf, Mask):
    """The reference's nn.Module of the experiment at its initial model (Main-00{1,2,3}-...py:116-124)."""
    from sepfwi import modules as M
    vp_i, vs_i, rho_i = models_i
    if exp == "001":
        return M.FWI(T(vp_i), T(vs_i), T(rho_i), Stf, su["opt"], Mask=Mask)
    if exp == "002":
        lam_i = rho_i * (vp_i ** 2 - 2.0 * vs_i ** 2) / 1e6          # Main-002:119-120
        mu_i = rho_i * vs_i ** 2 / 1e6
        return M.FWI_Lame_Den(T(lam_i), T(mu_i), T(rho_i), Stf, su["opt"], Mask=Mask)
    vpk, vsk = vp_i / 1e3, vs_i / 1e3                                  # Main-003:119-122
    return M.FWI_IP_IS_Den(T(vpk * rho_i), T(vsk * rho_i), T(rho_i), Stf, su["opt"], Mask=Mask)


def run_lbfgs(exp, workdir, nIter=3, ngpu=1, with_projg=False):
    """The reference's inversion driver, unchanged in structure (Main-001-...py:126-168): SciPy L-BFGS-B with the
    reference's options on top of PyTorchObjective.  Returns the list of misfits at the accepted iterates (and, with
    with_projg, the gradient inf-norms there: what the L-BFGS-B log prints as |proj g| for an unconstrained problem)."""
    from scipy import optimize
    from sepfwi import modules as M
    from sepfwi import utils as ft
    from sepfwi.obj_wrapper import PyTorchObjective
    su = setup(exp, workdir)
    (vp_t, vs_t, rho_t), models_i = models(exp)
    pad = lambda a: torch.tensor(ft.padding_numpy_array(a, nPml, su["nPad"]), dtype=torch.float32)
    M.FWI_obscalc(pad(vp_t), pad(vs_t), pad(rho_t), su["Stf"], su["para_fname"])(su["Shot_ids"], ngpu=ngpu)
    T = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
    fwi = _module_for(exp, su, models_i, T, su["Stf"], su["Mask"])
    obj = PyTorchObjective(fwi, lambda: fwi(su["Shot_ids"], ngpu=ngpu))
    hist, projg = [], []
    fun, jac = obj.fun, obj.jac
    f0 = fun(obj.x0)
    hist.append(f0)
    projg.append(float(np.abs(jac(obj.x0)).max()))

    def cb(x):
        hist.append(fun(x))                      # cached: the line search ended at x
        projg.append(float(np.abs(jac(x)).max()))

    optimize.minimize(fun, obj.x0, method="L-BFGS-B", jac=jac, bounds=obj.bounds, tol=None, callback=cb,
                      options={"gtol": 1e-16, "maxiter": nIter, "ftol": 1e-12, "maxcor": 5, "maxfun": 1500, "maxls": 6})
    return (hist, projg) if with_projg else hist
