#!/usr/bin/env python
"""Directional-derivative check of the GPU gradient at the headline size (2000x1000, 4000 steps, 3 shots): the misfit
along -g must fall as -a |g|^2 predicts (measured: within 0.3-1 % for model changes of 0.01 ... 10 m/s).  Needs a GPU."""
import os, sys, tempfile, numpy as np, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import bench
from sepfwi import fwi_ops, modules as M, utils as ft
from sepfwi.obj_wrapper import PyTorchObjective
dev = torch.device("cuda", 0)
nz, nx, ns, shots = 1000, 2000, 4000, 3
work = tempfile.mkdtemp()
nPml = 32; nPad = ft.nPad_for(nz, nPml)
pb = bench.setup_problem(work, nz, nx, ns, shots)
true, init = bench.marmousi_style(nz, nx)
Stf = pb["Stf"].to(dev); ids = torch.arange(shots, dtype=torch.int32)
opt = dict(nz=nz, nx=nx, nz_orig=nz, nx_orig=nx, nPml=nPml, nPad=nPad, para_fname=pb["para_fname"])
pad = lambda m: torch.tensor(ft.padding_numpy_array(m, nPml, nPad), dtype=torch.float32, device=dev)
M.FWI_obscalc(pad(true[0]), pad(true[1]), pad(true[2]), Stf, pb["para_fname"])(ids, ngpu=1)
Mask = torch.zeros((pb["nz_pad"], pb["nx_pad"]), dtype=torch.float32, device=dev)
Mask[nPml + 4:nPml + nz, nPml:nPml + nx] = 1.0
T = lambda m: torch.tensor(m, dtype=torch.float32, device=dev, requires_grad=True)
fwi = M.FWI(T(init[0]), T(init[1]), T(init[2]), Stf, opt, Mask=Mask)
obj = PyTorchObjective(fwi, lambda: fwi(ids, ngpu=1))
fun, jac = obj.fun, obj.jac
f0 = fun(obj.x0); g = jac(obj.x0).copy()
n = nz * nx
print("f0 %.8e  |g|2 %.4e  |g|inf %.4e  per-block inf: vp %.3e vs %.3e rho %.3e" % (f0, np.linalg.norm(g), np.abs(g).max(), np.abs(g[:n]).max(), np.abs(g[n:2*n]).max(), np.abs(g[2*n:]).max()))
print("nan in g:", np.isnan(g).any(), " g finite:", np.isfinite(g).all())
for dmax in (1e-3, 1e-2, 0.1, 1.0, 10.0):
    a = dmax / np.abs(g).max()
    f1 = fun(obj.x0 - a * g)
    print("max change %.3g  -> f-f0 = %.6e   predicted %.6e" % (dmax, f1 - f0, -a * np.dot(g, g)))
