"""SciPy <-> torch glue for L-BFGS-B (reference: obj_wrapper.py:10-97): flat float64 vector in, (loss,
flat float64 gradient) out, with a one-entry cache so fun(x) and jac(x) share one propagation."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
from scipy import optimize


class PyTorchObjective(object):
    def __init__(self, obj, loss):
        self.obj = obj      # nn.Module holding the parameters (and .Bounds)
        self.loss = loss    # callable -> scalar tensor
        params = OrderedDict(obj.named_parameters())
        self.param_shapes = OrderedDict((n, tuple(p.shape)) for n, p in params.items())
        self.x0 = np.concatenate([p.data.cpu().numpy().ravel() for p in params.values()]).astype(np.float64)
        self.bounds = self.pack_bounds() if getattr(obj, "Bounds", {}) != {} else None

    def unpack_parameters(self, x):
        out, i = OrderedDict(), 0
        for n, shp in self.param_shapes.items():
            k = int(np.prod(shp))
            out[n] = torch.from_numpy(np.asarray(x[i:i + k]).reshape(shp))
            i += k
        return out

    def pack_grads(self):
        return np.concatenate([p.grad.data.cpu().numpy().ravel() for p in self.obj.parameters()]).astype(np.float64)

    def pack_bounds(self):
        lo = [np.asarray(self.obj.Bounds[n][0]).ravel() for n in self.param_shapes]
        hi = [np.asarray(self.obj.Bounds[n][1]).ravel() for n in self.param_shapes]
        return optimize.Bounds(np.concatenate(lo).astype(np.float64), np.concatenate(hi).astype(np.float64))

    def is_new(self, x):
        if not hasattr(self, "cached_x"):
            return True
        return np.abs(np.array(x) - np.array(self.cached_x)).max() > 1e-8

    def cache(self, x):
        state = self.unpack_parameters(x)
        for name, buf in self.obj.named_buffers():
            state[name] = buf
        self.obj.load_state_dict(state)
        self.cached_x = x
        self.obj.zero_grad()
        val = self.loss()
        self.f = val.item()
        val.backward()
        self.jac = self.pack_grads()

    def fun(self, x):
        if self.is_new(x):
            self.cache(x)
        return self.f

    def jac(self, x):
        if self.is_new(x):
            self.cache(x)
        return self.jac


_SETULB_SIGNATURE = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)"


def minimize_lbfgsb(fun, x0, jac, bounds=None, callback=None, maxiter=15000, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5,
                    maxfun=15000, maxls=20):
    """`optimize.minimize(fun, x0, method="L-BFGS-B", jac=jac, bounds=bounds, callback=callback, options={...})` -- what the
    reference's experiment scripts call (Main-001-FWI-Anomaly-Vp-Vs-Den.py:157-168) -- with the same compiled routine
    (`scipy.optimize._lbfgsb.setulb`) driven directly, so the iterates are the same bit for bit, but without SciPy's
    per-element Python loops over the bounds: on the 6 M unknowns of a 2000 x 1000 model those cost 10-20 s per minimize() call
    (new-style -> old-style -> new-style conversions and a dict look-up per variable), the one serial term of a multi-GPU
    inversion that does not shrink with the number of GPUs.  Here the bound codes are three vectorised
    numpy expressions.  `bounds`: an `optimize.Bounds` or None.  Falls back to `optimize.minimize` when there are no bounds
    (nothing to save) or when the installed SciPy's private routine does not have the signature this driver was written for."""
    options = dict(maxiter=maxiter, maxcor=maxcor, ftol=ftol, gtol=gtol, maxfun=maxfun, maxls=maxls)
    try:
        from scipy.optimize import _lbfgsb
        fast = bounds is not None and (_lbfgsb.setulb.__doc__ or "").strip().startswith(_SETULB_SIGNATURE)
    except ImportError:
        fast = False
    if not fast:
        return optimize.minimize(fun, x0, method="L-BFGS-B", jac=jac, bounds=bounds, callback=callback, options=options)
    x = np.array(x0, dtype=np.float64).ravel()
    n, m = x.size, int(maxcor)
    lb = np.broadcast_to(np.asarray(bounds.lb, dtype=np.float64), (n,))
    ub = np.broadcast_to(np.asarray(bounds.ub, dtype=np.float64), (n,))
    if (lb > ub).any():
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    if not maxls > 0:
        raise ValueError("maxls must be positive.")
    x = np.clip(x, lb, ub)
    has_lo, has_hi = np.isfinite(lb), np.isfinite(ub)
    nbd = np.where(has_lo & has_hi, 2, np.where(has_lo, 1, np.where(has_hi, 3, 0))).astype(np.int32)   # setulb's bound codes
    low_bnd, upper_bnd = np.where(has_lo, lb, 0.0), np.where(has_hi, ub, 0.0)
    factr = ftol / np.finfo(float).eps

    state = {"x": None, "f": None, "g": None, "nfev": 0}

    def fun_and_grad(xk):       # one evaluation per distinct point, as SciPy's ScalarFunction counts them
        if state["x"] is None or not np.array_equal(xk, state["x"]):
            state["x"] = np.copy(xk)
            state["f"] = float(fun(np.copy(xk)))
            state["g"] = np.asarray(jac(np.copy(xk)), dtype=np.float64)
            state["nfev"] += 1
        return state["f"], state["g"]

    fun_and_grad(x)             # SciPy evaluates at x0 while it builds its ScalarFunction
    f = np.array(0.0, dtype=np.float64)
    g = np.zeros(n, dtype=np.float64)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task, ln_task = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
    lsave, isave, dsave = np.zeros(4, dtype=np.int32), np.zeros(44, dtype=np.int32), np.zeros(29, dtype=np.float64)
    nit = 0
    while True:
        g = np.asarray(g, dtype=np.float64)
        _lbfgsb.setulb(m, x, low_bnd, upper_bnd, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        if task[0] == 3:        # FG: the routine wants f and g at x
            f, g = fun_and_grad(x)
        elif task[0] == 1:      # NEW_X: an iteration is complete
            nit += 1
            halt = False
            if callback is not None:
                try:
                    callback(np.copy(x))
                except StopIteration:
                    halt = True
            if halt:
                task[0], task[1] = 5, 505
            if nit >= maxiter:
                task[0], task[1] = 5, 504
            elif state["nfev"] > maxfun:
                task[0], task[1] = 5, 502
        else:
            break
    from scipy.optimize._lbfgsb_py import status_messages, task_messages
    status = 0 if task[0] == 4 else (1 if (state["nfev"] > maxfun or nit >= maxiter) else 2)
    n_corrs = min(int(isave[30]), m)
    s, y = wa[0:m * n].reshape(m, n), wa[m * n:2 * m * n].reshape(m, n)
    return optimize.OptimizeResult(fun=float(f), jac=g, nfev=state["nfev"], njev=state["nfev"], nit=nit, status=status,
                                   message=status_messages[task[0]] + ": " + task_messages[task[1]], x=x, success=(status == 0),
                                   hess_inv=optimize.LbfgsInvHessProduct(s[:n_corrs], y[:n_corrs]))
