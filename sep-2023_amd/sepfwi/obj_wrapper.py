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
