#!/usr/bin/env python
"""How launch-bound is a reference-notebook-sized problem (101x201, 1501 steps, 19 shots)?  Prints wall time of one
gradient evaluation and the GPU-side loop times of the session."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import torch
import experiments as E
from sepfwi import fwi_ops, _native
work = tempfile.mkdtemp()
for kv in os.environ.get("SEPFWI_OPTS", "").split(","):
    if kv:
        _native.check(_native.lib().sepfwi_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
r = E.run_iterate0("001", work, device="cuda")
su = E.setup("001", work)
import numpy as np
(vp_t, vs_t, rho_t), (vp_i, vs_i, rho_i) = E.models("001")
from sepfwi import utils as ft
pad = lambda a: torch.tensor(ft.padding_numpy_array(a, E.nPml, su["nPad"]), dtype=torch.float32, device="cuda")
vp, vs, rho = pad(vp_i), pad(vs_i), pad(rho_i)
lam, mu = (vp ** 2 - 2 * vs ** 2) * rho / 1e6, vs ** 2 * rho / 1e6
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fwi_ops.backward(lam, mu, rho, su["Stf"], 1, su["Shot_ids"], su["para_fname"])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    st = fwi_ops.stats(su["para_fname"], 0)
    print("evaluation %.1f ms wall; forward loops %.1f ms, backward loops %.1f ms (GPU events); %d launches -> %.2f us per launch; "
          "%.1f Gcell-updates/s" % (el * 1e3, st["fwd_ms"], st["bwd_ms"], st["launches"], el * 1e6 / st["launches"], st["cell_updates"] / el / 1e9))
