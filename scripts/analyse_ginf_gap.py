#!/usr/bin/env python
"""Where do the small |g|_inf gaps between the reference's printed L-BFGS-B logs and the CPU oracle come from?
(tests/golden/known_answers.json "_ginf_analysis").  Works on the committed oracle gradients of the three experiments:
locates each maximum, prints its neighbourhood, separates the raw density gradient of experiment 003 from the chain-rule
terms and tests the hypothesis "the reference's raw density gradient is the oracle's times a constant"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import experiments as E  # noqa: E402

np.set_printoptions(precision=4, linewidth=160, suppress=True)
G = {e: np.load(os.path.join(ROOT, "tests", "golden", "oracle_exp%s_iterate0.npz" % e)) for e in ("001", "002", "003")}
for e, g in G.items():
    for k in g.keys():
        if g[k].ndim == 2:
            i = np.unravel_index(np.abs(g[k]).argmax(), g[k].shape)
            print(e, k, "max|.| %.6f at (z,x) = %s" % (np.abs(g[k]).max(), tuple(int(v) for v in i)))
    print(e, "oracle |g|_inf %.6f   printed %.6f   ratio %.5f" % (float(g["ginf"]), E.KNOWN[e]["ginf"], float(g["ginf"]) / E.KNOWN[e]["ginf"]))
print("\n002 raw density gradient around its maximum:\n", G["002"]["grad_Den"][50:57, 90:99])
# 003: d/drho = g_rho(raw) - gLambda (IP^2 - 2 IS^2) / rho^2 - gMu IS^2 / rho^2   (FWI_ops.py:261-262)
(_, _, _), (vp, vs, rho) = E.models("003")
IP, IS = vp / 1e3 * rho, vs / 1e3 * rho
gl = G["003"]["grad_IP"] / (2 * IP / rho)
gm = (G["003"]["grad_IS"] + gl * 4 * IS / rho) / (2 * IS / rho)
chain = -gl * (IP ** 2 - 2 * IS ** 2) / rho ** 2 - gm * IS ** 2 / rho ** 2
raw = G["003"]["grad_Den"] - chain
s = E.KNOWN["002"]["ginf"] / float(G["002"]["ginf"])
z, x = np.unravel_index(np.abs(G["003"]["grad_Den"]).argmax(), raw.shape)
print("\n003: raw g_rho at the maximum (%d,%d): %.4f; chain-rule part %.4f" % (z, x, raw[z, x], chain[z, x]))
print("003 with raw g_rho scaled by 002's ratio %.5f: |g|_inf = %.4f (printed %.5f, oracle %.5f)" %
      (s, np.abs(s * raw + chain).max(), E.KNOWN["003"]["ginf"], float(G["003"]["ginf"])))
need = (float(G["003"]["ginf"]) - E.KNOWN["003"]["ginf"]) / abs(raw[z, x])
print("raw-density deficit that reproduces the printed 003 value at that cell: %.2f %%  (002: %.2f %%)" % (100 * need, 100 * (1 - s)))
