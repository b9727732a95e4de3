#!/usr/bin/env python
"""tests/golden/oracle_long4000.npz: the CPU oracle (oracle/torchfwi_oracle.c, the float32 restatement of the reference's
cufd) on a HEADLINE-LENGTH run -- 4000 time steps, forward + boundary-saving adjoint -- on a grid it can afford
(300 x 150 cells + 20-cell layers, tests/problems.py LONG_RUN).  SURVEY.md Appendix A-18: reverse-time reconstruction
cancels to round-off only if forward and reverse kernels keep the same expression order; over 4000 steps a fused or
re-ordered GPU kernel that breaks it shows up here.  Run once in the build container (about 3 minutes on one core):

    python scripts/make_golden_long.py

Stored: the observed axial-strain gather of the "true" model, and for the initial model misfit, gradients (physical
interior only) and source-function gradient; plus a checksum of the padded models so a
drift of the problem generator is detected instead of being compared against stale numbers."""
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import problems as P  # noqa: E402
from oracle import oracle as O  # noqa: E402


def model_digest(pb):
    h = hashlib.sha256()
    for t in list(pb["lame_true"]) + list(pb["lame_init"]) + [pb["Stf"]]:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def main():
    O.build()
    with tempfile.TemporaryDirectory() as d:
        pb = P.make_long_problem(d)
        ids = pb["Shot_ids"].numpy()
        stf = pb["Stf"].numpy()
        t0 = time.time()
        lam, mu, den = [t.numpy() for t in pb["lame_true"]]
        obs = O.cufd(lam, mu, den, stf, 2, ids, pb["para"], pb["survey"])["syn"]
        print("observe: %.1f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        lam, mu, den = [t.numpy() for t in pb["lame_init"]]
        ref = O.cufd(lam, mu, den, stf, 1, ids, pb["para"], pb["survey"], obs=obs)
        print("gradient: %.1f s, misfit %.6e" % (time.time() - t0, ref["misfit"]), flush=True)
        nPml, nz, nx = pb["nPml"], P.LONG_RUN["nz"], P.LONG_RUN["nx"]
        crop = lambda g: np.ascontiguousarray(g[nPml:nPml + nz, nPml:nPml + nx + 1])   # + the column the x+1 spray reaches
        out = os.path.join(ROOT, "tests", "golden", "oracle_long4000.npz")
        np.savez_compressed(out, obs_ett=obs[0, 3], misfit=np.float64(ref["misfit"]),
                            gLambda=crop(ref["gLambda"]), gMu=crop(ref["gMu"]), gDen=crop(ref["gDen"]), gStf=ref["gStf"][0],
                            obs_peak=np.array([np.abs(obs[0, k]).max() for k in range(4)]), digest=model_digest(pb))
        full = [ref["gLambda"], ref["gMu"], ref["gDen"]]
        for g in full:   # nothing outside the cropped window
            z = g.copy()
            z[nPml:nPml + nz, nPml:nPml + nx + 1] = 0
            assert not z.any()
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
