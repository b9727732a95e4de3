#!/usr/bin/env python
"""Generate golden vectors from the reference's own Python code (build container only).

Imports DAS_Waveform_Modeling/src/elasticSolver.py and analyticalSolution.py FROM /root/reference
(never copied) and stores inputs + outputs as small .npz fixtures under tests/golden/.

numba is not installed in the image, so ``numba.jit`` is provided as the identity decorator: the
reference kernels then run as the plain Python they are written in (jit does not change results).

  python scripts/make_golden_numba.py small      # ~10 s   heterogeneous 48x40, 100 steps
  python scripts/make_golden_numba.py config1    # ~6 min  BASELINE config 1: 200x200, 500 steps
  python scripts/make_golden_numba.py analytic   # ~1-2 min Aki-Richards 2-D line-source solution
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/DAS_Waveform_Modeling/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault("numba", nb)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import elasticSolver  # noqa
    return elasticSolver


def smooth_random(rng, shape, lo, hi, passes=6):
    a = rng.standard_normal(shape)
    for _ in range(passes):
        a = 0.25 * (np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1))
    a = (a - a.min()) / (a.max() - a.min())
    return lo + (hi - lo) * a


def run(es, **kw):
    solver = es.elasticSolver(kw["nx"], kw["nz"], kw["ndamp"], kw["dx"], kw["dz"], kw["dt"], kw["nt"], kw["f0"],
                              kw["vp"], kw["vs"], kw["rho"], kw["src_coord"], kw["das_coord"], kw["geo_coord"],
                              kw["das_sensitivity"])
    return [solver.forward_it(i, False) for i in range(len(kw["src_coord"]))]


def small():
    es = import_reference()
    rng = np.random.default_rng(2023)
    nx, nz, ndamp = 48, 40, 8
    dx = dz = 10.0
    vp = smooth_random(rng, (nx, nz), 2500.0, 4000.0)
    vs = vp / smooth_random(rng, (nx, nz), 1.6, 1.9)
    rho = smooth_random(rng, (nx, nz), 2000.0, 2600.0)
    kw = dict(nx=nx, nz=nz, ndamp=ndamp, dx=dx, dz=dz, dt=1.0e-3, nt=100, f0=25.0, vp=vp, vs=vs, rho=rho,
              src_coord=np.array([[24 * dx, 20 * dz], [10 * dx, 8 * dz]]),
              das_coord=np.array([[12 * dx, 10 * dz], [30 * dx, 25 * dz], [40 * dx, 33 * dz]]),
              geo_coord=np.array([[14 * dx, 30 * dz], [30 * dx, 12 * dz], [24 * dx, 24 * dz]]),
              das_sensitivity=rng.uniform(-1, 1, (3, 6)))
    sol = run(es, **kw)
    out = {k: v for k, v in kw.items()}
    for i, s in enumerate(sol):
        for c in ("vx", "vz", "pr", "exx", "ezz", "exz", "ett"):
            out["shot%d_%s" % (i, c)] = s[c]
    np.savez_compressed(os.path.join(OUT, "numba_small.npz"), **out)
    print("wrote numba_small.npz")


def config1_setup():
    nx = nz = 200
    dx = dz = 10.0
    vp = np.ones((nx, nz)) * 4000.0
    vs = vp / np.sqrt(3)
    rho = np.ones((nx, nz)) * 2500.0
    src = np.array([[100 * dx, 100 * dz]])
    rec = np.array([[60 * dx, 70 * dz], [130 * dx, 140 * dz], [100 * dx, 55 * dz], [145 * dx, 100 * dz]])
    return dict(nx=nx, nz=nz, ndamp=40, dx=dx, dz=dz, dt=1.0e-3, nt=500, f0=10.0, vp=vp, vs=vs, rho=rho,
                src_coord=src, das_coord=rec, geo_coord=rec, das_sensitivity=np.zeros((4, 6)))


def config1():
    es = import_reference()
    kw = config1_setup()
    sol = run(es, **kw)[0]
    out = {k: v for k, v in kw.items() if k not in ("vp", "vs", "rho")}
    out.update(vp0=4000.0, vs0=4000.0 / np.sqrt(3), rho0=2500.0)
    for c in ("vx", "vz", "pr", "exx", "ezz", "exz"):
        out[c] = sol[c]
    np.savez_compressed(os.path.join(OUT, "numba_config1.npz"), **out)
    print("wrote numba_config1.npz")


def analytic():
    sys.path.insert(0, REF)
    from analyticalSolution import AnalyticalSolution
    kw = config1_setup()
    src = kw["src_coord"][0]
    out = dict(receivers=kw["geo_coord"], src=src, vp=4000.0, vs=4000.0 / np.sqrt(3), rho=2500.0,
               f0=10.0, dt=1.0e-3, tmax=0.5)
    M = np.eye(3)
    for r in (0, 1):   # two receivers (one oblique each side)
        x = abs(kw["geo_coord"][r, 0] - src[0])
        z = abs(kw["geo_coord"][r, 1] - src[1])
        V = AnalyticalSolution(4000.0, 4000.0 / np.sqrt(3), 2500.0, x, 0, z, 0.0, 0.5, 1.0e-3, 10.0, 1e16, M,
                               dim="2D", comp="displacement", verbose=False)
        out["rec%d_Ux" % r] = V["Ux"]
        out["rec%d_Uz" % r] = V["Uz"]
        out["rec%d_t" % r] = V["t"] if "t" in V else np.arange(len(V["Ux"])) * 1e-3
    np.savez_compressed(os.path.join(OUT, "analytic_config1.npz"), **out)
    print("wrote analytic_config1.npz")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    {"small": small, "config1": config1, "analytic": analytic}[sys.argv[1]]()
