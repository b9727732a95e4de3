"""Seeded synthetic set-ups shared by the parity tests (inputs only; no reference code)."""
import contextlib
import json
import os

import numpy as np
import torch

from sepfwi import utils as ft


def smooth_random(rng, shape, lo, hi, passes=8):
    a = rng.standard_normal(shape)
    for _ in range(passes):
        a = 0.25 * (np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1))
    a = (a - a.min()) / (a.max() - a.min())
    return lo + (hi - lo) * a


def make_problem(workdir, nz=44, nx=60, nPml=10, nSteps=240, nshots=2, dh=10.0, dt=1.0e-3, f0=25.0,
                 hetero=True, seed=7, rec_z=None, src_z=2, nrec_stride=1, nPad=None, src_x=None, das_fiber="horizontal",
                 rec_x=None, stf=None, das_sensitivity=None, dz=None):
    """Writes para/survey JSON under workdir and returns everything a test needs.
    Models: `true` (with anomalies) and `init` (smooth), both (nz, nx) float32, plus padded versions."""
    rng = np.random.default_rng(seed)
    if nPad is None:
        nPad = ft.nPad_for(nz, nPml)
    nz_pad, nx_pad = nz + 2 * nPml + nPad, nx + 2 * nPml
    if hetero:
        vp0 = smooth_random(rng, (nz, nx), 2600.0, 3800.0)
        vs0 = vp0 / smooth_random(rng, (nz, nx), 1.65, 1.85)
        rho0 = smooth_random(rng, (nz, nx), 2100.0, 2600.0)
    else:
        vp0 = np.full((nz, nx), 3000.0)
        vs0 = vp0 / 1.732
        rho0 = np.full((nz, nx), 2400.0)
    vp1, vs1, rho1 = vp0.copy(), vs0.copy(), rho0.copy()
    z0, x0 = nz // 2, nx // 3
    vp1[z0 - 4:z0 + 4, x0 - 4:x0 + 4] *= 1.05
    vs1[z0 - 4:z0 + 4, 2 * x0 - 4:2 * x0 + 4] *= 0.95
    rho1[z0 + 5:z0 + 11, nx // 2 - 4:nx // 2 + 4] *= 1.04
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    true = dict(vp=f32(vp1), vs=f32(vs1), rho=f32(rho1))
    init = dict(vp=f32(vp0), vs=f32(vs0), rho=f32(rho0))

    os.makedirs(workdir, exist_ok=True)
    para_fname = os.path.join(workdir, "para_file.json")
    survey_fname = os.path.join(workdir, "survey_file.json")
    data_dir = os.path.join(workdir, "Data")
    ft.paraGen(nz_pad, nx_pad, dh if dz is None else dz, dh, nSteps, dt, f0, nPml, nPad, para_fname, survey_fname, data_dir, das_fiber=das_fiber)
    src_x = np.linspace(6, nx - 7, nshots).round().astype(int) if src_x is None else np.asarray(src_x, dtype=int)
    src_zs = np.full(nshots, src_z, dtype=int)
    rec_x = np.arange(4, nx - 4, nrec_stride).astype(int) if rec_x is None else np.asarray(rec_x, dtype=int)
    rec_zs = np.full(rec_x.shape, (nz - 6) if rec_z is None else rec_z, dtype=int)
    if das_fiber == "vertical":   # a borehole fibre: one column, consecutive depths
        rec_zs = np.arange(4, nz - 4, nrec_stride).astype(int)
        rec_x = np.full(rec_zs.shape, nx // 2 + 3, dtype=int)
    if das_sensitivity == "random":   # a shaped fibre: every channel its own direction cosines (seeded)
        th = np.random.default_rng(seed + 99).uniform(0.0, np.pi, rec_x.size)
        das_sensitivity = np.zeros((rec_x.size, 6))
        das_sensitivity[:, 0], das_sensitivity[:, 3], das_sensitivity[:, 1] = np.cos(th) ** 2, np.sin(th) ** 2, 2.0 * np.sin(th) * np.cos(th)
    ft.surveyGen(src_zs, src_x, rec_zs, rec_x, survey_fname, Das_sensitivity=das_sensitivity)
    stf = ft.sourceGene(f0, nSteps, dt) if stf is None else np.asarray(stf)
    Stf = torch.tensor(stf, dtype=torch.float32).repeat(nshots, 1)
    Shot_ids = torch.arange(nshots, dtype=torch.int32)
    opt = dict(nz=nz, nx=nx, nz_orig=nz, nx_orig=nx, nPml=nPml, nPad=nPad, para_fname=para_fname)

    def padded(m):
        t = {k: torch.tensor(ft.padding_numpy_array(v, nPml, nPad)) for k, v in m.items()}
        lam = (t["vp"] ** 2 - 2.0 * t["vs"] ** 2) * t["rho"] / 1e6   # FWI_ops.py:134-135
        mu = t["vs"] ** 2 * t["rho"] / 1e6
        return lam.contiguous(), mu.contiguous(), t["rho"].contiguous()

    return dict(para_fname=para_fname, survey_fname=survey_fname, data_dir=data_dir, opt=opt, true=true, init=init,
                Stf=Stf, Shot_ids=Shot_ids, nrec=int(rec_x.size), nSteps=nSteps, nPml=nPml, nPad=nPad,
                nz_pad=nz_pad, nx_pad=nx_pad, lame_true=padded(true), lame_init=padded(init),
                para=json.load(open(para_fname)), survey=json.load(open(survey_fname)))


def sustained_source(f0, nSteps, dt, period=0.31):
    """Ricker wavelets (fwi_utils.sourceGene) re-fired every `period` seconds with changing sign and size: keeps the
    wavefield alive over thousands of time steps, so a long run tests more than the decay of one pulse."""
    base = ft.sourceGene(f0, nSteps, dt)
    out = np.zeros(nSteps)
    k, shift = 0, 0
    while shift < nSteps:
        out[shift:] += base[: nSteps - shift] * (1.0 if k % 2 == 0 else -0.7) * (1.0 + 0.13 * (k % 5))
        k += 1
        shift = int(round(k * period / dt))
    return out


LONG_RUN = dict(nz=150, nx=300, nPml=20, nSteps=4000, nshots=1, f0=15.0, seed=11, src_x=[120], rec_x=list(range(100, 164)))


def make_long_problem(workdir):
    """The 4000-step problem of tests/golden/oracle_long4000.npz (scripts/make_golden_long.py): 300 x 150 cells + 20-cell
    layers, one shot with a sustained source, a 64-channel DAS line (the fused in-kernel sampling / injection path)."""
    kw = dict(LONG_RUN)
    return make_problem(workdir, hetero=True, stf=sustained_source(kw["f0"], kw["nSteps"], 1.0e-3), **kw)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a))


# defaults of struct KernelOptions (csrc/kernels.hpp); `probe` is bench.py's business
OPTION_DEFAULTS = dict(bz=2, xcd_remap=1, bwd_fuse=4, line_fuse=1, pair_fwd=1, fwd_lanes=3, early=0, rho_fly=1, amu_fly=1,
                       rk_lazy=1, batch=2, batch_f=0, batch_b=0, batch_mb=200, batch_order=1, batch_split=2, img_every=1, obs_cache_mb=0, quiet_skip=0, quiet_rows=4, pk_lmask=16, pk_wpc=2, pk_px=3, pk_nosync=0, pk_lock=0, pk_snake=1, pk_ms=0, pk_quiet=0, pk_waves=16, pk_order=2, pk_prio=1, pk_wx=150, pk_wxp=150, pk_wz=115)


def _needs_probes(opts):
    from sepfwi import _native
    return any(k not in _native.PUBLIC_OPTIONS for k in opts)


@contextlib.contextmanager
def kernel_options(**opts):
    """Process-wide kernel options for the duration of a block, defaults restored afterwards.  Options the shipped library does not
    expose (everything but _native.PUBLIC_OPTIONS: tile shapes, launch structures, ...) route the block to the -DSEPFWI_PROBES build of
    the same sources (libsepfwi_probes.so), which has its own sessions and its own option block; a test whose blocks share a session
    (observed data stored in it, statistics) asks for the `probes_lib` fixture instead and runs on that build from start to end."""
    from sepfwi import _native
    switched = _needs_probes(opts) and _native._active != "probes"
    variant = "probes" if (switched or _native._active == "probes") else "default"
    with _native.use_variant(variant) as L:
        try:
            for k, v in opts.items():
                _native.check(L.sepfwi_set_option(k.encode(), int(v)))
            yield
        finally:
            for k, v in OPTION_DEFAULTS.items():
                if variant == "probes" or k in _native.PUBLIC_OPTIONS:
                    L.sepfwi_set_option(k.encode(), int(v))
            if switched:
                L.sepfwi_release_all()      # nothing of a probe session outlives its block (HBM)
