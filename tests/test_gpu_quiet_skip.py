"""Option quiet_skip (-m gpu): updates of row segments whose every input is exactly +0 -- the fields ahead of the wave front -- are
left out.  An update of zeros stores zeros, so NOTHING may change: seismograms, misfit, the three gradients and the source gradient
are compared bit for bit with the option off, in every launch structure that honours it (streams, batched launches, the two-launch
backward step that replaces the persistent loop while the option is on), on grids where the wave front really leaves most of the
grid quiet for most of the record, with the source and the fibre inside and next to the absorbing layers."""
import os

import numpy as np
import pytest
import torch

import problems as P

pytestmark = pytest.mark.gpu

_NAMES = ("misfit", "gLambda", "gMu", "gDen", "gStf")


def _gathers(pb):
    out = []
    for sid in pb["Shot_ids"].tolist():
        for c in ("pr", "vx", "vz", "ett"):
            f = os.path.join(pb["data_dir"], "Shot_%s%d.bin" % (c, sid))
            if os.path.exists(f):
                out.append(np.fromfile(f, dtype=np.float32))
    return out


@pytest.mark.parametrize("mode", [dict(batch=0), dict(batch=0, pair_fwd=0), dict(batch=1), dict(batch=1, batch_f=2, batch_b=1),
                                  dict(batch=0, bwd_fuse=2, img_every=2), dict(batch=0, early=3, bwd_fuse=2)])
def test_quiet_skip_changes_nothing(tmp_path, hip_ops, mode, probes_lib):
    pb = P.make_problem(str(tmp_path), nz=260, nx=900, nPml=12, nSteps=420, nshots=3, hetero=True, rec_z=30)   # the front crosses a third of the grid
    lt, mt, dt_ = pb["lame_true"]
    lam, mu, den = pb["lame_init"]
    lam = (lam * 1.04).contiguous()
    res = {}
    for q in (0, 1):
        with P.kernel_options(quiet_skip=q, **mode):
            hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])          # all four components to files
            data = _gathers(pb)
            out = [t.numpy().copy() for t in hip_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])]
            st = hip_ops.stats(pb["para_fname"], 0)
            res[q] = (data, out, st)
        hip_ops.release()
    assert len(res[0][0]) == len(res[1][0]) > 0
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for name, a, b in zip(_NAMES, res[0][1], res[1][1]):
        assert np.array_equal(a, b), (mode, name, float(np.abs(a - b).max()), float(np.abs(a).max()))
    assert np.abs(res[0][1][1]).max() > 0 and np.abs(res[0][1][3]).max() > 0 and np.abs(res[0][1][4]).max() > 0
    assert 0 < res[1][2]["quiet_active"] < 0.8 * res[1][2]["quiet_total"], res[1][2]    # the maps were in use and much of the grid never held a value
    assert res[0][2]["quiet_total"] == 0


@pytest.mark.parametrize("geo", [dict(nz=90, nx=1300, nPml=20, nSteps=500, src_z=1, rec_z=2),          # source and fibre in the first rows under the top layer
                                 dict(nz=500, nx=200, nPml=16, nSteps=900, nPad=5, rec_z=120),          # tall: the lower half of the grid stays quiet
                                 dict(nz=150, nx=700, nPml=8, nSteps=900, rec_z=100)])                  # long record: the grid fills up and the layers absorb
def test_quiet_skip_on_other_geometries(tmp_path, hip_ops, geo, probes_lib):
    pb = P.make_problem(str(tmp_path), nshots=2, hetero=True, **geo)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"], to_store=True)
    lam, mu, den = pb["lame_init"]
    lam = (lam * 1.05).contiguous()
    outs = []
    nseg_rows = (geo["nz"] + 2 * geo["nPml"]) * ((geo["nx"] + 2 * geo["nPml"] + 63) // 64)
    for q in (0, 1):
        # the two-launch step (what the shipped library runs with the option on) and the persistent loop's own quiet variant (words per row
        # segment in LDS, neighbour summaries in the phase flags; bit-identical, slower -- EXPERIMENTS #49 -- hence in the probe build only)
        for fuse in (4, 2):
            with P.kernel_options(quiet_skip=q, batch=0, bwd_fuse=fuse, pk_quiet=1):
                outs.append([t.numpy().copy() for t in hip_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])])
                steps = hip_ops.stats(pb["para_fname"], 0)["persist_steps"]
                assert steps == (2 * (pb["nSteps"] - 1) if fuse == 4 and nseg_rows >= 2048 else 0), (geo, q, fuse, steps)
    for k in range(1, 4):
        for name, a, b in zip(_NAMES, outs[0], outs[k]):
            assert np.array_equal(a, b), (geo, k, name, float(np.abs(a - b).max()), float(np.abs(a).max()))
    assert np.abs(outs[0][1]).max() > 0


_SEEDS = list(range(int(os.environ.get("SEPFWI_QFUZZ_N", "5"))))


@pytest.mark.parametrize("seed", _SEEDS)
def test_quiet_skip_random_geometry(tmp_path, hip_ops, seed, probes_lib):
    """Seeded random grids, layer widths, record lengths, source and fibre depths, launch structures: with and without the option the
    same bits.  One-off sweeps: SEPFWI_QFUZZ_N=300 (profiles/r05_quiet_fuzz.txt)."""
    rng = np.random.default_rng(7000 + seed)
    nPml = int(rng.integers(6, 33))
    nz, nx = int(rng.integers(40, 500)), int(rng.integers(100, 1500))
    nPad = int(rng.integers(0, 9))
    nSteps = int(rng.integers(120, 500))
    nshots = int(rng.integers(1, 4))
    pb = P.make_problem(str(tmp_path), nz=nz, nx=nx, nPml=nPml, nPad=nPad, nSteps=nSteps, nshots=nshots, hetero=True, seed=seed,
                        src_z=int(rng.integers(1, 5)), rec_z=int(rng.integers(2, max(3, min(nz - 3, 50)))), f0=float(rng.uniform(10.0, 30.0)))
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"], to_store=True)
    lam, mu, den = pb["lame_init"]
    lam = (lam * 1.05).contiguous()
    mode = dict(batch=int(rng.choice([0, 0, 1])), img_every=int(rng.choice([1, 1, 2])), bz=int(rng.choice([1, 2, 2, 4])), xcd_remap=int(rng.integers(0, 2)),
                pk_quiet=int(seed % 2),      # odd seeds: the persistent loop's quiet variant where the grid feeds the loop (probe build)
                rho_fly=int(rng.choice([1, 1, 0, 3])), pair_fwd=int(rng.integers(0, 2)))
    outs = []
    for q in (0, 1):
        with P.kernel_options(quiet_skip=q, **mode):
            outs.append([t.numpy().copy() for t in hip_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])])
            st = hip_ops.stats(pb["para_fname"], 0)
    hip_ops.release()
    desc = dict(seed=seed, nz=nz, nx=nx, nPml=nPml, nPad=nPad, nSteps=nSteps, nshots=nshots, **mode)
    assert st["quiet_total"] > 0 and st["quiet_active"] > 0, desc
    for name, a, b in zip(_NAMES, outs[0], outs[1]):
        assert np.array_equal(a, b), (desc, name, float(np.abs(a - b).max()), float(np.abs(a).max()))
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][4]).max() > 0, desc
