"""Headline-scale parity (-m gpu): what the small oracle problems cannot show.

* 4000 time steps (BASELINE.json's sweep length) against the CPU oracle on a grid it can afford, from a committed
  golden file (scripts/make_golden_long.py): seismograms <= 1e-4, misfit 1e-4, gradients <= 1e-3 -- in the stream
  structure (what runs at 2000 x 1000) and in the batched structure (what the heuristics pick for this grid).
* The full 2000 x 1000 x 4000 configuration through a size-independent property: after the
  backward pass the reverse-time reconstruction has been run back to time step 0 and must have returned to the zero
  initial state to <= 1e-5 of the peak forward amplitude.
* BASELINE.json configs[1] (2000 x 500, 2000 steps, forward only): against the CPU oracle at full size (round 3), full-size
  properties, plus an oracle-checked cropped twin of the same model, spacing, time step and source.
* BASELINE.json configs[2] (2000 x 1000, 4000 steps, forward + adjoint): one shot and all 32 shots against the CPU oracle at
  full size (round 3), the 32-shot call against the sum of its groups.
* A grid 7.5 x the headline's (5600 x 2800) for a few steps against the oracle: launch geometry and offsets at size.
"""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

import problems as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "oracle_long4000.npz")


def _digest(pb):
    h = hashlib.sha256()
    for t in list(pb["lame_true"]) + list(pb["lame_init"]) + [pb["Stf"]]:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def _interior(pb, f, nz, nx):
    n = pb["nPml"]
    return f[n:n + nz, n:n + nx]


@pytest.mark.timeout(600)
def test_long_run_4000_steps_matches_oracle(tmp_path, hip_ops):
    from sepfwi import utils as ft
    G = np.load(GOLDEN)
    pb = P.make_long_problem(str(tmp_path))
    assert _digest(pb) == str(G["digest"]), "problem generator drifted: regenerate with scripts/make_golden_long.py"
    nz, nx, nPml, nS = P.LONG_RUN["nz"], P.LONG_RUN["nx"], pb["nPml"], pb["nSteps"]
    crop = lambda g: g[nPml:nPml + nz, nPml:nPml + nx + 1]
    lt, mt, dt_ = pb["lame_true"]
    lam, mu, den = pb["lame_init"]
    for name, opts in (("streams", dict(batch=0)), ("batched", dict())):
        with P.kernel_options(**opts):
            hip_ops.release()
            hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
            ett = ft.read_shot_gather(pb["data_dir"], "ett", 0, nS)
            assert ett.shape == G["obs_ett"].shape
            e_obs = P.rel_l2(ett, G["obs_ett"])
            assert e_obs <= 1e-4, (name, e_obs)
            late = slice(3 * nS // 4, nS)                       # the last quarter of the run on its own
            assert np.abs(G["obs_ett"][:, late]).max() > 0.05 * np.abs(G["obs_ett"]).max()   # the source is still firing
            assert P.rel_l2(ett[:, late], G["obs_ett"][:, late]) <= 1e-4, name
            G["obs_ett"].tofile(os.path.join(pb["data_dir"], "Shot_ett0.bin"))   # the oracle's observed data, bit for bit
            hip_ops.release()
            m, gL, gM, gD, gS = hip_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
            assert abs(float(m) - float(G["misfit"])) <= 1e-4 * float(G["misfit"]), (name, float(m), float(G["misfit"]))
            for key, g in (("gLambda", gL), ("gMu", gM), ("gDen", gD)):
                g = g.numpy()
                r = G[key]
                e = P.rel_l2(crop(g), r)
                assert e <= 1e-3, (name, key, e)
                assert np.abs(crop(g) - r).max() <= 1e-3 * np.abs(r).max(), (name, key)
                z = g.copy()
                crop(z)[...] = 0
                assert not z.any(), (name, key)                  # nothing outside the window the oracle writes
            assert P.rel_l2(gS.numpy()[0], G["gStf"]) <= 1e-3, name
            # reverse-time reconstruction ran back to step 0: the physical interior has returned to the zero initial state
            peak_v = max(float(G["obs_peak"][1]), float(G["obs_peak"][2]))
            peak_s = float(G["obs_peak"][0])
            for which in range(5):
                f = _interior(pb, hip_ops.debug_field(pb["para_fname"], which).numpy(), nz, nx)
                assert np.abs(f).max() <= 1e-5 * (peak_v if which < 2 else peak_s), (name, which, np.abs(f).max())


@pytest.mark.timeout(900)
def test_headline_2000x1000x4000_reconstruction_returns_to_zero(tmp_path, hip_ops):
    """BASELINE.json configs[2] at FULL size and length (one shot): forward, boundary saving, 3999 reverse-time steps.
    The oracle cannot go there; the property can -- and it is sharp: any asymmetry between the forward and the reverse
    kernels, a wrong frame slot, a lost source sample leaves a wavefield behind instead of round-off."""
    sys.path.insert(0, ROOT)
    import bench
    from sepfwi import utils as ft
    nS = 4000
    pb = bench.setup_problem(str(tmp_path), 1000, 2000, nS, 3)
    ids = torch.tensor([1], dtype=torch.int32)                   # the shot in the middle of the line
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, ids, pb["para_fname"])
    data_dir = str(tmp_path / "Data")
    d = {c: ft.read_shot_gather(data_dir, c, 1, nS) for c in ("pr", "vx", "vz")}
    peak_v = max(np.abs(d["vx"]).max(), np.abs(d["vz"]).max())
    peak_s = np.abs(d["pr"]).max()                                # szz + sxx along the line through the source
    assert peak_v > 0 and peak_s > 0
    nP, nz, nx = pb["nPml"], 1000, 2000
    inner = lambda f: f[nP:nP + nz, nP:nP + nx]
    # forward only (misfit call): the final wavefield is NOT small -- the property below is not vacuous
    hip_ops.forward(lam, mu, den, pb["Stf"], 0, ids, pb["para_fname"])
    end_v = max(np.abs(inner(hip_ops.debug_field(pb["para_fname"], k).numpy())).max() for k in (0, 1))
    assert end_v > 1e-3 * peak_v
    m, gL, gM, gD, gS = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    st = hip_ops.stats(pb["para_fname"], 0)
    assert st["fwd_steps"] == nS - 1 and st["bwd_steps"] == nS - 1 and st["persist_steps"] == nS - 1
    # ... and the two-launch step gives the same bits at the full size (3999 time steps x 512 tiles x 2 phases of flag hand-offs)
    with P.kernel_options(bwd_fuse=2):
        ref2 = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
        assert hip_ops.stats(pb["para_fname"], 0)["persist_steps"] == 0
    for a_, b_ in zip((m, gL, gM, gD, gS), ref2):
        assert torch.equal(a_, b_)
    hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])      # (leave the persistent loop's final state for the field checks below)
    assert float(m) > 0 and all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in (gL, gM, gD))
    worst = {}
    for which, name in enumerate(("vz", "vx", "szz", "sxx", "sxz")):
        f = inner(hip_ops.debug_field(pb["para_fname"], which).numpy())
        worst[name] = float(np.abs(f).max() / (peak_v if which < 2 else peak_s))
    print("reconstructed field at step 0 / peak forward amplitude:", worst)
    assert max(worst.values()) <= 1e-5, worst


@pytest.mark.timeout(900)
def test_config1_shape_2000x500_forward_only(tmp_path, hip_ops, oracle):
    """BASELINE.json configs[1]: 2000 x 500 Marmousi-style model, one shot, 2000 steps, forward only."""
    sys.path.insert(0, ROOT)
    import bench
    from sepfwi import utils as ft
    nS = 2000
    pb = bench.setup_problem(str(tmp_path / "full"), 500, 2000, nS, 3)
    ids = torch.tensor([1], dtype=torch.int32)
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    data_dir = str(tmp_path / "full" / "Data")
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, ids, pb["para_fname"])
    st = hip_ops.stats(pb["para_fname"], 0)
    assert st["fwd_steps"] == nS - 1 and st["bwd_steps"] == 0
    d1 = {c: ft.read_shot_gather(data_dir, c, 1, nS).copy() for c in ("pr", "vx", "vz", "ett")}
    for c, a in d1.items():
        assert a.shape == (pb["nrec"], nS) and np.isfinite(a).all() and np.abs(a).max() > 0, c
        assert np.all(a[:, 0] == 0.0), c                                   # column 0 stays zero (Appendix A-7)
    # round 3: the same run through the CPU oracle at the FULL size (scripts/make_golden_config1shape.py): 32 channels of every
    # component and the norms over all 1980
    import scripts.make_golden_config1shape as mc
    G = np.load(os.path.join(ROOT, "tests", "golden", "oracle_config1shape.npz"))
    assert mc.digest(pb) == str(G["digest"]), "bench.py's problem generator drifted: regenerate with scripts/make_golden_config1shape.py"
    dev = {c: P.rel_l2(d1[c][G["channels"]], G[c]) for c in ("pr", "vx", "vz", "ett")}
    print("configs[1] at full size, HIP vs oracle, rel-L2 per component:", dev)
    assert max(dev.values()) <= 1e-4, dev
    for c in dev:
        assert abs(np.linalg.norm(d1[c].astype(np.float64)) - float(G[c + "_norm"])) <= 1e-4 * float(G[c + "_norm"]), c
    assert np.array_equal(d1["ett"][1:], d1["vx"][1:] - d1["vx"][:-1])       # consecutive channels: exx_r = vx_r - vx_(r-1), exactly
    assert np.abs(d1["vz"][:, nS // 2:]).max() > 0                           # the wavefield is alive in the second half
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, ids, pb["para_fname"])       # bit-identical repeat
    for c in d1:
        assert np.array_equal(ft.read_shot_gather(data_dir, c, 1, nS), d1[c]), c
    hip_ops.obscalc(lt, mt, dt_, 2.0 * pb["Stf"], 1, ids, pb["para_fname"])   # linearity in the source (x2 is exact in float32)
    for c in d1:
        assert P.rel_l2(ft.read_shot_gather(data_dir, c, 1, nS), 2.0 * d1[c]) <= 1e-6, c
    hip_ops.release()

    # cropped twin: the 200 x 100 window of the SAME model under the source, same spacing / time step / wavelet, 700 steps,
    # against the CPU oracle (all four components)
    nz, nx, nPml, nSt = 100, 200, pb["nPml"], 700
    (vp, vs, rho), _ = bench.marmousi_style(500, 2000)
    x0 = 1000 - nx // 2
    win = [a[:nz, x0:x0 + nx] for a in (vp, vs, rho)]
    nPad = ft.nPad_for(nz, nPml)
    work = tmp_path / "crop"
    os.makedirs(work, exist_ok=True)
    para_fname, survey_fname = str(work / "para_file.json"), str(work / "survey_file.json")
    ft.paraGen(nz + 2 * nPml + nPad, nx + 2 * nPml, 10.0, 10.0, nSt, 1.0e-3, 10.0, nPml, nPad, para_fname, survey_fname, str(work / "Data"))
    rec_x = np.arange(10, nx - 10)
    ft.surveyGen(np.array([2]), np.array([nx // 2]), np.full(rec_x.shape, 2), rec_x, survey_fname)
    pad = [torch.tensor(ft.padding_numpy_array(a, nPml, nPad)) for a in win]
    lam = ((pad[0] ** 2 - 2.0 * pad[1] ** 2) * pad[2] / 1e6).contiguous()
    mu = (pad[1] ** 2 * pad[2] / 1e6).contiguous()
    Stf = torch.tensor(ft.sourceGene(10.0, nSt, 1.0e-3), dtype=torch.float32).reshape(1, -1)
    import json
    ref = oracle.cufd(lam.numpy(), mu.numpy(), pad[2].numpy(), Stf.numpy(), 2, [0], json.load(open(para_fname)),
                      json.load(open(survey_fname)))["syn"][0]
    hip_ops.obscalc(lam.cuda(), mu.cuda(), pad[2].contiguous().cuda(), Stf, 1, torch.tensor([0], dtype=torch.int32), para_fname)
    for k, c in enumerate(("pr", "vx", "vz", "ett")):
        got = ft.read_shot_gather(str(work / "Data"), c, 0, nSt)
        assert P.rel_l2(got, ref[k]) <= 1e-4, (c, P.rel_l2(got, ref[k]))


@pytest.mark.timeout(900)
def test_headline_grid_shot_groups_add_up(tmp_path, hip_ops):
    """configs[2] schedules its 32 shots in groups of three forward lanes: at the full grid (fewer steps), the gradient of
    five shots in one call (a full group + a partial one, lanes re-used) equals the sum of the five single-shot gradients,
    and the per-shot source gradients land in their rows."""
    sys.path.insert(0, ROOT)
    import bench
    nS = 300
    pb = bench.setup_problem(str(tmp_path), 1000, 2000, nS, 5)
    ids = torch.arange(5, dtype=torch.int32)
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, ids, pb["para_fname"])
    all5 = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    st = hip_ops.stats(pb["para_fname"], 0)
    assert st["fwd_steps"] == 5 * (nS - 1) and st["bwd_steps"] == 5 * (nS - 1)
    tot = [0.0, 0.0, 0.0, 0.0]
    for k in range(5):
        one = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids[k:k + 1], pb["para_fname"])
        for j in range(4):
            tot[j] = tot[j] + one[j].double()
        assert P.rel_l2(one[4][0].numpy(), all5[4][k].numpy()) <= 1e-6           # gStf row k of the group call
    assert abs(float(all5[0]) - float(tot[0])) <= 1e-5 * float(tot[0])
    for j in (1, 2, 3):
        assert float((all5[j].double() - tot[j]).norm()) <= 1e-5 * float(tot[j].norm())


@pytest.mark.timeout(1100)
def test_headline_32_shot_call_equals_the_sum_of_its_groups(tmp_path, hip_ops):
    """BASELINE.json configs[2] as one operator call: 32 shots, 2000 x 1000, 4000 time steps, forward + adjoint.  The session
    walks them as ten groups of three forward lanes + one of two with 1 GB of observed gathers cached in HBM; the call must
    equal the sum of the same groups issued as eleven separate calls (gradients, misfit) and put every shot's source
    gradient in its own row.  Prints the call's throughput (recorded in profiles/)."""
    sys.path.insert(0, ROOT)
    import time

    import bench
    from sepfwi import utils as ft
    nS, n_shots = 4000, 32
    pb = bench.setup_problem(str(tmp_path), 1000, 2000, nS, n_shots)
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    data_dir = str(tmp_path / "Data")
    for k in range(0, n_shots, 4):                                   # observed data: modelled, handed to the HBM store, files removed
        grp = torch.arange(k, min(k + 4, n_shots), dtype=torch.int32)
        hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, grp, pb["para_fname"])
        for sid in grp.tolist():
            hip_ops.set_observed(pb["para_fname"], sid, torch.from_numpy(ft.read_shot_gather(data_dir, "ett", sid, nS).copy()))
            for c in ("pr", "vx", "vz", "ett"):
                os.remove(os.path.join(data_dir, "Shot_%s%d.bin" % (c, sid)))
    ids = torch.arange(n_shots, dtype=torch.int32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    all32 = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = hip_ops.stats(pb["para_fname"], 0)
    assert st["fwd_steps"] == n_shots * (nS - 1) and st["bwd_steps"] == n_shots * (nS - 1)
    rate = 3.0 * pb["n_c"] * (nS - 1) * n_shots / el / 1e9
    print("32-shot call: %.2f s wall, %.2f Gcell-updates/s (fwd %.1f ms, bwd %.1f ms per shot; %.2f GB held by the session)"
          % (el, rate, st["fwd_ms"] / n_shots, st["bwd_ms"] / n_shots, st["device_bytes"] / 1e9))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "call32.txt"), "w") as fp:
        fp.write("32 shots, 2000x1000x4000 fwd+adj in ONE fwi_ops.backward call: %.3f s wall, %.3f Gcell-updates/s, fwd %.2f ms/shot, "
                 "bwd %.2f ms/shot, session holds %.2f GB\n" % (el, rate, st["fwd_ms"] / n_shots, st["bwd_ms"] / n_shots, st["device_bytes"] / 1e9))
    assert rate > 60.0, rate                                          # a 32-shot call must not fall off the three-shot bench line
    tot = [0.0, 0.0, 0.0, 0.0]
    for k in range(0, n_shots, 3):
        one = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids[k:k + 3], pb["para_fname"])
        for j in range(4):
            tot[j] = tot[j] + one[j].double()
        assert torch.equal(one[4][: min(3, n_shots - k)], all32[4][k:k + 3])      # gStf rows of the group, bit for bit
    assert abs(float(all32[0]) - float(tot[0])) <= 1e-5 * float(tot[0])
    # one call accumulates 32 x 3999 imaging terms per cell in float32, the eleven calls 3 x 3999 each and are summed in
    # double here: the two differ by the round-off of the longer float32 sum (measured 4e-6 ... 1e-5), not by a group
    dev = [float((all32[j].double() - tot[j]).norm()) / float(tot[j].norm()) for j in (1, 2, 3)]
    print("32-shot call vs the sum of its groups, rel-L2 of gLambda, gMu, gDen:", dev)
    assert max(dev) <= 5e-5, dev


HEADLINE_GOLDEN = os.path.join(ROOT, "tests", "golden", "oracle_headline.npz")


@pytest.mark.timeout(600)
def test_headline_full_size_matches_oracle(tmp_path, hip_ops):
    """BASELINE.json configs[2] ("grad checked vs reference") at its REAL size: one shot of bench.py's 2000 x 1000 problem,
    4000 time steps, 1980 DAS channels, forward + boundary-saving adjoint, against the CPU oracle's run of the same inputs
    (scripts/make_golden_headline.py: 35e9 cell-updates on the host, decimated to tests/golden/oracle_headline.npz).
    The gradient call uses the observed gather this library modelled itself -- the oracle's full gather (32 MB) is not
    committed; 64 of its channels are, and are compared first.  Tolerances are SURVEY.md 8c's."""
    sys.path.insert(0, ROOT)
    import bench
    import scripts.make_golden_headline as mg
    from sepfwi import utils as ft
    G = np.load(HEADLINE_GOLDEN)
    nS = mg.NSTEPS
    pb = bench.setup_problem(str(tmp_path), mg.NZ, mg.NX, nS, mg.NSHOTS)
    assert mg.digest(pb) == str(G["digest"]), "bench.py's problem generator drifted: regenerate with scripts/make_golden_headline.py"
    ids = torch.tensor([mg.SHOT], dtype=torch.int32)
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    data_dir = str(tmp_path / "Data")
    ch, ch8 = G["channels"], G["channels_other"]
    # observed data of the "true" model: all four components on the committed channels
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, ids, pb["para_fname"])
    obs = {c: ft.read_shot_gather(data_dir, c, mg.SHOT, nS) for c in ("pr", "vx", "vz", "ett")}
    assert obs["ett"].shape == (pb["nrec"], nS)
    dev = {"ett": P.rel_l2(obs["ett"][ch], G["obs_ett"])}
    for c in ("pr", "vx", "vz"):
        dev[c] = P.rel_l2(obs[c][ch8], G["obs_" + c])
    print("observed gather vs oracle (rel-L2 on the committed channels):", dev)
    assert max(dev.values()) <= 1e-4, dev
    assert abs(np.linalg.norm(obs["ett"].astype(np.float64)) - float(G["obs_ett_norm"])) <= 1e-4 * float(G["obs_ett_norm"])   # ALL 1980 channels
    late = slice(3 * nS // 4, nS)                                             # the last 1000 steps on their own
    assert P.rel_l2(obs["ett"][ch][:, late], G["obs_ett"][:, late]) <= 1e-4
    # gradient of the initial model
    m, gL, gM, gD, gS = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    assert hip_ops.stats(pb["para_fname"], 0)["persist_steps"] == nS - 1        # at this size the backward pass IS the persistent loop (default)
    e_m = abs(float(m) - float(G["misfit"])) / float(G["misfit"])
    out = {"misfit": e_m}
    assert e_m <= 1e-4, (float(m), float(G["misfit"]))
    d = int(G["decim"])
    z0, z1, x0, x1 = [int(v) for v in G["win"]]
    for key, g in (("gLambda", gL), ("gMu", gM), ("gDen", gD)):
        g = g.cpu().numpy()
        gmax = float(G[key + "_max"])
        dec, win = g[::d, ::d], g[z0:z1, x0:x1]
        out[key] = (P.rel_l2(dec, G[key + "_dec"]), P.rel_l2(win, G[key + "_win"]),
                    abs(np.linalg.norm(g.astype(np.float64)) - float(G[key + "_norm"])) / float(G[key + "_norm"]))
        assert out[key][0] <= 1e-3 and out[key][1] <= 1e-3 and out[key][2] <= 1e-3, (key, out[key])
        assert np.abs(dec - G[key + "_dec"]).max() <= 1e-3 * gmax and np.abs(win - G[key + "_win"]).max() <= 1e-3 * gmax, key
        assert abs(float(np.abs(g).max()) - gmax) <= 1e-3 * gmax, key
    out["gStf"] = P.rel_l2(gS.numpy()[0], G["gStf"])
    assert out["gStf"] <= 1e-3, out
    print("gradient call vs oracle: misfit rel, (rel-L2 every 8th cell, rel-L2 96x96 window under the source, rel norm) per gradient, gStf:", out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "headline_vs_oracle.txt"), "w") as fp:
        fp.write("2000x1000x4000, one shot, HIP path vs CPU oracle (tests/golden/oracle_headline.npz)\nobserved gather rel-L2: %r\ngradient call: %r\n" % (dev, out))


HEADLINE32_GOLDEN = os.path.join(ROOT, "tests", "golden", "oracle_headline32.npz")


@pytest.mark.timeout(900)
def test_headline_32_shots_match_oracle(tmp_path, hip_ops):
    """BASELINE.json configs[2] LITERALLY: "2000 x 1000 model, 32 shots, forward + boundary-saving adjoint gradient on 1 MI355X,
    grad checked vs reference" -- ONE 32-shot call of the HIP path against the CPU oracle's 32 shots (0.9e12 cell-updates on
    the host, scripts/make_golden_headline32.py; per-shot float32 gradients summed in float64).  Observed data: modelled by
    this library (the oracle's 1 GB of gathers are not committed; the single-shot test compares them channel by channel)."""
    golden = os.environ.get("SEPFWI_GOLDEN32_PARTIAL", HEADLINE32_GOLDEN)    # a partial file (first N shots) for a dry run
    if not os.path.exists(golden):
        pytest.skip("tests/golden/oracle_headline32.npz not generated (scripts/make_golden_headline32.py, about 3 hours of CPU)")
    sys.path.insert(0, ROOT)
    import bench
    import scripts.make_golden_headline32 as mg
    from sepfwi import utils as ft
    G = np.load(golden)
    nS, n_shots = mg.NSTEPS, int(G["n_shots"])
    assert n_shots == mg.NSHOTS or golden != HEADLINE32_GOLDEN
    pb = bench.setup_problem(str(tmp_path), mg.NZ, mg.NX, nS, mg.NSHOTS)
    assert mg.digest(pb) == str(G["digest"]), "bench.py's problem generator drifted: regenerate with scripts/make_golden_headline32.py"
    lt, mt, dt_ = [t.cuda() for t in pb["lame_true"]]
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    data_dir = str(tmp_path / "Data")
    for k in range(0, n_shots, 4):
        grp = torch.arange(k, min(k + 4, n_shots), dtype=torch.int32)
        hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, grp, pb["para_fname"])
        for sid in grp.tolist():
            hip_ops.set_observed(pb["para_fname"], sid, torch.from_numpy(ft.read_shot_gather(data_dir, "ett", sid, nS).copy()))
            for c in ("pr", "vx", "vz", "ett"):
                os.remove(os.path.join(data_dir, "Shot_%s%d.bin" % (c, sid)))
    ids = torch.arange(n_shots, dtype=torch.int32)
    m, gL, gM, gD, gS = hip_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    assert hip_ops.stats(pb["para_fname"], 0)["persist_steps"] == n_shots * (nS - 1)   # every one of the 32 backward passes in the persistent loop
    out = {"misfit": abs(float(m) - float(G["misfit"])) / float(G["misfit"])}
    assert out["misfit"] <= 1e-4, (float(m), float(G["misfit"]))
    d = int(G["decim"])
    z0, z1, x0, x1 = [int(v) for v in G["win"]]
    for key, g in (("gLambda", gL), ("gMu", gM), ("gDen", gD)):
        g = g.cpu().numpy()
        gmax = float(G[key + "_max"])
        dec, win = g[::d, ::d], g[z0:z1, x0:x1]
        out[key] = (P.rel_l2(dec, G[key + "_dec"]), P.rel_l2(win, G[key + "_win"]),
                    abs(np.linalg.norm(g.astype(np.float64)) - float(G[key + "_norm"])) / float(G[key + "_norm"]))
        assert max(out[key]) <= 1e-3, (key, out[key])
        assert np.abs(dec - G[key + "_dec"]).max() <= 1e-3 * gmax and np.abs(win - G[key + "_win"]).max() <= 1e-3 * gmax, key
    out["gStf"] = P.rel_l2(gS.numpy()[:n_shots], G["gStf"])
    assert out["gStf"] <= 1e-3, out
    print("32-shot call vs the oracle's 32 shots: misfit rel, (rel-L2 every 8th cell, window, rel norm) per gradient, gStf:", out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "headline32_vs_oracle.txt"), "w") as fp:
        fp.write("2000x1000x4000, %d shots in one call, HIP path vs CPU oracle (%s): %r\n" % (n_shots, os.path.basename(golden), out))


@pytest.mark.timeout(900)
def test_grid_seven_times_the_headline_matches_oracle(tmp_path, oracle, hip_ops):
    """5600 x 2800 cells (16.6 M with the layers, 7.5 x the headline grid: 66 MB per array, tile and frame indices far beyond
    anything else in the suite), 16 time steps, two shots near the middle, forward + adjoint against the CPU oracle: the launch
    geometry, XCD banding and 64-bit offsets at a size where a 32-bit slip or a mis-banded tile would show."""
    from sepfwi import utils as ft
    nz, nx, nS = 2800, 5600, 16
    stf = ft.sourceGene(25.0, 64, 1e-3)[34:34 + nS]          # the wavelet's main lobe inside the 16 steps
    pb = P.make_problem(str(tmp_path), nz=nz, nx=nx, nPml=32, nSteps=nS, nshots=2, hetero=True, seed=5, src_z=1400,
                        src_x=[2790, 2830], rec_z=1404, rec_x=list(range(2760, 2860)), stf=stf)
    lt, mt, dt_ = pb["lame_true"]
    # the trial medium differs from the observed one AT the source (the anomalies of make_problem are out of reach in 16 steps)
    lam, mu, den = [(t * f).contiguous() for t, f in zip(pb["lame_init"], (0.92, 0.95, 1.03))]
    obs = oracle.cufd(lt.numpy(), mt.numpy(), dt_.numpy(), pb["Stf"].numpy(), 2, pb["Shot_ids"].numpy(), pb["para"], pb["survey"])["syn"]
    ref = oracle.cufd(lam.numpy(), mu.numpy(), den.numpy(), pb["Stf"].numpy(), 1, pb["Shot_ids"].numpy(), pb["para"], pb["survey"], obs=obs)
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    for i in range(2):
        got = ft.read_shot_gather(pb["data_dir"], "ett", i, nS)
        assert np.abs(obs[i, 3]).max() > 0 and P.rel_l2(got, obs[i, 3]) <= 1e-4, (i, P.rel_l2(got, obs[i, 3]))
        obs[i, 3].tofile(os.path.join(pb["data_dir"], "Shot_ett%d.bin" % i))
    hip_ops.release()
    m, gL, gM, gD, gS = hip_ops.backward(lam.cuda(), mu.cuda(), den.cuda(), pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    assert abs(float(m) - ref["misfit"]) <= 1e-4 * abs(ref["misfit"]) and ref["misfit"] > 0
    lines = []
    for name, g, r in (("gLambda", gL, ref["gLambda"]), ("gMu", gM, ref["gMu"]), ("gDen", gD, ref["gDen"])):
        g = g.cpu().numpy()
        e, emax = P.rel_l2(g, r), np.abs(g - r).max() / np.abs(r).max()
        lines.append("%s rel-L2 %.2e max %.2e nonzero cells %d" % (name, e, emax, np.count_nonzero(r)))
        assert np.abs(r).max() > 0 and e <= 1e-3 and emax <= 1e-3, lines[-1]
        odd = (g != 0) != (r != 0)      # the fringe of the stencil's light cone: values that underflow on one side only
        assert np.abs(g[odd]).max(initial=0.0) <= 1e-12 * np.abs(r).max() and np.abs(r[odd]).max(initial=0.0) <= 1e-12 * np.abs(r).max(), name
        far = np.ones(r.shape, bool)
        far[1400 + 32 - 80:1400 + 32 + 80, 2760 + 32 - 80:2860 + 32 + 80] = False
        assert not g[far].any(), name   # and nothing anywhere else on the 16.3 M cells
    assert P.rel_l2(gS.numpy()[:2], ref["gStf"]) <= 1e-3
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "big_grid_vs_oracle.txt"), "w") as fp:
        fp.write("5600 x 2800 x 16 steps, 2 shots, HIP vs CPU oracle: misfit %.6e vs %.6e\n" % (float(m), ref["misfit"]) + "\n".join(lines) + "\n")
