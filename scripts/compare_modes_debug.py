#!/usr/bin/env python
"""Debug aid for tests/test_gpu_fuzz.py: rebuilds the random problem of one seed (line-receiver cases), prints GPU-vs-oracle and
GPU-vs-GPU (stream / batched / unfused kernels) deviations of misfit, gradients and observed gathers and where they sit.
usage: python scripts/fuzz_debug.py <seed>      (needs a GPU)"""
import json, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import problems as P
from oracle import oracle as O
from sepfwi import fwi_ops, _native
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
nPml = int(rng.integers(4, 13)); nz, nx = int(rng.integers(24, 60)), int(rng.integers(30, 100)); nPad = int(rng.integers(0, 9))
nSteps = int(rng.integers(90, 200)); nshots = int(rng.integers(1, 5))
tmp = tempfile.mkdtemp()
pb = P.make_problem(tmp, nz=nz, nx=nx, nPml=nPml, nSteps=nSteps, nshots=nshots, nPad=nPad, hetero=True, seed=seed, src_z=int(rng.integers(1, 5)), rec_z=int(rng.integers(2, nz - 3)))
sv = json.load(open(pb["survey_fname"]))
kind = int(rng.integers(0, 3))
print("nz nx nPml nPad nSteps nshots kind", nz, nx, nPml, nPad, nSteps, nshots, kind, "src_z", sv["shot0"]["z_src"], "rec_z", sv["shot0"]["z_rec"][0])
assert kind == 0
lam_t, mu_t, den_t = pb["lame_true"]; ids = pb["Shot_ids"].numpy()
obs = O.cufd(lam_t.numpy(), mu_t.numpy(), den_t.numpy(), pb["Stf"].numpy(), 2, ids, pb["para"], sv)["syn"]
os.makedirs(pb["data_dir"], exist_ok=True)
for i, sid in enumerate(ids.tolist()):
    for k, c in enumerate(("pr", "vx", "vz", "ett")):
        obs[i, k].tofile(os.path.join(pb["data_dir"], "Shot_%s%d.bin" % (c, sid)))
lam, mu, den = pb["lame_init"]
ref = O.cufd(lam.numpy(), mu.numpy(), den.numpy(), pb["Stf"].numpy(), 1, ids, pb["para"], sv, obs=obs)
outs = {}
for name, opts in (("default", {}), ("batch0", {"batch": 0}), ("unfused", {"bwd_fuse": 0, "line_fuse": 0, "batch": 0, "pair_fwd": 0})):
    for k, v in opts.items(): _native.check(_native.lib().sepfwi_set_option(k.encode(), v))
    m, gL, gM, gD, gS = fwi_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    outs[name] = (float(m), gL.numpy(), gM.numpy(), gD.numpy())
    for k, v in dict(bwd_fuse=2, line_fuse=1, batch=2, pair_fwd=1).items(): _native.lib().sepfwi_set_option(k.encode(), v)
    print(name, "misfit rel", abs(float(m) - ref["misfit"]) / ref["misfit"], "gL %.3e gM %.3e gD %.3e" % tuple(P.rel_l2(a, ref[k]) for a, k in zip(outs[name][1:], ("gLambda", "gMu", "gDen"))))
d = outs["default"][1] - ref["gLambda"]
iz, ix = np.unravel_index(np.abs(d).argmax(), d.shape)
print("max |dgL| %.3e at (z=%d,x=%d) of max|gL| %.3e; interior rows %d..%d cols %d..%d" % (np.abs(d).max(), iz, ix, np.abs(ref["gLambda"]).max(), nPml, nPml + nz - 1, nPml, nPml + nx - 1))
rows = np.sqrt((d ** 2).sum(1)); print("row error profile (top 8):", np.argsort(rows)[-8:][::-1], np.sort(rows)[-8:][::-1] / np.linalg.norm(ref["gLambda"]))
print("gpu default vs gpu unfused gL:", P.rel_l2(outs["default"][1], outs["unfused"][1]))
print("misfits:", {k: v[0] for k, v in outs.items()})
print("gpu default vs gpu batch0 gL: %.3e  gM %.3e gD %.3e" % tuple(P.rel_l2(outs["default"][i], outs["batch0"][i]) for i in (1, 2, 3)))
print("gpu batch0 vs gpu unfused gL: %.3e" % P.rel_l2(outs["batch0"][1], outs["unfused"][1]))
dd = outs["default"][1] - outs["batch0"][1]
rows = np.sqrt((dd ** 2).sum(1)); print("default-batch0 row error profile:", np.argsort(rows)[-5:][::-1], np.sort(rows)[-5:][::-1] / np.linalg.norm(ref["gLambda"]))
# bisect: forward-only pieces
from sepfwi import utils as ft
res = {}
for name, opts in (("batch1", {"batch": 1}), ("batch0", {"batch": 0})):
    for k, v in opts.items(): _native.check(_native.lib().sepfwi_set_option(k.encode(), v))
    m0 = float(fwi_ops.forward(lam, mu, den, pb["Stf"], 0, pb["Shot_ids"], pb["para_fname"])[0])
    res[name] = m0
    _native.lib().sepfwi_set_option(b"batch", 2)
print("misfit-only (calc 0):", res, "rel diff %.3e" % (abs(res["batch1"] - res["batch0"]) / res["batch0"]))
gat = {}
for name, opts in (("batch1", {"batch": 1}), ("batch0", {"batch": 0})):
    for k, v in opts.items(): _native.check(_native.lib().sepfwi_set_option(k.encode(), v))
    fwi_ops.obscalc(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    gat[name] = [ft.read_shot_gather(pb["data_dir"], c, 0, nSteps).copy() for c in ("pr", "vx", "vz", "ett")]
    _native.lib().sepfwi_set_option(b"batch", 2)
for k, c in enumerate(("pr", "vx", "vz", "ett")):
    a, b = gat["batch1"][k], gat["batch0"][k]
    print("observe %s: rel diff %.3e  first differing column %s" % (c, P.rel_l2(a, b), (np.nonzero(np.abs(a - b).max(0))[0][:3])))
