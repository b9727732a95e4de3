#!/usr/bin/env python
"""Stress of the persistent backward loop's hand-offs at the headline size: R repetitions of an S-shot gradient call through the loop
(3999 time steps x 512 tiles x 2 phases of flag hand-offs per shot) against ONE evaluation with the two-launch step -- every bit of
misfit, the three gradients and the source gradients must agree every time.     python scripts/persist_stress.py [--shots 6] [--reps 4]"""
import argparse
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch

import bench
from sepfwi import _native, fwi_ops

ap = argparse.ArgumentParser()
ap.add_argument("--shots", type=int, default=6)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--nsteps", type=int, default=4000)
a = ap.parse_args()
L = _native.lib()
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="sepfwi_stress_")
try:
    pb = bench.setup_problem(work, 1000, 2000, a.nsteps, a.shots)
    lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
    lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
    ids = torch.arange(a.shots, dtype=torch.int32)
    fwi_ops._cufd(3, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
    _native.check(L.sepfwi_set_option(b"bwd_fuse", 2))
    ref = [t.clone() for t in fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])]
    assert fwi_ops.stats(pb["para_fname"], 0)["persist_steps"] == 0
    _native.check(L.sepfwi_set_option(b"bwd_fuse", 4))
    bad = 0
    for r in range(a.reps):
        got = fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
        st = fwi_ops.stats(pb["para_fname"], 0)
        same = [bool(torch.equal(x, y)) for x, y in zip(got, ref)]
        bad += not all(same)
        print("repetition %d: persist_steps %d of %d, bit-identical (misfit, gLambda, gMu, gDen, gStf): %s, %.2f us per backward time step"
              % (r, st["persist_steps"], a.shots * (a.nsteps - 1), same, st["bwd_ms"] * 1e3 / st["bwd_steps"]), flush=True)
        assert st["persist_steps"] == a.shots * (a.nsteps - 1)
    print("misfit %.6e, |gLambda|_max %.3e; %d of %d repetitions differ" % (float(ref[0]), float(ref[1].abs().max()), bad, a.reps))
    sys.exit(1 if bad else 0)
finally:
    shutil.rmtree(work, ignore_errors=True)
