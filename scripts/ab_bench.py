#!/usr/bin/env python
"""A/B timing of kernel options in ONE process on one GPU (interleaved rounds, cdna guide rule 24).
usage: python scripts/ab_bench.py --nsteps 400 --rounds 3 "bz=4,xcd_remap=0" "bz=4,xcd_remap=1" ...
Prints per-variant forward / backward microseconds per time step (HIP events on the session stream)."""
import argparse
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import numpy as np
import torch

import bench
from sepfwi import _native, fwi_ops


DEFAULTS = dict(bz=2, xcd_remap=1, bwd_fuse=4, line_fuse=1, pair_fwd=1, fwd_lanes=3, early=0, rho_fly=1, amu_fly=1, rk_lazy=1, batch=2, batch_f=0,
                batch_b=0, batch_mb=200, batch_order=1, batch_split=2, img_every=1, obs_cache_mb=0, quiet_skip=0, quiet_rows=4, pk_lmask=16, pk_wpc=2, pk_px=3, pk_nosync=0, pk_lock=0, pk_snake=1, pk_ms=0, pk_quiet=0, pk_waves=16, pk_order=2, pk_prio=1, pk_wx=150, pk_wxp=150, pk_wz=115)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--nsteps", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--nz", type=int, default=1000)
    ap.add_argument("--nx", type=int, default=2000)
    ap.add_argument("--shots", type=int, default=3)
    ap.add_argument("--rec-stride", type=int, default=1, help="a channel every N cells (N > 1: not a fused line -- k_inject, or the loop's general injection)")
    a = ap.parse_args()
    _native._active = "probes"   # the tuning knobs exist only in the -DSEPFWI_PROBES build of the library
    _native.build(variant="probes")
    L = _native.lib()
    dev = torch.device("cuda", 0)
    work = tempfile.mkdtemp(prefix="sepfwi_ab_")
    try:
        pb = bench.setup_problem(work, a.nz, a.nx, a.nsteps, a.shots, rec_stride=a.rec_stride)
        lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
        lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
        ids = torch.arange(a.shots, dtype=torch.int32)
        fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
        res = {v: [] for v in a.variants}
        ref = None
        for r in range(a.rounds + 1):
            for v in a.variants:
                for k, val in DEFAULTS.items():   # options are sticky in the library: start every variant from the defaults
                    _native.check(L.sepfwi_set_option(k.encode(), val))
                for kv in v.split(","):
                    if not kv:
                        continue
                    k, val = kv.split("=")
                    _native.check(L.sepfwi_set_option(k.encode(), int(val)))
                try:
                    out = fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
                except RuntimeError as e:   # a variant the library refuses
                    if r == 0:
                        print("variant %s failed: %s" % (v, str(e)[:200]))
                    continue
                st = fwi_ops.stats(pb["para_fname"], 0)
                if r > 0:
                    res[v].append((st["fwd_ms"] * 1e3 / st["fwd_steps"], st["bwd_ms"] * 1e3 / st["bwd_steps"]))
                sig = (float(out[0]), float(out[1].abs().sum()), float(out[3].abs().sum()))
                if ref is None:
                    ref = sig
                dev_ = max(abs(x - y) / max(abs(y), 1e-30) for x, y in zip(sig, ref))
                if dev_ > 1e-5:
                    print("WARNING variant %s result deviates from first variant by %.2e" % (v, dev_))
        nc = pb["n_c"]
        for v in a.variants:
            if not res[v]:
                continue
            f = np.array([x[0] for x in res[v]]); b = np.array([x[1] for x in res[v]])
            tot = np.median(f) + np.median(b)
            print("%-40s fwd %7.2f us (min %7.2f)  bwd %7.2f us (min %7.2f)  -> %6.2f Gcell-updates/s" %
                  (v, np.median(f), f.min(), np.median(b), b.min(), 3 * nc / tot / 1e3))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
