#!/usr/bin/env python
"""How much of the headline grid holds a non-zero value after n time steps (option quiet_skip's map of the forward stresses),
and what the option buys at that record length: python scripts/quiet_coverage.py 500 1000 2000 3000 4000"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch

import bench
from sepfwi import _native, fwi_ops


def main():
    L = _native.lib()
    for kv in os.environ.get("QC_OPTS", "").split(","):      # e.g. QC_OPTS=pair_fwd=0,batch=0
        if kv:
            _native.check(L.sepfwi_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
    dev = torch.device("cuda", 0)
    for nsteps in [int(a) for a in sys.argv[1:]] or [1000, 4000]:
        work = tempfile.mkdtemp(prefix="sepfwi_qc_")
        try:
            pb = bench.setup_problem(work, 1000, 2000, nsteps, 3)
            lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
            lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
            ids = torch.arange(3, dtype=torch.int32)
            fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
            row = []
            for q in (0, 1):
                _native.check(L.sepfwi_set_option(b"quiet_skip", q))
                for rep in range(2):
                    fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
                st = fwi_ops.stats(pb["para_fname"], 0)
                row.append((1e3 * st["fwd_ms"] / st["fwd_steps"], 1e3 * st["bwd_ms"] / st["bwd_steps"], st["quiet_active"], st["quiet_total"]))
            _native.check(L.sepfwi_set_option(b"quiet_skip", 0))
            print("nSteps %5d: fwd %6.2f -> %6.2f us, bwd %6.2f -> %6.2f us per step and shot; segments that ever held a value: %d of %d (%.0f %%)"
                  % (nsteps, row[0][0], row[1][0], row[0][1], row[1][1], row[1][2], row[1][3], 100.0 * row[1][2] / max(1, row[1][3])), flush=True)
            fwi_ops.release()
        finally:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
