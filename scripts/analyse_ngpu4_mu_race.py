#!/usr/bin/env python
"""Two last cheap hypotheses for the gap between the reference's printed iterate-0 |proj g| (notebooks 001-003, run with --ngpu 4)
and the oracle's (+0.07 % / +1.3 % / +0.3 %), VERDICT round 4 task 6:
  (a) the printed runs split the 19 shots over four GPUs as [0, 4, 9, 14, 19] (Src/Torch_Fwi.cpp:59-60,78-80); each GPU accumulates
      its block in float32 and the host adds the four blocks in order (:96-101) -- another association of the same float32 sum;
  (b) the mu image's plain `+=` against the neighbours' atomicAdd sprays (Src/el_stress.cu:110,116-122): lost updates, here at
      their largest (every neighbour spray dropped, ofwi_set_debug_mu_lost).
    python scripts/analyse_ngpu4_mu_race.py        (about 7 min on 8 cores)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import experiments as E
import oracle_backend
import sepfwi.ops as ops
from oracle import oracle as O

O.build(force=True)


class SplitOps(oracle_backend.OracleOps):
    """`backward` as the reference's fwi_backward does it with ngpu = 4: block-wise calls, float32 sums in block order."""

    def backward(self, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname):
        ids = self._np(Shot_ids)
        bars = np.linspace(0, ids.size, 5, dtype=np.float32).astype(np.int32)        # [0, 4, 9, 14, 19] for 19 shots
        tot = None
        for i in range(4):
            out = super().backward(Lambda, Mu, Den, Stf, 1, ids[bars[i]:bars[i + 1]], para_fname)
            tot = out if tot is None else [tot[0] + out[0], tot[1] + out[1], tot[2] + out[2], tot[3] + out[3], tot[4]]
        return tot


for label, mk, lost in (("(a) four blocks [0,4,9,14,19], float32 block sums", SplitOps, 0), ("(b) every neighbour spray of the mu image lost", oracle_backend.OracleOps, 1)):
    ops.fwi_ops = mk()
    O.lib().ofwi_set_debug_mu_lost(lost)
    print(label)
    for exp in ("001", "002", "003"):
        k = E.KNOWN[exp]
        with tempfile.TemporaryDirectory() as d:
            r = E.run_iterate0(exp, d)
        per = {n: float(np.abs(a).max()) for n, a in r["grads"].items()}
        print("  exp %s: f %.6e (printed %.6e)  |g|inf %.6f (printed %.5f, dev %+.3e; one block, no loss: %.6f)  per parameter %s" % (
            exp, r["f"], k["f"], r["ginf"], k["ginf"], (r["ginf"] - k["ginf"]) / k["ginf"], k["ginf_oracle"],
            " ".join("%s %.5f" % kv for kv in per.items())))
        sys.stdout.flush()
O.lib().ofwi_set_debug_mu_lost(0)
