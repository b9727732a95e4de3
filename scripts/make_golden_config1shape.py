#!/usr/bin/env python
"""tests/golden/oracle_config1shape.npz: BASELINE.json configs[1] at its REAL size -- the 2000 x 500 Marmousi-style model of
bench.py (padded 2064 x 576), one shot, 2000 time steps, forward only -- through the CPU oracle (about 1.5 minutes on two
cores).  Stored: 32 of the 1980 channels of all four components, the norms over all channels, a digest of the inputs.
tests/test_gpu_headline.py::test_config1_shape_2000x500_forward_only compares the HIP path with it."""
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

NZ, NX, NSTEPS, NSHOTS, SHOT = 500, 2000, 2000, 3, 1
CHANNELS = np.arange(12, 1980, 62)[:32]


def digest(pb):
    h = hashlib.sha256()
    for t in list(pb["lame_true"]) + [pb["Stf"]]:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def main():
    O.build()
    with tempfile.TemporaryDirectory() as d:
        pb = bench.setup_problem(d, NZ, NX, NSTEPS, NSHOTS)
        para, survey = json.load(open(pb["para_fname"])), json.load(open(os.path.join(d, "survey_file.json")))
        lam, mu, den = [t.numpy() for t in pb["lame_true"]]
        t0 = time.time()
        syn = O.cufd(lam, mu, den, pb["Stf"].numpy(), 2, [SHOT], para, survey)["syn"][0]
        print("forward: %.1f s" % (time.time() - t0), flush=True)
        out = dict(channels=CHANNELS, digest=digest(pb))
        for k, c in enumerate(("pr", "vx", "vz", "ett")):
            out[c] = syn[k][CHANNELS]
            out[c + "_norm"] = np.float64(np.linalg.norm(syn[k].astype(np.float64)))
        fn = os.path.join(ROOT, "tests", "golden", "oracle_config1shape.npz")
        np.savez_compressed(fn, **out)
        print("wrote", fn, os.path.getsize(fn), "bytes")


if __name__ == "__main__":
    main()
