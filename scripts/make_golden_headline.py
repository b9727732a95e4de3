#!/usr/bin/env python
"""tests/golden/oracle_headline.npz: ONE shot of BASELINE.json configs[2] at its REAL size -- the 2000 x 1000 model of
bench.py (padded 2064 x 1088), 4000 time steps, the 1980-channel DAS line, forward + boundary-saving adjoint -- through
the CPU oracle (oracle/torchfwi_oracle.c, the float32 restatement of the reference's cufd).  "grad checked vs reference"
at the size the metric is quoted on; tests/test_gpu_headline.py::test_headline_full_size_matches_oracle compares the HIP
path with it.

About 35e9 cell-updates: the oracle's forward and adjoint loop nests share their columns between OpenMP threads (bit-identical
to the serial loops; the imaging loops stay serial), about 25 minutes on 8 cores, 3 GB of memory.

    python scripts/make_golden_headline.py            # the golden file
    python scripts/make_golden_headline.py --nsteps 60 --out /tmp/x.npz   # a short calibration run (not a golden)

Stored (decimated, < 2 MB): 64 channels of the observed ("true" model) and synthetic (initial model) axial-strain gathers,
8 channels of the other three components of the observed gather, misfit, every 8th cell of the three gradients plus one
full-resolution 96 x 96 window of each under the source (the gradient sprays are one-cell features), the source-function
gradient, peak amplitudes, and a digest of the inputs so a drift of bench.py's problem generator is detected."""
import argparse
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

NZ, NX, NSTEPS, NSHOTS, SHOT = 1000, 2000, 4000, 3, 1      # bench.setup_problem(..., 3 shots): shot 1 sits mid-line
CHANNELS = np.arange(8, 1980, 31)[:64]                      # 64 of the 1980 channels
CHANNELS_OTHER = CHANNELS[::8]                              # 8 channels of pr, vx, vz
DECIM = 8
WIN = (slice(32, 128), slice(984, 1080))                    # padded-grid window under the source (z, x), full resolution


def digest(pb):
    h = hashlib.sha256()
    for t in list(pb["lame_true"]) + list(pb["lame_init"]) + [pb["Stf"]]:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsteps", type=int, default=NSTEPS)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "oracle_headline.npz"))
    args = ap.parse_args()
    if args.nsteps != NSTEPS and args.out.startswith(os.path.join(ROOT, "tests", "golden")):
        raise SystemExit("a shortened run is not the golden file: give --out somewhere else")
    O.build()
    import json
    with tempfile.TemporaryDirectory() as d:
        pb = bench.setup_problem(d, NZ, NX, args.nsteps, NSHOTS)
        para, survey = json.load(open(pb["para_fname"])), json.load(open(os.path.join(d, "survey_file.json")))
        stf = pb["Stf"].numpy()
        ids = [SHOT]
        t0 = time.time()
        lam, mu, den = [t.numpy() for t in pb["lame_true"]]
        obs = O.cufd(lam, mu, den, stf, 2, ids, para, survey)["syn"]
        print("observe: %.1f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        lam, mu, den = [t.numpy() for t in pb["lame_init"]]
        ref = O.cufd(lam, mu, den, stf, 1, ids, para, survey, obs=obs)
        print("gradient: %.1f s, misfit %.6e" % (time.time() - t0, ref["misfit"]), flush=True)
        g = {k: ref[k] for k in ("gLambda", "gMu", "gDen")}
        out = dict(misfit=np.float64(ref["misfit"]), gStf=ref["gStf"][0], digest=digest(pb), channels=CHANNELS,
                   channels_other=CHANNELS_OTHER, decim=DECIM, win=np.array([WIN[0].start, WIN[0].stop, WIN[1].start, WIN[1].stop]),
                   obs_ett=obs[0, 3][CHANNELS], syn_ett=ref["syn"][0, 3][CHANNELS],
                   obs_pr=obs[0, 0][CHANNELS_OTHER], obs_vx=obs[0, 1][CHANNELS_OTHER], obs_vz=obs[0, 2][CHANNELS_OTHER],
                   obs_peak=np.array([np.abs(obs[0, k]).max() for k in range(4)]),
                   obs_ett_norm=np.float64(np.linalg.norm(obs[0, 3].astype(np.float64))),
                   syn_ett_norm=np.float64(np.linalg.norm(ref["syn"][0, 3].astype(np.float64))))
        for k, a in g.items():
            out[k + "_dec"] = np.ascontiguousarray(a[::DECIM, ::DECIM])
            out[k + "_win"] = np.ascontiguousarray(a[WIN])
            out[k + "_norm"] = np.float64(np.linalg.norm(a.astype(np.float64)))
            out[k + "_max"] = np.float64(np.abs(a).max())
        np.savez_compressed(args.out, **out)
        print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
