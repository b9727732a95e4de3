#!/usr/bin/env python
"""Would overlapping the backward passes of one shot group with the three-lane forward passes of the next group pay?
Two sessions (own state, own streams) on one GPU, each evaluating gradients of 3 shots in a loop, the second thread
started when the first one's forward phase is over, so that forward and backward phases of the two overlap most of the time.
Prints the aggregate rate next to the rate of the same calls one after the other.
    python scripts/overlap_probe.py [nsteps] [calls]"""
import os, shutil, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch
import bench
from sepfwi import fwi_ops

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
works = [tempfile.mkdtemp(prefix="sepfwi_ov%d_" % i) for i in range(2)]
try:
    pbs = [bench.setup_problem(w, 1000, 2000, nsteps, 3) for w in works]
    ids = torch.arange(3, dtype=torch.int32)
    ins = []
    for pb in pbs:
        lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
        fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
        ins.append([t.to(dev) for t in pb["lame_init"]])

    def run(i, n, delay=0.0):
        time.sleep(delay)
        lam, mu, den = ins[i]
        for _ in range(n):
            fwi_ops.backward(lam, mu, den, pbs[i]["Stf"], 1, ids, pbs[i]["para_fname"])

    run(0, 1); run(1, 1); torch.cuda.synchronize()
    upd = 3.0 * pbs[0]["n_c"] * (nsteps - 1) * 3
    t0 = time.perf_counter(); run(0, calls); run(1, calls); torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    st = fwi_ops.stats(pbs[0]["para_fname"], 0)
    fwd_phase = st["fwd_ms"] * 1e-3
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(0, calls)), threading.Thread(target=run, args=(1, calls, fwd_phase))]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_con = time.perf_counter() - t0
    print("2 x %d calls of 3 shots, %d steps: one after the other %.1f ms = %.1f Gcell-updates/s; overlapped (offset %.0f ms) %.1f ms = %.1f "
          "Gcell-updates/s  (x%.3f)" % (calls, nsteps, t_seq * 1e3, 2 * calls * upd / t_seq / 1e9, fwd_phase * 1e3, t_con * 1e3,
                                        2 * calls * upd / (t_con - fwd_phase) / 1e9, t_seq / (t_con - fwd_phase)))
finally:
    for w in works:
        shutil.rmtree(w, ignore_errors=True)
