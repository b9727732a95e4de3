#!/usr/bin/env python
"""Can ONE low-priority forward lane fill the kernel-boundary bubbles of the (single-lane) backward passes?
Session A: gradient calls of 3 shots on normal-priority streams.  Session B: forward-only calls of one shot, every stream of the
session at the lowest HIP priority (SEPFWI_STREAM_PRIO=low while the session is created).  Timed alone, then A's calls with B
looping beside them until A is through.  Prints A's slow-down, B's progress and what the sum is worth.
    python scripts/prio_probe.py [nsteps] [calls] [low|normal]"""
import os, shutil, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch
import bench
from sepfwi import fwi_ops

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prio = sys.argv[3] if len(sys.argv) > 3 else "low"
dev = torch.device("cuda", 0)
works = [tempfile.mkdtemp(prefix="sepfwi_pr%d_" % i) for i in range(2)]
try:
    pbs = [bench.setup_problem(w, 1000, 2000, nsteps, 3) for w in works]
    ids = torch.arange(3, dtype=torch.int32)
    ins = []
    for i, pb in enumerate(pbs):
        if i == 1 and prio == "low":
            os.environ["SEPFWI_STREAM_PRIO"] = "low"          # read when session B is created (its first call)
        lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
        fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids[:1] if i == 1 else ids, pb["para_fname"])
        os.environ.pop("SEPFWI_STREAM_PRIO", None)
        ins.append([t.to(dev) for t in pb["lame_init"]])

    def run_a(n):
        lam, mu, den = ins[0]
        for _ in range(n):
            fwi_ops.backward(lam, mu, den, pbs[0]["Stf"], 1, ids, pbs[0]["para_fname"])
        torch.cuda.synchronize()

    stop = threading.Event()
    done_b = [0]

    def run_b(n=None):
        lam, mu, den = ins[1]
        k = 0
        while (n is None and not stop.is_set()) or (n is not None and k < n):
            fwi_ops.forward(lam, mu, den, pbs[1]["Stf"], 0, ids[:1], pbs[1]["para_fname"])
            k += 1
        done_b[0] = k

    run_a(1); run_b(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run_a(calls); t_a = time.perf_counter() - t0
    nb = 3 * calls
    t0 = time.perf_counter(); run_b(nb); torch.cuda.synchronize(); t_b1 = (time.perf_counter() - t0) / nb
    st = fwi_ops.stats(pbs[0]["para_fname"], 0)
    th = threading.Thread(target=run_b)
    t0 = time.perf_counter()
    th.start(); run_a(calls); t_both = time.perf_counter() - t0
    stop.set(); th.join(); torch.cuda.synchronize()
    k = done_b[0] - 1                      # the call in flight when A finished is not counted
    lane3 = 22.5e-6 * (nsteps - 1)         # what a forward pass costs inside a three-lane group
    print("priority %s, %d steps: A alone %.1f ms (%d calls of 3 shots); B alone %.1f ms per forward shot; together %.1f ms "
          "(A x%.3f slower) while B finished %d forward shots  => sequential equivalent %.1f ms with B at its own rate (x%.3f), "
          "%.1f ms with B at the three-lane rate (x%.3f)" %
          (prio, nsteps, t_a * 1e3, calls, t_b1 * 1e3, t_both * 1e3, t_both / t_a, k, (t_a + k * t_b1) * 1e3, (t_a + k * t_b1) / t_both,
           (t_a + k * lane3) * 1e3, (t_a + k * lane3) / t_both))
finally:
    for w in works:
        shutil.rmtree(w, ignore_errors=True)
