#!/usr/bin/env python
"""Does running two shots concurrently (two sessions, two streams, two host threads) on one GPU raise the aggregate
throughput (kernel-boundary gaps and tails filled) or lower it (working set beyond the 256 MB Infinity Cache)?"""
import os, shutil, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch
import bench
from sepfwi import fwi_ops

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mode = sys.argv[2] if len(sys.argv) > 2 else "bwd"
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 2
NZ = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
dev = torch.device("cuda", 0)
from sepfwi import _native
for kv in os.environ.get("SEPFWI_OPTS", "").split(","):   # e.g. SEPFWI_OPTS=batch=0,pair_fwd=0
    if kv:
        _native.check(_native.lib().sepfwi_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
works = [tempfile.mkdtemp(prefix="sepfwi_cc%d_" % i) for i in range(NC)]
try:
    pbs = [bench.setup_problem(w, NZ, 2000, nsteps, 1) for w in works]
    ids = torch.tensor([0], dtype=torch.int32)
    ins = []
    for pb in pbs:
        lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
        fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
        ins.append([t.to(dev) for t in pb["lame_init"]])
    def run(i):
        lam, mu, den = ins[i]
        if mode == "fwd":
            return fwi_ops.forward(lam, mu, den, pbs[i]["Stf"], 0, ids, pbs[i]["para_fname"])
        return fwi_ops.backward(lam, mu, den, pbs[i]["Stf"], 1, ids, pbs[i]["para_fname"])
    [run(i) for i in range(NC)]; torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter(); [run(i) for i in range(NC)]; torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=NC) as ex:
            list(ex.map(run, range(NC)))
        torch.cuda.synchronize(); t_con = time.perf_counter() - t0
        print("nz %d %s nsteps %d: %d shots sequential %.1f ms, concurrent %.1f ms  (x%.3f)" % (NZ, mode, nsteps, NC, t_seq * 1e3, t_con * 1e3, t_seq / t_con))
finally:
    for w in works:
        shutil.rmtree(w, ignore_errors=True)
