#!/usr/bin/env python
"""The headline step with the reference's own calling convention -- CPU tensors in, CPU tensors out (3 x 9 MB each way over PCIe per call) --
beside the bench's HBM-resident tensors: the PCIe-inclusive rate DESIGN.md section 1 quotes.  One GPU, three shots, 4000 steps."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import torch

import bench
from sepfwi import fwi_ops

with tempfile.TemporaryDirectory() as d:
    pb = bench.setup_problem(d, 1000, 2000, 4000, 3)
    dev = torch.device("cuda", 0)
    ids = torch.arange(3, dtype=torch.int32)
    lt = [t.to(dev) for t in pb["lame_true"]]
    fwi_ops._cufd(3, 0, lt[0], lt[1], lt[2], pb["Stf"], ids, pb["para_fname"])
    host = list(pb["lame_init"])
    hbm = [t.to(dev) for t in host]
    upd = 3 * 3.0 * pb["n_c"] * 3999
    for name, m in (("HBM tensors (bench)", hbm), ("CPU tensors (the reference's convention)", host), ("HBM tensors (bench)", hbm), ("CPU tensors (the reference's convention)", host)):
        fwi_ops.backward(m[0], m[1], m[2], pb["Stf"], 1, ids, pb["para_fname"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = fwi_ops.backward(m[0], m[1], m[2], pb["Stf"], 1, ids, pb["para_fname"])
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 3
        print("%-42s %8.2f ms per call  %.3f Gcell-updates/s  (gradients on %s)" % (name, 1e3 * el, upd / el / 1e9, out[1].device))
