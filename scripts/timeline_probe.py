#!/usr/bin/env python
"""Wave timeline of one k_bwd_a and one k_bwd_b launch at the headline size (probe build of the library:
    SEPFWI_HIPCC_FLAGS=-DSEPFWI_TIMELINE python -c "import sys; sys.path.insert(0, 'sep-2023_amd'); from sepfwi import _native; _native.build(force=True)"
    python scripts/timeline_probe.py [nsteps]
): when every wave started and ended (100 MHz clock), on which XCD, how long the launch ramps up and tails off, how even the
XCDs finish, how long a wave lives by row class.  The last launches of a backward pass are recorded (time step 0)."""
import ctypes as C
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import numpy as np
import torch
import bench
from sepfwi import _native, fwi_ops

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
L = _native.lib()
if not hasattr(L, "sepfwi_probe_timeline"):
    raise SystemExit("build the library with -DSEPFWI_TIMELINE first (see the docstring)")
TL_MAX = 1 << 17
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="sepfwi_tl_")
try:
    pb = bench.setup_problem(work, 1000, 2000, nsteps, 1)
    ids = torch.arange(1, dtype=torch.int32)
    lt, mt, dt_ = [t.to(dev) for t in pb["lame_true"]]
    fwi_ops._cufd(2, 0, lt, mt, dt_, pb["Stf"], ids, pb["para_fname"])
    lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
    for _ in range(2):
        fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    torch.cuda.synchronize()
    times = np.zeros((2, TL_MAX, 2), dtype=np.uint64)
    xcc = np.zeros((2, TL_MAX), dtype=np.uint32)
    L.sepfwi_probe_timeline.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sepfwi_probe_timeline(times.ctypes.data, xcc.ctypes.data) == TL_MAX
    bz = L.sepfwi_get_option(b"bz")
    nz_c, gx = 1064, 33
    for k, name in enumerate(("k_bwd_a", "k_bwd_b")):
        t = times[k].astype(np.int64)
        live = t[:, 1] > 0
        n = int(live.sum())
        w = np.nonzero(live)[0]
        t0 = t[live, 0].min()
        s, e = (t[live, 0] - t0) / 100.0, (t[live, 1] - t0) / 100.0          # microseconds
        x = xcc[k][live] & 15
        blk = w // bz
        print("%s: %d waves recorded, launch span %.2f us (first wave start -> last wave end); wave lifetime mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f"
              % (name, n, e.max(), (e - s).mean(), *np.percentile(e - s, [10, 50, 90])))
        print("  XCC_ID == blockIdx %% 8 for %.2f %% of the waves" % (100.0 * np.mean(x == (blk & 7))))
        for q in range(8):
            m = x == q
            print("  XCD %d: %5d waves, first start %.2f us, last start %.2f us, last end %.2f us, mean lifetime %.2f us" %
                  (q, int(m.sum()), s[m].min(), s[m].max(), e[m].max(), (e[m] - s[m]).mean()))
        grid = np.arange(0.0, e.max() + 0.25, 0.25)
        act = np.array([np.count_nonzero((s <= g) & (e > g)) for g in grid])
        peak = act.max()
        up = grid[np.argmax(act >= 0.9 * peak)]
        down = grid[len(act) - 1 - np.argmax(act[::-1] >= 0.9 * peak)]
        print("  resident waves: peak %d of 8192 slots; >= 90 %% of the peak from %.2f us to %.2f us => ramp %.2f us, tail %.2f us of %.2f"
              % (peak, up, down, up, e.max() - down, e.max()))
        print("  resident waves every 2 us: " + " ".join("%d" % a for a in act[::8]))
        # logical tile of each wave: the XCD-banded order of my_cell()
        nblk_pad = ((gx * ((nz_c + bz - 1) // bz) + 7) // 8) * 8
        per = nblk_pad // 8
        lt_ = (blk & 7) * per + (blk >> 3)
        row = (lt_ // gx) * bz + (w % bz)
        life = e - s
        for lab, m in (("C-PML rows (z < 32 or z >= 1032)", (row < 32) | ((row >= 1032) & (row < nz_c))), ("interior rows", (row >= 32) & (row < 1032)),
                       ("surplus / out-of-range waves", row >= nz_c)):
            if m.any():
                print("  %-34s %6d waves, mean lifetime %.2f us" % (lab, int(m.sum()), life[m].mean()))
finally:
    shutil.rmtree(work, ignore_errors=True)
