#!/usr/bin/env python
"""Wave timeline of the persistent backward loop from the SEPFWI_PK_TRACE dump (kernels.hip, built with -DSEPFWI_PK_TRACE):
where a wave's time goes inside a phase -- items, the closing drain, the wait at the barrier."""
import sys

import numpy as np

T, W, PH, S = 512, 16, 8, 12
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(T, W, PH, S).astype(np.float64)
tick = 0.01  # us per s_memrealtime tick (100 MHz)
ok = a[..., 0] > 0
print("traced (tile, wave, phase) records: %d of %d" % (ok.sum(), ok.size))
start, end_items, end_drain, nk = a[..., 0], a[..., 9], a[..., 10], a[..., 11].astype(int)
t0 = np.where(ok, start, np.inf).min(axis=1)            # [T, PH] earliest wave start of a phase
plen = (t0[:, 1:] - t0[:, :-1]) * tick
print("phase length per tile [us]: mean %.2f  min %.2f  max %.2f" % (plen.mean(), plen.min(), plen.max()))
items = nk - 1
print("items per wave and phase: mean %.2f (min %d, max %d)" % (items[ok].mean(), items[ok].min(), items[ok].max()))
busy = (end_items - start) * tick
drain = (end_drain - end_items) * tick
print("wave time in items: mean %.2f us; closing drain %.2f us" % (busy[ok].mean(), drain[ok].mean()))
nxt = np.empty_like(start); nxt[:, :, :-1] = start[:, :, 1:]; nxt[:, :, -1] = np.nan
wait = (nxt - end_drain) * tick
m = ok & np.isfinite(wait)
print("wave waits for the next phase (barrier + poll + invalidate): mean %.2f us, median %.2f, p90 %.2f" % (wait[m].mean(), np.median(wait[m]), np.percentile(wait[m], 90)))
sk = (start - t0[:, None, :]) * tick
print("start skew after the barrier: mean %.3f us max %.3f" % (sk[ok].mean(), sk[ok].max()))
for k in range(1, 7):
    m2 = ok & (nk > k + 1)
    if m2.sum():
        d = (a[..., k + 1] - a[..., k])[m2] * tick
        print("item %d of a wave: mean %.2f us (n=%d)" % (k, d.mean(), m2.sum()))
m3 = ok & (nk >= 2)
idx = np.nonzero(m3)
last = (end_items[idx] - a[idx + (nk[idx] - 1,)]) * tick
print("last item of a wave (start -> stores issued): mean %.2f us" % last.mean())
first_gap = (a[..., 1] - start)[m3] * tick
print("phase start -> first item start: mean %.3f us" % first_gap.mean())
tot = plen.mean()
print("share of the phase: items %.1f %%, drain %.1f %%, wait %.1f %%" % (100 * busy[ok].mean() / tot, 100 * drain[ok].mean() / tot, 100 * wait[m].mean() / tot))
# per phase parity (A = even local index)
for par, nm in ((0, "A"), (1, "B")):
    sel = np.zeros_like(ok); sel[:, :, par::2] = True
    mm = ok & sel
    print("phase %s: items %.2f us, wait %.2f us" % (nm, busy[mm].mean(), wait[mm & m].mean()))
# per tile and phase: work = first wave start -> last wave drained; sync = last wave drained -> next phase start (poll + barrier + invalidate)
last_d = np.where(ok, end_drain, -np.inf).max(axis=1)
first_d = np.where(ok, end_drain, np.inf).min(axis=1)
work = (last_d - t0)[:, :-1] * tick
sync = (t0[:, 1:] - last_d[:, :-1]) * tick
spread = (last_d - first_d)[:, :-1] * tick
print("per tile-phase: work %.2f us (p10 %.2f, p90 %.2f); after the LAST wave until the next phase starts %.2f us (p10 %.2f, median %.2f, p90 %.2f)" %
      (work.mean(), np.percentile(work, 10), np.percentile(work, 90), sync.mean(), np.percentile(sync, 10), np.median(sync), np.percentile(sync, 90)))
print("first-to-last wave finishing inside a tile: mean %.2f us (p90 %.2f)" % (spread.mean(), np.percentile(spread, 90)))
for ti in (0, 56, 160, 264, 504):
    print("tile %3d: work " % ti + " ".join("%5.1f" % v for v in work[ti]) + " | sync " + " ".join("%5.1f" % v for v in sync[ti]))
print("mean work [us] per tile, rows = band (XCD), columns = tile index in the band 0 ... 63:")
wm = work.mean(axis=1).reshape(8, 64)
for b in range(8):
    print("  band %d: " % b + " ".join("%2.0f" % v for v in wm[b]))
print("mean wait after the last wave [us]:")
sm = sync.mean(axis=1).reshape(8, 64)
for b in range(8):
    print("  band %d: " % b + " ".join("%2.0f" % v for v in sm[b]))
print("column means of work: " + " ".join("%2.0f" % v for v in wm.mean(axis=0)))
