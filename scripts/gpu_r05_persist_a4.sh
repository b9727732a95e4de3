#!/bin/bash
# NOTE: runs at commit f8b3d74 (the probe kernel; git history c42f644: scripts/probes/r05_persistent_bwd_probe.patch), not on the current tree.
# round 5, Step A (fourth pass): natural (row-major within the strip) order of a tile's segments against edge-first.
mkdir -p gpurun_out
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=0,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=7,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0,pk_px=3" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0,pk_px=8" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0,pk_px=33" \
  > gpurun_out/r05_persist_a4.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a4.log
exit $rc
