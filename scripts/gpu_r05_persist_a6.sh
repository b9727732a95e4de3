#!/bin/bash
# NOTE: runs on the probe kernel (commit f8b3d74 + git history c42f644: scripts/probes/r05_persistent_bwd_probe_byvalue.patch), not on the current tree.
# round 5, Step A (sixth pass): the argument block passed BY VALUE -- pointers that come from the kernel-argument segment are known to be
# global, so the bodies use global_load with graded vmcnt waits instead of flat_load with full vmcnt(0) lgkmcnt(0) drains (passes 1-5).
mkdir -p gpurun_out
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=7,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=1" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0,pk_px=3" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=3,pk_order=0" \
  "bwd_fuse=4,pk_lmask=0,pk_flags=5,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1,pk_wpc=3,pk_waves=8,pk_order=0" \
  > gpurun_out/r05_persist_a6.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a6.log
exit $rc
