#!/bin/bash
# NOTE: runs at commit f8b3d74 (the probe kernel; git history c42f644: scripts/probes/r05_persistent_bwd_probe.patch), not on the current tree.
# round 5, Step A (third pass): balanced tiles from the host-built plan (137-138 row segments each), three or four
# accumulators in LDS.  Still no cross-tile synchronisation: results wrong, timing only.
mkdir -p gpurun_out
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=1" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=3" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1,pk_px=6" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=7,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_px=6" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=1,pk_wpc=3,pk_waves=8" \
  "bwd_fuse=4,pk_lmask=0,pk_flags=5,pk_wpc=2" \
  > gpurun_out/r05_persist_a3.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a3.log
exit $rc
