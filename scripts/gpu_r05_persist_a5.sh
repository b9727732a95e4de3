#!/bin/bash
# NOTE: runs on the probe kernel (commit f8b3d74 + the pipelined hand-out; git history c42f644: scripts/probes/r05_persistent_bwd_probe.patch + r05_persistent_bwd_probe_pipelined.patch), not on the current tree.
# round 5, Step A (fifth pass): the segment hand-out and the segment descriptor of the NEXT trips fetched while the current one waits for memory.
mkdir -p gpurun_out
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=5,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=13,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=15,pk_wpc=2,pk_order=0" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=13,pk_wpc=2,pk_order=0,pk_px=3" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=13,pk_wpc=2,pk_order=1" \
  "bwd_fuse=4,pk_lmask=15,pk_flags=9,pk_order=0" \
  "bwd_fuse=4,pk_lmask=0,pk_flags=13,pk_wpc=2,pk_order=0" \
  > gpurun_out/r05_persist_a5.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a5.log
exit $rc
