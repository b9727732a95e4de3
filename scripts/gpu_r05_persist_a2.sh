#!/bin/bash
# NOTE: runs at commit f8b3d74 (the probe kernel; git history c42f644: scripts/probes/r05_persistent_bwd_probe.patch), not on the current tree.
# round 5, Step A (second pass): dynamic segment hand-out, no barrier between phases, 32 waves per CU; a grid with 32 column
# segments (no tile imbalance) for comparison.  No cross-tile synchronisation: results wrong, timing only.
mkdir -p gpurun_out
python -c "
import sys; sys.path[:0]=['sep-2023_amd']
from sepfwi import _native; _native.build()"
timeout -k 10 500 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=1" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=3" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=1,pk_waves=8,pk_wpc=3" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=3,pk_waves=8,pk_wpc=3" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=5,pk_waves=16,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=7,pk_waves=16,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=7,pk_waves=8,pk_wpc=4" \
  "bwd_fuse=4,pk_lmask=0,pk_flags=7,pk_waves=8,pk_wpc=4" \
  > gpurun_out/r05_persist_a2.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/ab_bench.py --nsteps 2000 --rounds 2 --nx 1984 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=1" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=3" \
  "bwd_fuse=4,pk_lmask=7,pk_flags=3,pk_waves=8,pk_wpc=3" \
  > gpurun_out/r05_persist_a2_nx1984.log 2>&1
rc=$?
grep -v WARNING gpurun_out/r05_persist_a2_nx1984.log
exit $rc
