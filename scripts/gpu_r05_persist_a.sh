#!/bin/bash
# NOTE: runs at commit f8b3d74 (the probe kernel; git history c42f644: scripts/probes/r05_persistent_bwd_probe.patch), not on the current tree.
# round 5, Step A: the persistent backward time loop WITHOUT cross-tile synchronisation (results wrong, timing only) against
# the two-launch step.  Variants: accumulators in HBM / in LDS, one or two workgroups per CU, tile shapes.
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_identical or variants_agree or gradient_matches" ) > gpurun_out/r05_persist_a_pytest.log 2>&1
rc=$?
tail -3 gpurun_out/r05_persist_a_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" \
  "bwd_fuse=4,pk_lmask=0" \
  "bwd_fuse=4,pk_lmask=7" \
  "bwd_fuse=4,pk_lmask=24" \
  "bwd_fuse=4,pk_lmask=0,pk_waves=8,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=7,pk_waves=8,pk_wpc=2" \
  "bwd_fuse=4,pk_lmask=7,pk_waves=8,pk_wpc=3" \
  "bwd_fuse=4,pk_lmask=7,pk_px=4" \
  "bwd_fuse=4,pk_lmask=7,pk_px=16" \
  "bwd_fuse=4,pk_lmask=7,pk_waves=12" \
  > gpurun_out/r05_persist_a.log 2>&1
rc=$?
cat gpurun_out/r05_persist_a.log
exit $rc
