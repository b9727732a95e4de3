#!/bin/bash
# round 5, Step B first run: the synchronised persistent backward loop -- bit-identity test, then timing on the headline grid
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_backward or bit_identical" ) > gpurun_out/r05_persist_b1_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/r05_persist_b1_pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" "bwd_fuse=4" "bwd_fuse=4,pk_nosync=1" "bwd_fuse=4,pk_wpc=1" "bwd_fuse=4,pk_px=3" "bwd_fuse=4,pk_lmask=7" \
  > gpurun_out/r05_persist_b1.log 2>&1
rc=$?
cat gpurun_out/r05_persist_b1.log
exit $rc
