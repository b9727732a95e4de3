#!/bin/bash
# one-launch backward step experiment: parity of every tile shape, then A/B timing
mkdir -p gpurun_out
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_identical or variants" ) > gpurun_out/r03_fused_pytest.log 2>&1
rc=$?
tail -12 gpurun_out/r03_fused_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/ab_bench.py --nsteps 600 --rounds 2 "" "bwd_fuse=3" "bwd_fuse=3,fuse_cfg=1" "bwd_fuse=3,fuse_cfg=2" "bwd_fuse=3,fuse_cfg=3" "bwd_fuse=3,fuse_cfg=4" "bwd_fuse=3,fuse_cfg=5" > gpurun_out/r03_fused_ab1.log 2>&1
rc=$?
cat gpurun_out/r03_fused_ab1.log
exit $rc
