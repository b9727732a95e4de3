#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/ab_bench.py --nsteps 200 --rounds 2 "fwd_fuse=0" "fwd_fuse=2,march_waves=1280" "fwd_fuse=2,march_waves=2560" "fwd_fuse=2,march_waves=4700" "fwd_fuse=2,march_waves=7000" "fwd_fuse=2,march_waves=9400" > gpurun_out/ab9.log 2>&1
cat gpurun_out/ab9.log
