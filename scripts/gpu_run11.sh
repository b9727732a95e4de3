#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "bwd_fuse=2,bz=1" "bwd_fuse=1,bz=1" > gpurun_out/ab11.log 2>&1
cat gpurun_out/ab11.log
