#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "rz=1" "rz=2" "rz=4" "rz=8" "rz=2,bz=2" > gpurun_out/ab12.log 2>&1
cat gpurun_out/ab12.log
