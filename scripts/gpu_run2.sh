#!/bin/bash
mkdir -p gpurun_out
./scripts/probes/bw_probe > gpurun_out/bw_probe.log 2>&1
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "bz=4,xcd_remap=0" "bz=4,xcd_remap=1" "bz=8,xcd_remap=0" "bz=8,xcd_remap=1" "bz=16,xcd_remap=1" "bz=2,xcd_remap=1" > gpurun_out/ab2.log 2>&1
cat gpurun_out/bw_probe.log gpurun_out/ab2.log
