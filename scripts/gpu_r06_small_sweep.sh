#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python scripts/ab_bench.py --rounds 2 --nsteps 1500 --nz 100 --nx 200 --shots 19 "" "bz=1" "bz=4" "bz=8" "batch_split=3" "batch_order=0" "early=3" "rho_fly=3" "amu_fly=3" "rk_lazy=0" "xcd_remap=0" "batch_f=10,batch_b=10" "batch_b=10" "batch_b=5" "line_fuse=0" 2>&1 | grep -v -e amdgpu.ids -e "^WARNING" | tee gpurun_out/r06_small_sweep.txt
