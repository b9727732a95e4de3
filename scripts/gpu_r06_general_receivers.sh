#!/bin/bash
# round 6, review item 2: receivers that are not a fused line of consecutive channels now run inside the persistent loop (folded adjoint
# source, k_bwd_persist<LMASK, GINJ>) instead of the two-launch step + k_inject.  Headline model, three shots, a channel every 3 cells.
mkdir -p gpurun_out
OUT=gpurun_out/r06_general_receivers.txt; : > $OUT
echo "== 2000x1000, 1500 time steps, 660 channels (every third cell): us per time step and shot" | tee -a $OUT
timeout -k 10 900 python scripts/ab_bench.py --nsteps 1500 --rounds 2 --rec-stride 3 "" "bwd_fuse=2" 2>&1 | grep -v -e amdgpu.ids -e "^WARNING" | tee -a $OUT
echo "== the same model with the fused line of 1980 consecutive channels" | tee -a $OUT
timeout -k 10 900 python scripts/ab_bench.py --nsteps 1500 --rounds 2 "" "bwd_fuse=2" 2>&1 | grep -v -e amdgpu.ids -e "^WARNING" | tee -a $OUT
