#!/bin/bash
# round 6: the multi-shot persistent loop (one launch per backward sub-batch of the batched schedule; option pk_ms of the -DSEPFWI_PROBES
# build) against the per-step batched launches (the default: first line of every block)
# on grids below the headline's size, the reference's own experiment size first (101x201 cells x 19 shots, notebooks/Main-001-...py:30-34)
mkdir -p gpurun_out
OUT=gpurun_out/r06_other_grids.txt; : > $OUT
run() { echo "== $1" | tee -a $OUT; shift; timeout -k 10 600 python scripts/ab_bench.py --rounds 3 "$@" 2>&1 | grep -v -e amdgpu.ids -e "^WARNING" | tee -a $OUT; }
run "200x100 x 19 shots, 1500 steps (the notebooks' shape on bench.py's model)" --nsteps 1500 --nz 100 --nx 200 --shots 19 "" "pk_ms=1" "pk_ms=1,pk_nosync=1" "pk_ms=1,pk_px=5" "pk_ms=1,pk_wpc=1" "pk_ms=1,pk_wpc=4,pk_waves=8"
run "500x250 x 16 shots" --nsteps 600 --nz 250 --nx 500 --shots 16 "" "pk_ms=1" "pk_ms=1,pk_nosync=1"
run "1000x500 x 12 shots" --nsteps 600 --nz 500 --nx 1000 --shots 12 "" "pk_ms=1" "batch=0"
run "1500x500 x 6 shots" --nsteps 600 --nz 500 --nx 1500 --shots 6 "" "pk_ms=1" "batch=0"
run "2000x500 x 3 shots (configs[1] shape, fwd+adj)" --nsteps 1000 --nz 500 "" "batch=1" "batch=1,pk_ms=1"
run "200x100 x 19 shots, 1500 steps, a channel every third cell (not a fused line: batched k_record / k_inject, one launch per sub-batch)" --nsteps 1500 --nz 100 --nx 200 --shots 19 --rec-stride 3 ""
run "500x250 x 16 shots, a channel every third cell" --nsteps 600 --nz 250 --nx 500 --shots 16 --rec-stride 3 ""
echo "== experiment 001 of the reference (101x201, 1501 steps, 19 shots): one gradient evaluation" | tee -a $OUT
timeout -k 10 300 python scripts/small_grid_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT
