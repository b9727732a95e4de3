#!/bin/bash
# round 4: are the option defaults of round 1-2 still the best on today's boxes?  (interleaved A/B, 2000x1000, three shots, 1200 steps)
mkdir -p gpurun_out
timeout -k 10 900 python scripts/ab_bench.py --nsteps 1200 --rounds 3 "" "bz=1" "bz=4" "early=1" "early=2" "early=3" "xcd_remap=0" "rho_fly=3" "amu_fly=3" "rk_lazy=0" "fwd_lanes=2" "fwd_lanes=4" "batch=1" > gpurun_out/r04_options_ab.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r04_options_ab.log
exit $rc
