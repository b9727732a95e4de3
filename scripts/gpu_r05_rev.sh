#!/bin/bash
# round 5, experiment #36i: phase B walks the tile's segments backwards (L2 reuse between the phases) -- timing + bit identity
set -e
mkdir -p gpurun_out
python scripts/ab_bench.py --nsteps 2000 --rounds 3 "bwd_fuse=4" "pk_rev=1" "pk_order=0" "pk_order=0,pk_rev=1" "pk_order=0,pk_rev=1,pk_px=2" "pk_order=0,pk_rev=1,pk_px=4" > gpurun_out/r05_rev.txt 2>&1
tail -12 gpurun_out/r05_rev.txt
