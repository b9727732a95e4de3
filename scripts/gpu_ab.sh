#!/bin/bash
# usage: gpu_ab.sh <nsteps> variant...
mkdir -p gpurun_out
N=$1; shift
timeout 900 python scripts/ab_bench.py --nsteps $N --rounds 3 "$@" > gpurun_out/ab_last.log 2>&1
cat gpurun_out/ab_last.log
