#!/bin/bash
# usage: gpu_ab2.sh <nsteps> <rounds> variant...   (A/B of kernel options, log kept under gpurun_out/ab_last.log)
mkdir -p gpurun_out
N=$1; R=$2; shift; shift
timeout -k 10 900 python scripts/ab_bench.py --nsteps $N --rounds $R "$@" > gpurun_out/ab_last.log 2>&1
rc=$?
grep -v WARNING gpurun_out/ab_last.log | tail -30
exit $rc
