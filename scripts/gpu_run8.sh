#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "fwd_fuse=0" "fwd_fuse=1" "fwd_fuse=2" > gpurun_out/ab8.log 2>&1
cat gpurun_out/ab8.log
