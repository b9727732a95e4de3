#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "line_fuse=0" "line_fuse=1" > gpurun_out/ab6.log 2>&1
cat gpurun_out/ab6.log
