#!/bin/bash
# round 5: option quiet_skip -- the GPU suite at this tree, the coverage / timing table of DESIGN 3.3, and the default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r05_pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r05_pytest_gpu.log
timeout -k 10 600 python scripts/quiet_coverage.py 500 1000 2000 3000 4000 > gpurun_out/r05_quiet_skip.txt 2>&1
tail -5 gpurun_out/r05_quiet_skip.txt
timeout -k 10 400 python bench.py > gpurun_out/r05_bench_after_quiet.log 2>&1
tail -1 gpurun_out/r05_bench_after_quiet.log | cut -c1-400
