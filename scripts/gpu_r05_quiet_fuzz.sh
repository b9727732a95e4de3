#!/bin/bash
# round 5: quiet_skip against the plain kernels on 300 seeded random geometries / launch structures, bit for bit
mkdir -p gpurun_out
SEPFWI_QFUZZ_N=300 timeout -k 10 1100 python -m pytest tests/test_gpu_quiet_skip.py -x -q -m gpu -k random > gpurun_out/r05_quiet_fuzz.txt 2>&1
rc=$?
tail -5 gpurun_out/r05_quiet_fuzz.txt | cut -c1-300
exit $rc
