#!/bin/bash
# round 4, call B: z-register-tiling probe, the updated GPU tests touched this round, then a fuzz sweep on the new yardstick
mkdir -p gpurun_out
( cd scripts/probes && timeout -k 10 120 ./xtap_probe ) > gpurun_out/r04_xtap_probe.txt 2>&1 || { cat gpurun_out/r04_xtap_probe.txt; exit 1; }
cat gpurun_out/r04_xtap_probe.txt
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_conditioning.py tests/test_gpu_fuzz.py -m gpu -x -q ) > gpurun_out/r04_b_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04_b_tests.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_r04_fuzz.sh 30000 1500
