#!/bin/bash
# round 3, first GPU call: the -m gpu suite (with the new 32-shot full-size call) and the 1-GPU end-to-end L-BFGS proxy of
# configs[4] (12 shots x 10 iterations) with its host-time split
mkdir -p gpurun_out
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s -k "32_shot" ) > gpurun_out/r03_call32_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/r03_call32_pytest.log
[ $rc -eq 0 ] || exit $rc
( time timeout -k 10 900 python examples/das_fwi_2000x1000.py --shots 12 --niter 10 ) > gpurun_out/r03_e2e_1gpu.log 2>&1
rc=$?
tail -12 gpurun_out/r03_e2e_1gpu.log
exit $rc
