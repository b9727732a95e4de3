#!/bin/bash
# round 4, call C: example smoke test, then 64 shots x 10 iterations with the L-BFGS restart
mkdir -p gpurun_out
python -m pytest tests/test_examples.py -m gpu -q -x > gpurun_out/r04_ex_test.log 2>&1 || { tail -30 gpurun_out/r04_ex_test.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/r04_ex_test.log
bash scripts/gpu_r04_e2e.sh 64
