#!/bin/bash
# GPU box: the uninitialised-read check.  SEPFWI_POISON=1 fills every fresh device allocation of the library with 0xFF bytes
# (csrc/device_alloc.hpp: NaN in every float); the whole -m gpu suite and a fuzz sweep then fail wherever a kernel reads memory
# nothing has written.  GPU AddressSanitizer is not available on the target pool; this is the check that is.
#   usage: gpu_poison.sh [fuzz seeds, default 600]
mkdir -p gpurun_out
export SEPFWI_POISON=1
( time timeout -k 10 800 python -m pytest tests -m gpu -q ) > gpurun_out/poison_suite.log 2>&1
grep "passed\|failed" gpurun_out/poison_suite.log | tail -1
export SEPFWI_FUZZ_N=${1:-600} OMP_NUM_THREADS=2
( time timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 ) > gpurun_out/poison_fuzz.log 2>&1
grep "passed\|failed" gpurun_out/poison_fuzz.log | tail -1
grep -h "^E   *AssertionError\|^FAILED\|crashed" gpurun_out/poison_suite.log gpurun_out/poison_fuzz.log | cut -c1-220 | head -30
exit 0
