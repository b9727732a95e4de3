#!/bin/bash
# round 5, validation of the refactored session / observed-data store: (1) the -m gpu suite and a fuzz sweep with NaN-poisoned
# allocations (scripts/gpu_poison.sh), (2) a 1500-seed fuzz sweep with the capped two-roundings yardstick (xfail = no parity target).
mkdir -p gpurun_out
bash scripts/gpu_poison.sh 400 > gpurun_out/r05_poison_check.txt 2>&1
cat gpurun_out/r05_poison_check.txt
A=${1:-70000}; N=${2:-1500}
SEEDS=$(python -c "print(','.join(str(s) for s in range($A, $A+$N)))")
rm -f gpurun_out/r05_fuzz_yard.txt
( time SEPFWI_FUZZ_SEEDS=$SEEDS SEPFWI_FUZZ_YARD=$PWD/gpurun_out/r05_fuzz_yard.txt OMP_NUM_THREADS=2 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 -p no:cacheprovider ) > gpurun_out/r05_fuzz_$A.log 2>&1
rc=$?
tail -6 gpurun_out/r05_fuzz_$A.log
grep -c XFAIL gpurun_out/r05_fuzz_$A.log
exit $rc
