#!/bin/bash
# round 4: layout experiment -- parity of the paired layouts, then A/B timing at the headline size
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or bit_identical or paired" ) > gpurun_out/r04_pair_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04_pair_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python scripts/ab_bench.py --nsteps 1200 --rounds 3 "pair=0" "pair=1" "pair=2" "pair=4" "pair=3" "pair=7" "pair=7,acc_nt=1" "acc_nt=1" > gpurun_out/r04_pair_ab.log 2>&1
rc=$?
cat gpurun_out/r04_pair_ab.log
exit $rc
