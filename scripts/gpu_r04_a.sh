#!/bin/bash
# round 4, call A: the new tests (known values, observe-to-store, collective record, 4-rank rehearsal), then one bench line
mkdir -p gpurun_out
( timeout -k 10 1000 python -m pytest tests/test_dist_gpu.py tests/test_conditioning.py -m gpu -x -q -k "four_ranks or rccl or conditioning_reference" ) > gpurun_out/r04_a_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r04_a_tests.log
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 600 python bench.py --steps 5 --warmup 1 ) > gpurun_out/r04_a_bench.log 2>&1
rc=$?
tail -3 gpurun_out/r04_a_bench.log
exit $rc
