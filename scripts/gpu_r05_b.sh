#!/bin/bash
# round 5, second GPU call: fused rock-physics maps, the multi-rank rehearsal tests, the bench line with call32 / traffic_ratio
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests/test_gpu_param_maps.py -x -q -m gpu -s ) > gpurun_out/r05_b_param_maps.log 2>&1
rc=$?; tail -5 gpurun_out/r05_b_param_maps.log; [ $rc -eq 0 ] || exit $rc
( timeout -k 10 1500 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "fail_cleanly or four_ranks_reproduce" ) > gpurun_out/r05_b_dist.log 2>&1
rc=$?; tail -5 gpurun_out/r05_b_dist.log; [ $rc -eq 0 ] || exit $rc
( timeout -k 10 600 python bench.py --no-cpu-baseline ) > gpurun_out/r05_b_bench.log 2>&1
rc=$?; tail -3 gpurun_out/r05_b_bench.log
exit $rc
