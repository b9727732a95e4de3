#!/bin/bash
# round 6: the closing GPU call -- the -m gpu suite + smoke, the record run (bench line, rocprofv3 kernel stats, PMC passes), general
# receivers inside the loop, the small grids
bash scripts/gpu_tests.sh || exit 1
cp gpurun_out/pytest_gpu.log gpurun_out/r06_pytest_gpu.log
bash scripts/gpu_bench_profile.sh r06 || exit 1
bash scripts/gpu_r06_general_receivers.sh
bash scripts/gpu_r06_other_grids.sh
