#!/bin/bash
# round-1 record run: GPU tests, smoke, full bench, kernel-trace profile and PMC traffic of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
( time timeout 900 python bench.py --steps 2 --warmup 1 ) > gpurun_out/bench_full.log 2>&1; tail -4 gpurun_out/bench_full.log
cd /tmp
rm -rf $R/gpurun_out/prof_r01;
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/trace -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r01_trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline > $R/gpurun_out/prof_r01_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_r01/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline > $R/gpurun_out/prof_r01_write.log 2>&1
cd $R
cat gpurun_out/prof_r01/trace/*/*kernel_stats.csv | head -12
python scripts/pmc_summary.py gpurun_out/prof_r01/pmc_fetch | head -12
python scripts/pmc_summary.py gpurun_out/prof_r01/pmc_write | head -16
