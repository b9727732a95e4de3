#!/bin/bash
# round 5: the two PMC passes of the bench command again (the summary script used to drop kernels with fewer than five launches:
# the persistent loop has three), FETCH_SIZE and WRITE_SIZE in separate runs
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline --no-call32 > $R/gpurun_out/${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline --no-call32 > $R/gpurun_out/${TAG}_write.log 2>&1 || exit 1
cd $R
python scripts/pmc_summary.py gpurun_out/prof_$TAG/pmc_fetch > gpurun_out/${TAG}_bench_pmc_fetch.txt
python scripts/pmc_summary.py gpurun_out/prof_$TAG/pmc_write > gpurun_out/${TAG}_bench_pmc_write.txt
rm -rf gpurun_out/prof_$TAG
cat gpurun_out/${TAG}_bench_pmc_fetch.txt gpurun_out/${TAG}_bench_pmc_write.txt | head -40
