#!/bin/bash
# usage: gpu_bench_profile.sh <tag>   -- the record run of a round: full bench line, rocprofv3 kernel-trace summary of the same
# command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) behind roofline.traffic.  Copies the summaries to gpurun_out/<tag>_*.
TAG=${1:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( time timeout -k 10 600 python bench.py --steps 4 --warmup 1 ) > gpurun_out/${TAG}_bench_full.log 2>&1 || exit 1
tail -4 gpurun_out/${TAG}_bench_full.log | cut -c1-900
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-call32 > $R/gpurun_out/${TAG}_trace.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline --no-call32 > $R/gpurun_out/${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline --no-call32 > $R/gpurun_out/${TAG}_write.log 2>&1 || exit 1
cd $R
cp gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
python scripts/pmc_summary.py gpurun_out/prof_$TAG/pmc_fetch > gpurun_out/${TAG}_bench_pmc_fetch.txt
python scripts/pmc_summary.py gpurun_out/prof_$TAG/pmc_write > gpurun_out/${TAG}_bench_pmc_write.txt
head -8 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-200
head -14 gpurun_out/${TAG}_bench_pmc_fetch.txt
# keep the raw kernel trace small enough to merge back: first shot group only
rm -rf gpurun_out/prof_$TAG   # raw traces and counter CSVs (tens of MB: gpurun merges at most 64 MiB back); the summaries above are what is kept
# the same bench line with the two-launch backward step (option bwd_fuse=2): what the persistent loop is compared with
( timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-call32 --option bwd_fuse=2 ) > gpurun_out/${TAG}_bench_two_launch.log 2>&1
tail -1 gpurun_out/${TAG}_bench_two_launch.log | cut -c1-400
# configs[1] shape, forward only, un-profiled
( timeout -k 10 300 python bench.py --mode fwd --nz 500 --nsteps 2000 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/${TAG}_fwd2000x500_bench.log 2>&1
tail -1 gpurun_out/${TAG}_fwd2000x500_bench.log | cut -c1-400
# configs[3]'s per-GPU load in one call: 32 shots per step (one step: 10.8 s)
( timeout -k 10 400 python bench.py --shots-per-step 32 --steps 1 --warmup 0 --no-cpu-baseline --no-call32 ) > gpurun_out/${TAG}_bench_32shots.log 2>&1
tail -1 gpurun_out/${TAG}_bench_32shots.log | cut -c1-400
