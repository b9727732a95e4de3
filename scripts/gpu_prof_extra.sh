#!/bin/bash
# kernel-trace summaries of the two other measured workloads: configs[1] shape (2000x500, forward only) and a
# reference-notebook-sized problem in batched mode (101x201, 19 shots)
mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_extra
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_extra/fwd500 -- python $R/bench.py --mode fwd --nz 500 --nsteps 2000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_extra_fwd500.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_extra/small -- python $R/scripts/small_grid_probe.py > $R/gpurun_out/prof_extra_small.log 2>&1
cd $R
head -6 gpurun_out/prof_extra/fwd500/*/*kernel_stats.csv | cut -c1-60,200-300
head -8 gpurun_out/prof_extra/small/*/*kernel_stats.csv | cut -c1-60,200-300
tail -1 gpurun_out/prof_extra_fwd500.log | cut -c1-200; tail -1 gpurun_out/prof_extra_small.log
