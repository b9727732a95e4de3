#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python scripts/ab_bench.py --nsteps 300 --rounds 3 "fwd_fuse=0" "fwd_fuse=1" > gpurun_out/ab5.log 2>&1
cat gpurun_out/ab5.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/pmc_q; mkdir -p $R/gpurun_out/pmc_q
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_q/p1 -- python $R/scripts/ab_bench.py --nsteps 40 --rounds 1 "fwd_fuse=1" > $R/gpurun_out/pmc_q/p1.log 2>&1
cd $R; python scripts/pmc_summary.py gpurun_out/pmc_q | grep -A9 "k_fwd_fused<true>"
