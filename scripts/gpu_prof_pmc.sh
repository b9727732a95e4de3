#!/bin/bash
# usage: gpu_prof_pmc.sh <tag> <variant> ; collects three PMC passes + kernel trace for one ab_bench variant
TAG=$1; VAR=$2
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
if [ ! -f $R/gpurun_out/counters_list.txt ]; then rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1; fi
run() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG/$n -- python $R/scripts/ab_bench.py --nsteps 40 --rounds 1 "$VAR" > $R/gpurun_out/pmc_$TAG/$n.log 2>&1
}
run p1 FETCH_SIZE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run p2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU
run p3 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt 2>&1
cat gpurun_out/pmc_$TAG/summary.txt
