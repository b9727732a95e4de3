#!/bin/bash
# round 5: instruction-issue side of the persistent backward loop (k_bwd_persist): how busy is the vector ALU with the loop's own
# bookkeeping (793 v_readlane of scalar spills in the ISA) -- SQ instruction counts / active cycles and the vector-L1 miss path,
# in separate --pmc passes of the bench command at 400 time steps (--kernel-trace only, as the pool requires).
TAG=r05sq
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$n -- python $R/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline --no-call32 $EXTRA > $R/$OUT/$n.log 2>&1 || { echo "pass $n failed: $(tail -2 $R/$OUT/$n.log | cut -c1-300)"; return 1; }
}
run s1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run s2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS &&
run s3 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum &&
run s4 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
cd $R
python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/s*/
grep -A40 "k_bwd_persist" $OUT/summary.txt | cut -c1-160 | head -60
