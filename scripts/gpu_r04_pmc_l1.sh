#!/bin/bash
# round 4: is the vector-memory (TA / TCP) path the "second floor" of the field kernels?  SQ wait / issue split and TA / TCP busy counters
# on the default structure (separate passes, --kernel-trace only, as the pool requires).
mkdir -p gpurun_out/pmc_r04l1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
grep -o "Name:[A-Za-z0-9_]*" $R/gpurun_out/counters_list.txt | sort -u | grep -i "Name:TA_\|Name:TCP_\|Name:TD_" | tr "\n" " " | cut -c1-3000
echo
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_r04l1/$n -- python $R/scripts/ab_bench.py --nsteps 40 --rounds 1 "" > $R/gpurun_out/pmc_r04l1/$n.log 2>&1 || echo "pass $n failed: $(tail -2 $R/gpurun_out/pmc_r04l1/$n.log | cut -c1-200)"
}
if [ "$1" = "l2" ]; then   # second set: TLB and L2 <-> fabric
run r1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run r2 TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
run r3 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum
run r4 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
run r5 TCC_BUSY_sum TCC_TAG_STALL_sum
run r6 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run r7 TCC_REQ_sum TCC_HIT_sum
run r8 GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum
else
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p2 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE GRBM_TA_BUSY
run q1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run q2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run q3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run q4 TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ
run q5 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run q6 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
run q7 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
fi
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_r04l1 > gpurun_out/pmc_r04l1/summary.txt 2>&1
grep -A28 "k_bwd_b\|k_bwd_a\|k_velocity<true>\|k_stress<true, true>" gpurun_out/pmc_r04l1/summary.txt | cut -c1-160 | head -140
rm -rf gpurun_out/pmc_r04l1/p*/*/*kernel_trace.csv
