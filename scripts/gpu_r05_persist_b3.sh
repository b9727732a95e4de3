#!/bin/bash
# round 5, Step B: census inside the loop's own kernel; wave-count variants again; the bench line with the persistent loop
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_backward or bit_identical" ) > gpurun_out/r05_persist_b3_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r05_persist_b3_pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 800 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" "bwd_fuse=4,pk_px=3" "bwd_fuse=4,pk_px=2" \
  "bwd_fuse=4,pk_px=3,pk_waves=14,pk_wpe=7" "bwd_fuse=4,pk_px=3,pk_waves=12,pk_wpe=6" "bwd_fuse=4,pk_px=3,pk_waves=8,pk_wpe=6,pk_wpc=3" \
  > gpurun_out/r05_persist_b3.log 2>&1
cat gpurun_out/r05_persist_b3.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-call32 --option bwd_fuse=4 --option pk_px=3 > gpurun_out/r05_persist_b3_bench.log 2>&1
tail -1 gpurun_out/r05_persist_b3_bench.log | cut -c1-1500
