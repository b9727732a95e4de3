#!/bin/bash
# round 5, Step B: register budget of the persistent loop (waves per workgroup x waves per SIMD the kernel is compiled for), strip widths
mkdir -p gpurun_out
timeout -k 10 800 python scripts/ab_bench.py --nsteps 2000 --rounds 2 \
  "bwd_fuse=2" "bwd_fuse=4,pk_px=3" \
  "bwd_fuse=4,pk_px=3,pk_waves=14,pk_wpe=7" "bwd_fuse=4,pk_px=3,pk_waves=12,pk_wpe=6" \
  "bwd_fuse=4,pk_px=3,pk_waves=16,pk_wpe=4,pk_wpc=1" "bwd_fuse=4,pk_px=4,pk_waves=14,pk_wpe=7" \
  "bwd_fuse=4,pk_px=2,pk_waves=14,pk_wpe=7" "bwd_fuse=4,pk_px=3,pk_waves=13,pk_wpe=6" \
  > gpurun_out/r05_persist_b2.log 2>&1
rc=$?
cat gpurun_out/r05_persist_b2.log
exit $rc
