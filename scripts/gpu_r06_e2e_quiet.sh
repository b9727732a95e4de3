#!/bin/bash
# round 6: the 128-shot end-to-end run of gpu_r06_e2e.sh with the opt-in quiet_skip (bit-identical iterates, less time)
mkdir -p gpurun_out
( time timeout -k 10 1180 python -u examples/das_fwi_2000x1000.py --shots 128 --niter 10 --pert 0.03 --sigma-init 40 --max-seconds 0 --mask-rows 30 --quiet-skip ) > gpurun_out/r06_e2e_1gpu_128shots_quiet_skip.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_e2e_1gpu_128shots_quiet_skip.log | tail -6 | cut -c1-400
exit $rc
