#!/bin/bash
# round 6 (re-record of round 5's run on this round's library): configs[4] on ONE GPU with the inverse problem on which L-BFGS-B (the reference's options) completes ten iterations
# usage: gpu_r04_e2e.sh <shots> [max-seconds]     (the log is written directly, line by line: the box kills a run that is silent for 7 min)
mkdir -p gpurun_out
S=$1; T=${2:-0}; X=${3:-}
( time timeout -k 10 1180 python -u examples/das_fwi_2000x1000.py --shots $S --niter 10 --pert 0.03 --sigma-init 40 --max-seconds $T $X ) > gpurun_out/r06_e2e_1gpu_${S}shots.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r06_e2e_1gpu_${S}shots.log | tail -8 | cut -c1-400
exit $rc
