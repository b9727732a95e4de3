#!/bin/bash
# round 6: quiet row segments inside the persistent loop (quiet_skip=1,pk_quiet=1 of the -DSEPFWI_PROBES build) against the loop without them
# ("") and against the two-launch step with them (quiet_skip=1: what the shipped library runs with the option on), by record length on the headline model; then seeded sweeps of the bit-identity tests
mkdir -p gpurun_out
OUT=gpurun_out/r06_quiet_skip.txt; : > $OUT
for NS in 500 1000 2000 4000; do
  echo "== 2000x1000, $NS time steps, three shots (us per time step and shot)" | tee -a $OUT
  timeout -k 10 900 python scripts/ab_bench.py --nsteps $NS --rounds 2 "" "quiet_skip=1,pk_quiet=1" "quiet_skip=1" "bwd_fuse=2" 2>&1 | grep -v -e amdgpu.ids -e "^WARNING" | tee -a $OUT
done
( SEPFWI_QFUZZ_N=${QN:-80} SEPFWI_PFUZZ_N=${PN:-40} timeout -k 10 1000 python -m pytest -m gpu -x -q tests/test_gpu_quiet_skip.py tests/test_gpu_persist_fuzz.py ) > gpurun_out/r06_quiet_fuzz.txt 2>&1
tail -3 gpurun_out/r06_quiet_fuzz.txt | tee -a $OUT
