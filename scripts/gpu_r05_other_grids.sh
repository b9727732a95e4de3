#!/bin/bash
# round 5: the balanced persistent loop on other grid shapes, and where the batched launches (option batch = 2: by grid size) stand against it
mkdir -p gpurun_out
OUT=gpurun_out/r05_other_grids.txt; : > $OUT
run() { echo "== $1" | tee -a $OUT; shift; timeout 600 python scripts/ab_bench.py --nsteps 1000 --rounds 3 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $OUT; }
run "2000x500 (configs[1] shape), fwd+adj" --nz 500 "" "batch=0,pk_prio=0,pk_wx=100,pk_wxp=100,pk_wz=100" "batch=0,bwd_fuse=2" "batch=1"
run "1000x1500" --nz 1500 --nx 1000 "" "pk_prio=0,pk_wx=100,pk_wxp=100,pk_wz=100" "bwd_fuse=2"
run "1000x500" --nz 500 --nx 1000 "" "batch=0" "batch=0,bwd_fuse=2"
run "3000x1000" --nz 1000 --nx 3000 "" "pk_prio=0,pk_wx=100,pk_wxp=100,pk_wz=100" "bwd_fuse=2"
run "1500x500" --nz 500 --nx 1500 "" "batch=0"
run "1000x700" --nz 700 --nx 1000 "" "batch=0"
run "2000x300" --nz 300 --nx 2000 "" "batch=0"
# batched launches on one / two / three streams (option batch_split) with batches of realistic size
run2() { echo "== $1" | tee -a $OUT; shift; timeout 600 python scripts/ab_bench.py --nsteps 600 --rounds 2 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $OUT; }
run2 "1000x500 x 12 shots" --nz 500 --nx 1000 --shots 12 "batch_split=1" "" "batch_split=3"
run2 "500x250 x 16 shots" --nz 250 --nx 500 --shots 16 "batch_split=1" "" "batch_split=3"
run2 "200x100 x 19 shots" --nz 100 --nx 200 --shots 19 "batch_split=1" "" "batch_split=3"
