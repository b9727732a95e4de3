#!/bin/bash
# round 6, item 1 of the review: timing-only probe of a persistent backward loop whose phase B trails its phase A inside the tile
# (options pk_lock / pk_snake of the -DSEPFWI_PROBES build; WRONG results by design).  Kill criterion: >= 47 us per backward step.
mkdir -p gpurun_out
( timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent_backward or bit_identical or persistent_loop_leaves" ) > gpurun_out/r06_lock_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r06_lock_pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
NS=${NS:-1500}
timeout -k 10 900 python scripts/ab_bench.py --nsteps $NS --rounds 2 \
  "" "bwd_fuse=2" "pk_nosync=1" "pk_nosync=1,pk_order=0" "pk_nosync=1,pk_order=0,pk_px=1" "pk_nosync=1,pk_order=0,pk_px=1,pk_snake=0" \
  "pk_nosync=1,pk_order=0,pk_lock=6" "pk_nosync=1,pk_order=0,pk_lock=12" "pk_nosync=1,pk_order=0,pk_lock=24" "pk_nosync=1,pk_order=0,pk_lock=48" \
  "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=3" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=6" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=12" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=24" \
  "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=262" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=268" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=518" "pk_nosync=1,pk_order=0,pk_px=1,pk_lock=774" \
  "pk_nosync=1,pk_order=0,pk_px=1,pk_snake=0,pk_lock=6" "pk_nosync=1,pk_order=0,pk_px=1,pk_snake=0,pk_lock=262" "pk_nosync=1,pk_order=0,pk_px=2,pk_lock=8" "pk_nosync=1,pk_order=0,pk_px=2,pk_lock=264" \
  > gpurun_out/r06_lock_probe.log 2>&1
rc=$?
cat gpurun_out/r06_lock_probe.log | cut -c1-200
exit $rc
