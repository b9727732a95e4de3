#!/bin/bash
# round 5: wave timeline of the persistent loop (one-off build with -DSEPFWI_PK_TRACE on the GPU box; the shipped library is not traced)
mkdir -p gpurun_out
# the traced build replaces libsepfwi_probes.so (the build scripts/ab_bench.py loads): whatever happens, the normal library is rebuilt before the script ends (later runs on this
# box would otherwise measure the traced kernel under the untraced source digest)
trap 'cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."; (cd sep-2023_amd && python -c "from sepfwi import _native; _native.build(force=True, variant=\"probes\")") > gpurun_out/pk_trace_rebuild.log 2>&1' EXIT
cd sep-2023_amd && SEPFWI_HIPCC_FLAGS=-DSEPFWI_PK_TRACE python -c "from sepfwi import _native; _native.build(force=True, variant=\"probes\")" > ../gpurun_out/pk_trace_build.log 2>&1 || { tail -5 ../gpurun_out/pk_trace_build.log; exit 1; }
cd ..
OUT=gpurun_out/r05_pk_trace.txt; : > $OUT
for V in "$@"; do
  echo "== variant: $V" | tee -a $OUT
  SEPFWI_PK_TRACE=$PWD/gpurun_out/pk_trace.bin timeout -k 10 300 python scripts/ab_bench.py --nsteps 400 --rounds 1 --shots 1 "$V" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  python scripts/pk_trace.py gpurun_out/pk_trace.bin >> $OUT
done
rm -f gpurun_out/pk_trace.bin
grep -E 'variant|Gcell|phase length|per tile-phase|share of|column means' $OUT | cut -c1-260
