#!/bin/bash
# first GPU session: parity tests, smoke, a short bench and a kernel-trace profile
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/smi.txt
nproc >> gpurun_out/smi.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1
( time timeout 600 python bench.py --steps 1 --warmup 1 --nsteps 1000 --no-cpu-baseline ) > gpurun_out/bench_short.log 2>&1
( time timeout 900 python bench.py --steps 1 --warmup 1 ) > gpurun_out/bench_full.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --nsteps 400 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof1 -name '*stats*' | head; 
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench_short.log; tail -2 gpurun_out/bench_full.log
