#!/bin/bash
# GPU box: the -m gpu suite (one process), then smoke()
mkdir -p gpurun_out
( time timeout -k 10 1100 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -8 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] && ( timeout -k 10 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1 && tail -2 gpurun_out/smoke.log
exit $rc
