#!/bin/bash
# GPU box: a selection of the -m gpu suite in one process.  usage: bash scripts/gpu_pytest.sh <log name> <pytest arguments ...>
mkdir -p gpurun_out
LOG=gpurun_out/$1; shift
( time timeout -k 10 1100 python -m pytest -m gpu -x -q "$@" ) > $LOG 2>&1
rc=$?
tail -25 $LOG | cut -c1-400
exit $rc
