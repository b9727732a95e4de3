#!/bin/bash
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -6 gpurun_out/pytest_gpu.log
