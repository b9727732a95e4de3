#!/bin/bash
# round 4: which synthetic inverse problem lets SciPy's L-BFGS-B (the reference's options, maxls = 6) complete ten iterations?
mkdir -p gpurun_out
for v in "--pert 0.1 --sigma-init 40" "--pert 0.03 --sigma-init 40" "--pert 0.1 --sigma-init 15" "--pert 0.03 --sigma-init 15"; do
  echo "=== $v" >> gpurun_out/r04_e2e_probe.log
  timeout -k 10 420 python examples/das_fwi_2000x1000.py --shots 12 --niter 10 $v 2>&1 | grep -v amdgpu.ids | grep "iterate\|optimizer\|done" >> gpurun_out/r04_e2e_probe.log || exit 1
done
cat gpurun_out/r04_e2e_probe.log
