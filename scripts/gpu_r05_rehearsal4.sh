#!/bin/bash
# round 5: configs[4]'s rank share rehearsed on ONE card -- examples/das_fwi_2000x1000.py at full size (2000x1000, 4000 steps), 16 shots,
# 2 L-BFGS-B iterations: one rank, then four gloo ranks that all drive device 0 (launched before anything touches the GPU).
# The iterates of the two runs must agree to the printed digits (float32 block sums associate differently).
mkdir -p gpurun_out
A="--shots 16 --niter 2 --pert 0.03 --sigma-init 40"
( time timeout -k 10 500 python -u examples/das_fwi_2000x1000.py $A ) > gpurun_out/r05_e2e_16shots_1rank.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r05_e2e_16shots_1rank.log | grep "iterate\|done" | cut -c1-200
( time timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 \
    examples/das_fwi_2000x1000.py --backend gloo --share-gpu $A ) > gpurun_out/r05_e2e_16shots_4ranks.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r05_e2e_16shots_4ranks.log | grep "iterate\|done" | cut -c1-200
python - <<'PY'
import re
def its(fn):
    return [float(re.search(r"misfit ([0-9.e+-]+)", l).group(1)) for l in open(fn) if "iterate" in l and "misfit" in l]
a, b = its("gpurun_out/r05_e2e_16shots_1rank.log"), its("gpurun_out/r05_e2e_16shots_4ranks.log")
dev = max(abs(x - y) / abs(x) for x, y in zip(a, b))
print("iterates 1 rank:", a, "\niterates 4 ranks:", b, "\nlargest relative difference %.2e" % dev)
assert len(a) == len(b) >= 2 and dev <= 2e-6
PY
