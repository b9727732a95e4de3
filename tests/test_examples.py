"""The end-to-end driver of examples/ keeps running (-m gpu): a small grid, three shots, three L-BFGS-B iterations through the
device-resident chain, observed data straight into the HBM store."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_headline_example_runs_on_a_small_grid():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "das_fwi_2000x1000.py"), "--nz", "120", "--nx", "200", "--nsteps", "400",
                          "--shots", "3", "--niter", "3"], capture_output=True, text=True, timeout=500, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    its = [ln for ln in lines if ln.startswith("iterate ")]
    assert len(its) >= 2 and any(ln.startswith("done: ") for ln in lines), out.stdout[-2000:]
    f = [float(ln.split("misfit")[1].split()[0]) for ln in its]
    assert all(b <= a for a, b in zip(f, f[1:])) and f[-1] < f[0]          # the misfit goes down
    assert any(ln.startswith("split of the run") for ln in lines)
