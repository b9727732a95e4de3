"""The C ABI used from plain C (examples/c_api_demo.c): compile with gcc against include/sepfwi.h, link libsepfwi.so, run on the
GPU.  No Python, no torch in that process -- the boundary really is `extern "C"`, plain pointers and sizes."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_plain_c_program_through_the_c_abi(tmp_path, hip_ops):
    lib_dir = os.path.join(ROOT, "sep-2023_amd")
    exe = str(tmp_path / "c_api_demo")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_api_demo.c"),
                           "-L", lib_dir, "-lsepfwi", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe])
    out = subprocess.run([exe, str(tmp_path / "work")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout
    assert "expected failure reported" in out.stdout
