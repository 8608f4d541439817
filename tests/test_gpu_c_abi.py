"""`-m gpu`: a stand-alone C++ client of the C-ABI (tests/c_abi/client.cpp) - no Python, no torch in the process: it is
compiled against include/afm_hip.h, linked to libafm_hip.so and run as its own executable."""
import os
import shutil
import subprocess

import pytest

from afm import ffi
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_standalone_c_client(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    libdir = os.path.dirname(ffi.lib_path())
    exe = str(tmp_path / "afm_client")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "client.cpp"),
           "-L", libdir, "-lafm_hip", f"-Wl,-rpath,{libdir}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C-ABI client OK" in r.stdout
