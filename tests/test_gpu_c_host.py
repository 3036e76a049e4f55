"""The C ABI without Python or torch in the loop (SURVEY.md §8 row b): a C++ host program that includes include/hicom_hip.h, links
libhicom_hip.so, calls an operator on hipMalloc'd buffers and checks it against its own double-precision loop."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_links_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = os.path.join(ROOT, "hicom_amd")
    assert os.path.exists(os.path.join(lib_dir, "libhicom_hip.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "host_smoke")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_host", "host_smoke.cpp"), "-L", lib_dir, "-lhicom_hip", "-o", exe],
                   check=True, timeout=600)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C-HOST OK" in out.stdout, out.stdout + out.stderr
