"""Compiles the HIP sources into hicom_amd/libhicom_hip.so (gfx950 only, in-tree)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhicom_hip.so")


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libhicom_hip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "hicom_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    # HICOM_FORCE_BUILD=1: compile even when an up-to-date library is present (the driver's "does it build" check)
    force = force or os.environ.get("HICOM_FORCE_BUILD") == "1"
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", *extra_flags, "-o", LIB, *sources()]
    if verbose:
        print("[hicom_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True)
