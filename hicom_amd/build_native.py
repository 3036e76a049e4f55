"""Compiles the HIP sources into hicom_amd/libhicom_hip.so (gfx950 only, in-tree).

One object per source under hicom_amd/build/ (compiled in parallel, recompiled only when the source, a shared header or the
flags changed), then one link: an edit of one kernel file costs one compile, not thirteen."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhicom_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libhicom_hip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "hicom_hip.h")]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def _obj_of(src: str, tag: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + tag + ".o")


def build(force: bool = False, verbose: bool = True, extra_flags=(), lib_path: str = None) -> str:
    """extra_flags / lib_path: instrumented dev builds (tools/) go to their own library and their own objects."""
    # HICOM_FORCE_BUILD=1: compile even when an up-to-date library is present (the driver's "does it build" check)
    force = force or os.environ.get("HICOM_FORCE_BUILD") == "1"
    out = lib_path or LIB
    if not force and not extra_flags and out == LIB and not is_stale():
        return LIB
    cc = hipcc()
    flags = FLAGS + list(extra_flags)
    tag = "" if not extra_flags else "." + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo = []
    for s in sources():
        o = _obj_of(s, tag)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append((s, o))

    def compile_one(so):
        s, o = so
        cmd = [cc, *flags, "-c", s, "-o", o]
        if verbose:
            print("[hicom_amd] " + " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    workers = max(1, min(len(todo), (os.cpu_count() or 2) - 1, 8))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(compile_one, todo))
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[_obj_of(s, tag) for s in sources()]]
    if verbose:
        print("[hicom_amd] " + " ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    build(force=True)
