"""Compiles the HIP sources into hicom_amd/libhicom_hip.so (gfx950 only, in-tree).

One object per source under hicom_amd/build/ (compiled in parallel, recompiled only when the source, a shared header, the flag
list or the compiler changed: the object names carry a hash of the last two), then one link: an edit of one kernel file costs
one compile, not thirteen.  Objects and the library are written to private temporaries and renamed into place, so concurrent
builders of one tree cannot tear each other's files."""
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
# -amdgpu-kernarg-preload-count: leading SCALAR kernel arguments arrive in SGPRs at wave launch (struct arguments are not preloaded: the hot
# kernels repeat the fields their first requests need as leading scalars); a device without the feature runs the s_load prologue hipcc keeps
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm",
         "-amdgpu-kernarg-preload-count=" + os.environ.get("HICOM_KERNARG_PRELOAD", "14")]      # (0: dev A/B, tools/gpu_kernarg_ab.sh)


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


_CC_ID = {}


def _toolchain_id(cc: str) -> str:
    """`hipcc --version` (cached per process): part of the object tag, so objects of another compiler are never linked."""
    if cc not in _CC_ID:
        try:
            _CC_ID[cc] = subprocess.run([cc, "--version"], capture_output=True, text=True, timeout=60).stdout
        except Exception:
            _CC_ID[cc] = cc
    return _CC_ID[cc]


def build(force: bool = False, verbose: bool = True, extra_flags=(), lib_path: str = None) -> str:
    """extra_flags / lib_path: instrumented dev builds (tools/) go to their own library and their own objects."""
    # HICOM_FORCE_BUILD=1: compile even when an up-to-date library is present (the driver's "does it build" check)
    force = force or os.environ.get("HICOM_FORCE_BUILD") == "1"
    out = lib_path or LIB
    if not force and not extra_flags and out == LIB and not is_stale():
        return LIB
    cc = hipcc()
    flags = FLAGS + list(extra_flags)
    # the object tag covers the FULL flag list and the compiler's identity: a change of either recompiles everything
    tag = "." + hashlib.sha1((" ".join(flags) + "\n" + _toolchain_id(cc)).encode()).hexdigest()[:10]
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo = []
    for s in sources():
        o = _obj_of(s, tag)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append((s, o))

    def compile_one(so):
        # to a private temporary, then an atomic rename: several processes building the same tree (ranks of `bench.py --gpus N` on a
        # stale checkout) never see -- or link -- a half-written object
        s, o = so
        tmp = f"{o}.tmp.{os.getpid()}"
        cmd = [cc, *flags, "-c", s, "-o", tmp]
        if verbose:
            print("[hicom_amd] " + " ".join(cmd[:-1] + [o]), file=sys.stderr)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, o)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    workers = max(1, min(len(todo), (os.cpu_count() or 2) - 1, 8))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(compile_one, todo))
    tmp_out = f"{out}.tmp.{os.getpid()}"
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_out, *[_obj_of(s, tag) for s in sources()]]
    if verbose:
        print("[hicom_amd] " + " ".join(link), file=sys.stderr)
    try:
        subprocess.check_call(link)
        os.replace(tmp_out, out)
    finally:
        if os.path.exists(tmp_out):
            os.remove(tmp_out)
    return out


if __name__ == "__main__":
    build(force=True)
