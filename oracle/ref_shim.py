"""Import shim for the PUBLIC reference at /root/reference (build container only).

TEST INFRASTRUCTURE.  Used by `tests/golden/make_golden.py` (to emit golden vectors) and by
`tests/test_oracle_vs_reference.py` (skipped where /root/reference is absent, e.g. the GPU
box).  Nothing from the reference is copied: the package is imported in place.

Why a shim: `import hicom` pulls cv2/decord/imageio/moviepy (absent) through
`hicom/__init__.py`, and `projector.py` imports `transformers.TRANSFORMERS_CACHE`, which
transformers 5.x dropped.  We register empty stand-in *modules* for those unrelated video-IO
packages, restore the constant, and pre-register bare `hicom` / `hicom.model` package objects
so the package `__init__`s (which import the whole training stack) do not run.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("HICOM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "hicom", "model", "projector.py"))


def load():
    """Returns (projector_module, mm_utils_module) of the reference."""
    if not available():
        raise FileNotFoundError(REFERENCE_ROOT)
    if "hicom.model.projector" in sys.modules:
        return sys.modules["hicom.model.projector"], sys.modules["hicom.mm_utils"]

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
        return sys.modules[name]

    stub("cv2")
    stub("imageio")
    stub("decord", VideoReader=object, cpu=lambda *a, **k: None)
    stub("moviepy")
    stub("moviepy.editor", VideoFileClip=object)
    import transformers
    if not hasattr(transformers, "TRANSFORMERS_CACHE"):
        transformers.TRANSFORMERS_CACHE = "/tmp/hicom_ref_cache"

    pkg = types.ModuleType("hicom")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "hicom")]
    sys.modules["hicom"] = pkg
    sub = types.ModuleType("hicom.model")
    sub.__path__ = [os.path.join(REFERENCE_ROOT, "hicom", "model")]
    sys.modules["hicom.model"] = sub
    proj = importlib.import_module("hicom.model.projector")
    mmu = importlib.import_module("hicom.mm_utils")
    return proj, mmu
