"""Deterministic, library-independent synthetic data for tests, golden vectors and bench.

Values come from an integer counter hash (splitmix64) folded into an Irwin-Hall
sum of eight 16-bit uniforms -- integer arithmetic only, so the container that
generated `tests/golden/*.npz` and the GPU box produce identical bits -- then
scaled and rounded to the nearest bf16-representable float32 (SURVEY.md §8c/d:
inputs and weights must be exactly representable in bf16 so that the fp32
oracle and the bf16 kernels see the same numbers).
"""
from __future__ import annotations

import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bf16 -> float32 (finite inputs only)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns of an already bf16-representable float32 array."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def seed_of(name: str, base: int = 20250614) -> int:
    return (base * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFF


def normal_like(shape, seed: int, std: float = 1.0, chunk: int = 1 << 22) -> np.ndarray:
    """~N(0, std^2) float32 array of bf16-representable values, a pure function of (shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    out = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            ctr = np.arange(lo, hi, dtype=np.uint64) * np.uint64(2) + np.uint64(seed) * np.uint64(0x100000001B3)
            a = _splitmix64(ctr)
            b = _splitmix64(ctr + np.uint64(1))
            acc = np.zeros(hi - lo, dtype=np.int64)
            for word in (a, b):
                for sh in (0, 16, 32, 48):
                    acc += ((word >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
            # eight uniforms on [0, 65535]: mean 8*32767.5, var 8*(65536^2-1)/12
            z = (acc.astype(np.float64) - 8 * 32767.5) / np.sqrt(8 * (65536.0 ** 2 - 1) / 12.0)
            out[lo:hi] = (z * std).astype(np.float32)
    return round_to_bf16(out).reshape(shape)


def uniform_like(shape, seed: int, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        u = (_splitmix64(ctr) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return round_to_bf16((lo + (hi - lo) * u).astype(np.float32)).reshape(shape)


def synth_state_dict(shapes: dict, tag: str = "w", weight_std: float = 0.02,
                     peaky: float = 1.0) -> dict:
    """Reference init law with the numerically-invisible parts randomised:
    Linear weights ~ N(0, .02) (trunc-normal at +-2 never binds), biases small non-zero,
    LayerNorm gamma ~ 1 + N(0,.1), beta ~ N(0,.1), alphas = 0.5, global.query ~ N(0,.02*peaky).
    `peaky` scales the attention-side projections to sharpen the softmax (golden case G10)."""
    sd = {}
    for name, shape in shapes.items():
        s = seed_of(tag + ":" + name)
        if name.endswith("_alpha"):
            v = np.full(shape, 0.5, dtype=np.float32)
        elif name.endswith(".query"):
            v = normal_like(shape, s, weight_std * peaky)
        elif "norm.weight" in name:
            v = round_to_bf16(1.0 + normal_like(shape, s, 0.1))
        elif "norm.bias" in name:
            v = normal_like(shape, s, 0.1)
        elif name.endswith(".bias"):
            v = normal_like(shape, s, 0.02)
        else:
            std = weight_std
            if peaky != 1.0 and any(k in name for k in ("q_proj", "k_proj")):
                std = weight_std * peaky
            v = normal_like(shape, s, std)
        sd[name] = v
    return sd


def outlier_channels(D: int, n: int) -> np.ndarray:
    """The `n` channels a heavy-tailed case amplifies: fixed positions spread over the row (a pure function of D and n)."""
    return (np.arange(n, dtype=np.int64) * 97 + 5) % D


def synth_inputs(T: int, h: int, w: int, D: int, tag: str = "x", guide_len: int = 0,
                 scale: float = 1.0, outliers=None) -> dict:
    """`outliers = (n, gain)`: heavy-tailed channels -- `n` fixed channels of BOTH visual tensors carry `gain` times the bulk's
    standard deviation plus a per-channel offset of the same size (the statistics of a ViT's penultimate hidden states, which is what
    frames_feature is, reference encoder.py:253-259: a few channels two orders of magnitude above the rest, with a non-zero mean);
    frames_embed = x + head MLP keeps them (encoder.py:284-286).  The guide is left alone (a pooled, normalised text embedding)."""
    ff = normal_like((T, h, w, D), seed_of(f"{tag}:ff:{T}x{h}x{w}"), scale)
    fe = normal_like((T, h, w, D), seed_of(f"{tag}:fe:{T}x{h}x{w}"), scale)
    if outliers is not None:
        n, gain = outliers
        ch = outlier_channels(D, n)
        off = normal_like((n,), seed_of(f"{tag}:outlier_mean:{n}"), scale * gain)
        for x in (ff, fe):
            x[..., ch] = round_to_bf16(x[..., ch] * np.float32(gain) + off)
    gshape = (D,) if guide_len == 0 else (guide_len, D)
    g = normal_like(gshape, seed_of(f"{tag}:g:{guide_len}"), scale)
    return dict(ff=ff, fe=fe, g=g)
