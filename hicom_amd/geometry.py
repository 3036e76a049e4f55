"""Host-side geometry of the compressor: projector-type strings, window tilings, packing layouts
and the per-axis sinusoid tables.  Pure Python / NumPy (no torch, no HIP): unit-tested on CPU.

Reference behaviour restated here:
  build_vision_projector's type-string parser          hicom/model/projector.py:246-302
  divide_feature / balance_divide_feature              hicom/model/projector.py:473-522
  post_process_visual_feature (row layout only)        hicom/mm_utils.py:92-140
  get_3d_position_embedding (per-axis, fp64)           hicom/model/projector.py:57-101
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------
# mm_projector_type
# ------------------------------------------------------------------------------------------
@dataclass
class LocalSpec:
    temporal_kernel_size: int
    spatial_kernel_size: int
    adapt_q: bool = False
    adapt_k: bool = False
    adapt_v: bool = False
    adapt_guide: bool = False
    force_use_guide: object = False     # False or a mode string, exactly like the reference


@dataclass
class GlobalSpec:
    num_queries: int
    adapt_guide: bool = False
    force_use_guide: object = False


def _leading_digits(s: str) -> str:
    m = re.match(r"\d*", s)
    return m.group(0)


def parse_mm_projector_type(projector_type: str) -> Tuple[Optional[LocalSpec], Optional[GlobalSpec]]:
    """'local43_adaptkv_global32_coarse' -> (LocalSpec(4,3,adapt_k,adapt_v), GlobalSpec(32)).

    Same substring semantics as the reference (projector.py:247-302): the local phase is the text
    between the LAST 'local' and the next 'global'; digits -> kernel sizes (1 temporal digit, then
    1 or 2 spatial digits); 'adapt' + any of q/k/v/g; 'guide<mode>' up to the next '_'."""
    local = glob = None
    if "local" in projector_type:
        phase = projector_type.split("local")[-1].split("global")[0]
        digits = _leading_digits(phase)
        if len(digits) not in (2, 3):
            raise ValueError(f"cannot read local kernel sizes from {projector_type!r}")
        local = LocalSpec(int(digits[0]), int(digits[1:]))
        if "adapt" in phase:
            for ch in phase.split("adapt")[-1]:
                if ch == "q":
                    local.adapt_q = True
                elif ch == "k":
                    local.adapt_k = True
                elif ch == "v":
                    local.adapt_v = True
                elif ch == "g":
                    local.adapt_guide = True
                else:
                    break
        if "guide" in phase:
            local.force_use_guide = phase.split("guide")[-1].split("_")[0]
    if "global" in projector_type:
        phase = projector_type.split("global")[-1].split("local")[0]
        glob = GlobalSpec(int(_leading_digits(phase)), adapt_guide="adaptg" in phase)
        if "guide" in phase:
            glob.force_use_guide = phase.split("guide")[-1].split("_")[0]
    return local, glob


def tower_dims(mm_vision_tower: str) -> Tuple[int, int]:
    """(qk_dim, grid side) of the supported towers (projector.py:407-414)."""
    if "siglip-so400m-patch14-384" in mm_vision_tower:
        return 1152, 27
    if "clip-vit-large-patch14-336" in mm_vision_tower:
        return 768, 24
    raise NotImplementedError(f"unsupported vision tower {mm_vision_tower!r}")


# ------------------------------------------------------------------------------------------
# window tiling of one axis
# ------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class AxisTiling:
    n: int        # axis length
    k: int        # window length
    nwin: int     # number of windows
    nfull: int    # windows [0, nfull) start at i*k; the rest overlap their predecessor by one

    def start(self, i: int) -> int:
        return i * self.k if i < self.nfull else self.nfull * self.k + (i - self.nfull) * (self.k - 1) - 1

    @property
    def starts(self) -> List[int]:
        return [self.start(i) for i in range(self.nwin)]

    @property
    def is_partition(self) -> bool:
        return self.n % self.k == 0


def axis_tiling(n: int, k: int) -> AxisTiling:
    """Windows of length k over n elements.

    n % k == 0 -> plain tiling.  Otherwise the reference's balance_divide_feature: ceil(n/k)
    groups of which the first (n mod groups, or all) have k fresh elements and the others k-1
    fresh elements plus the last element of the previous group.  If that does not cover the
    axis exactly the reference's torch.stack fails; we raise the same RuntimeError."""
    if n <= 0 or k <= 0:
        raise ValueError("axis_tiling: n and k must be positive")
    if n % k == 0:
        return AxisTiling(n, k, n // k, n // k)
    if n < k:   # one short window holding the whole axis (x[0:k] is just shorter; nothing to stack)
        return AxisTiling(n, n, 1, 1)
    nwin = math.ceil(n / k)
    nfull = n % nwin or nwin
    covered = nfull * k + (nwin - nfull) * (k - 1)
    if covered != n:
        raise RuntimeError("stack expects each tensor to be equal size "
                           f"(axis of {n} elements cannot be balanced into windows of {k})")
    return AxisTiling(n, k, nwin, nfull)


# ------------------------------------------------------------------------------------------
# packing layout of post_process_visual_feature
# ------------------------------------------------------------------------------------------
@dataclass
class PackLayout:
    n_tokens: int                 # t*h*w visual tokens
    n_rows: int                   # rows after packing
    nl_group: int = 0             # newline after every nl_group tokens (0 = none interleaved)
    newline_rows: List[int] = field(default_factory=list)   # absolute row ids that hold the newline

    def row_of(self, m: int) -> int:
        return m + (m // self.nl_group if self.nl_group else 0)


def pack_layout(mm_patch_merge_type: str, mm_newline_position: str, modal: str, t: int, h: int, w: int,
                has_newline: bool, is_anyres: bool) -> PackLayout:
    n = t * h * w
    if mm_patch_merge_type == "flat" or not mm_patch_merge_type.startswith("spatial"):
        return PackLayout(n, n)
    if modal == "video":
        if mm_newline_position == "grid":
            _need(has_newline)
            return PackLayout(n, t * h * (w + 1), w, [(i + 1) * (w + 1) - 1 for i in range(t * h)])
        if mm_newline_position == "frame":
            _need(has_newline)
            return PackLayout(n, t * (h * w + 1), h * w, [(i + 1) * (h * w + 1) - 1 for i in range(t)])
        if mm_newline_position == "one_token":
            _need(has_newline)
            return PackLayout(n, n + 1, 0, [n])
        if mm_newline_position == "no_token":
            return PackLayout(n, n)
        raise ValueError(f"Unexpected mm_newline_position: {mm_newline_position}")
    if modal == "image":
        if t != 1:
            raise ValueError("image modality expects a single [1, h, w, d] grid (the reference's einops "
                             "pattern '1 h w d' fails for t > 1)")
        if is_anyres:
            _need(has_newline)
            return PackLayout(n, h * (w + 1), w, [(i + 1) * (w + 1) - 1 for i in range(h)])
        if has_newline:
            return PackLayout(n, n + 1, 0, [n])
        return PackLayout(n, n)
    raise ValueError(f"post_process_visual_feature: unsupported modal {modal!r} for spatial packing")


def _need(has_newline: bool):
    if not has_newline:
        raise ValueError("this packing mode needs image_newline")


# ------------------------------------------------------------------------------------------
# positional tables
# ------------------------------------------------------------------------------------------
def sinusoid_axis_table(n: int, d_model: int) -> np.ndarray:
    """float64 [n, d_model]: sin(p / 10000^(2*(c//2)/d)) on even c, cos on odd c.

    The reference divides by np.float32(d_model) inside a float64 expression; we reproduce that
    (the float32 value of 1152 or 768 is exact, so this is the plain float64 formula)."""
    pos = np.arange(n, dtype=np.float64)[:, None]
    c = np.arange(d_model)[None, :]
    ang = pos / np.power(10000.0, (2 * (c // 2)) / float(np.float32(d_model)))
    out = np.empty_like(ang)
    out[:, 0::2] = np.sin(ang[:, 0::2])
    out[:, 1::2] = np.cos(ang[:, 1::2])
    return out


def stacked_pos_tables(t_cap: int, h: int, w: int, d_model: int) -> np.ndarray:
    """float32 [t_cap + h + w, d_model]: rows [0,t_cap) = PE_t, then PE_y, then PE_x."""
    return np.concatenate([sinusoid_axis_table(t_cap, d_model), sinusoid_axis_table(h, d_model),
                           sinusoid_axis_table(w, d_model)], axis=0).astype(np.float32)
