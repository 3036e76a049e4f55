"""Visual-token packing: drop-in for `hicom/mm_utils.py:92-140` (post_process_visual_feature).

Inside HIComProjector the packing is fused into the readout GEMM's store (row map
`m -> m + m // nl_group`); this standalone function keeps the reference's call signature for
callers that pack an already-projected `[t, h, w, d]` tensor (ref hicom_arch.py:197-208), and does
the same row placement with the HIP row-scatter kernel.
"""
from __future__ import annotations

import torch

from . import geometry as geo
from . import native as nv


def post_process_visual_feature(config, visual_feature, modal, image_newline, is_anyres):
    if visual_feature.ndim != 4:
        raise ValueError("visual_feature must be [t, h, w, d]")
    t, h, w, d = visual_feature.shape
    lay = geo.pack_layout(getattr(config, "mm_patch_merge_type", "flat"),
                          getattr(config, "mm_newline_position", "one_token"),
                          modal, t, h, w, image_newline is not None, is_anyres)
    src = visual_feature.contiguous().view(t * h * w, d)
    if lay.n_rows == lay.n_tokens:
        return src
    out = torch.empty((lay.n_rows, d), dtype=visual_feature.dtype, device=visual_feature.device)
    nv.scatter_rows(src, out, 0, lay.n_tokens, nl_group=lay.nl_group)
    first = lay.newline_rows[0]
    step = lay.newline_rows[1] - first if len(lay.newline_rows) > 1 else 1
    nv.scatter_rows(image_newline.contiguous().view(1, -1), out, first, len(lay.newline_rows), row_step=step)
    return out
