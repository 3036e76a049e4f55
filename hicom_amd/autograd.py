"""Training path of the drop-in projector: autograd through hicom_compressor_fwd.

The projector is the trainable module of the reference's stages 1-2 (scripts/qwen2.5_7B/release/
directg_local43_global32.sh:54,113; hicom/train.py:704-712), called under autograd + gradient checkpointing.
"""
from __future__ import annotations

import torch


def forward_with_grad(proj, frames_feature, frames_embed, guide_embed, modal, image_newline):
    from .projector import _require_bf16_cuda
    some = frames_feature["patch"] if isinstance(frames_feature, dict) else frames_feature
    _require_bf16_cuda("frames_feature", some)
    raise NotImplementedError("hicom_amd: no backward for this recipe yet -- run inference under torch.no_grad() / "
                              "torch.inference_mode(); forward() never returns a silently detached tensor")
