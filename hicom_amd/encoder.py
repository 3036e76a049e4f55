"""MI355X-native form of the step right in front of the compressor: the per-patch SigLIP
pooling-head projection that produces `frames_embed`,

    image_embeds = head.layernorm(last_hidden_state)
    image_embeds = last_hidden_state + head.mlp(image_embeds)                  (reference hicom/model/encoder.py:284-286)

over all T*729 tokens: a LayerNorm and a 1152 -> 4304 -> 1152 MLP (925 GFLOP at 64 frames -- nine times the flops of the
whole compressor), i.e. matrix-core work.  `head` is whatever the reference hands over: HF's
SiglipMultiheadAttentionPoolingHead (or any object with .layernorm = nn.LayerNorm and .mlp.fc1 / .mlp.fc2 = nn.Linear).

HIP path: hicom_ln_stream_fwd (LayerNorm -> fp16, 16-byte accesses) -> hicom_dense16_gemm_fwd (fc1 + tanh-GELU -> fp16
hidden, K zero-padded to 4352) -> hicom_dense16_gemm_fwd (fc2 + bias + residual -> bf16).  fp16 operands: the normalised
activations keep 11 significand bits (LayerNorm output and tanh-GELU hidden: far inside the fp16 range whatever outlier
channels the raw hidden states carry -- the residual is added from the bf16 input in fp32); the bf16 weights convert exactly
above 2^-14 (nv.f16_weight_copy: range checked when the copy is first built; cached per weight state).

`siglip_head_scores` is the same chain for the release recipe (use_guide = "direct"), where frames_embed enters the compressor
ONLY through the local logit frames_embed_n . guide (reference projector.py:542-551 with the guide as the shared query, :352-368):
the fc2 launch dots its fp32 rows (value + residual) with the guide in its epilogue and frames_embed is never written.
"""
from __future__ import annotations

import torch

from . import native as nv

_ACTS = {"gelu_pytorch_tanh": nv.ACT_GELU_TANH, "gelu": nv.ACT_GELU}


def _head_cache(head):
    fc1, fc2 = head.mlp.fc1, head.mlp.fc2
    stamp = nv.weight_stamp(fc1.weight, fc2.weight)
    hit = head.__dict__.get("_hicom_f16")
    if hit is None or hit[0] != stamp:
        kpad = (fc2.weight.shape[1] + 63) // 64 * 64
        # row pitch of the hidden activations and of fc2's weight copy: K + 192 elements.  At a pitch of 4 352 elements (68 lines of
        # 128 B) the 128 rows of an operand tile fall on a quarter of the L2 channels at every K step: 780 -> 915 TFLOP/s on
        # fc2 with the padded pitch (tools/dense_ld.py); K = 1 152 shows no such effect
        ld = kpad + (192 if kpad > 2048 else 0)
        hit = (stamp, nv.f16_weight_copy(fc1.weight), nv.f16_weight_copy(fc2.weight, ld), kpad, ld)
        head.__dict__["_hicom_f16"] = hit
    return hit[1], hit[2], hit[3], hit[4]


def siglip_head_embed(last_hidden_state: torch.Tensor, head, hidden_act: str = None, out_dtype=None) -> torch.Tensor:
    """[..., D] bf16 tokens -> x + head.mlp(head.layernorm(x)), same shape (bf16; out_dtype=torch.float32 for parity tests)."""
    return _head_chain(last_hidden_state, head, hidden_act, out_dtype, None, True)[0]


def siglip_head_scores(last_hidden_state: torch.Tensor, head, guide_embed: torch.Tensor, hidden_act: str = None,
                       return_embed: bool = False):
    """[..., D] bf16 tokens, guide [D] bf16 -> fp32 [...] raw local logits  guide . (x_n + head.mlp(head.layernorm(x_n)))
    (the dot products of reference projector.py:551 before the 1/sqrt(D) scale, which the compressor applies), for
    `HIComProjector.forward(frames_feature, None, guide, modal, local_logits=...)`.  Nothing of frames_embed reaches HBM
    unless return_embed=True (then: (logits, frames_embed bf16), e.g. to cross-check)."""
    from .projector import _require_bf16_cuda
    _require_bf16_cuda("guide_embed", guide_embed)
    if guide_embed.ndim != 1 or guide_embed.shape[0] != last_hidden_state.shape[-1]:
        raise ValueError("siglip_head_scores takes the [D] guide embedding of use_guide='direct'")
    out, logits = _head_chain(last_hidden_state, head, hidden_act, None, guide_embed.contiguous(), return_embed)
    return (logits, out) if return_embed else logits


def _head_chain(last_hidden_state, head, hidden_act, out_dtype, guide, want_embed):
    from .projector import _require_bf16_cuda
    x = last_hidden_state
    _require_bf16_cuda("last_hidden_state", x)
    if torch.is_grad_enabled() and any(p.requires_grad for p in head.parameters()):
        raise RuntimeError("siglip_head_embed builds no autograd graph (the head trains only in stage 3 of the reference): call "
                           "it under torch.no_grad()")
    ln, fc1, fc2 = head.layernorm, head.mlp.fc1, head.mlp.fc2
    for t in (ln.weight, ln.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias):
        _require_bf16_cuda("head parameter", t)
    if hidden_act is None:
        hidden_act = getattr(getattr(head.mlp, "config", None), "hidden_act", "gelu_pytorch_tanh")
    if hidden_act not in _ACTS:
        raise NotImplementedError(f"siglip_head_embed: hidden_act={hidden_act!r}")
    D = x.shape[-1]
    x2 = x.contiguous().view(-1, D)
    M = x2.shape[0]
    w1, w2, kpad, ld = _head_cache(head)
    a16 = torch.empty((M, D), dtype=torch.float16, device=x.device)
    nv.ln_stream(x2, ln.weight.detach(), ln.bias.detach(), a16, eps=ln.eps)
    hid = torch.empty((M, ld), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(a16, w1, fc1.bias.detach(), act=_ACTS[hidden_act], out_f16=hid, n_store=kpad)
    out = torch.empty((M, D), dtype=out_dtype or x.dtype, device=x.device) if want_embed else None
    parts = torch.empty(((D + 63) // 64, M), dtype=torch.float32, device=x.device) if guide is not None else None
    nv.dense16_gemm(hid, w2, fc2.bias.detach(), N=D, K=kpad, y=out, res=x2, row_dot=(guide, parts) if guide is not None else None)
    logits = None
    if guide is not None:
        logits = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        nv.partials_sum(parts, logits.view(-1))
    return (out.view(x.shape) if want_embed else None), logits
