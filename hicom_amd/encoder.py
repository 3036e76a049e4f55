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


def _head_cache(head, store=True):
    """fp16 copies of fc1 / fc2 per weight state.  store=False (the backward): a miss rebuilds from the live weights WITHOUT
    stamping the result -- copies made between a forward and the optimizer step must not pass for post-step weights."""
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
        if store:
            head.__dict__["_hicom_f16"] = hit
    return hit[1], hit[2], hit[3], hit[4]


def siglip_head_embed(last_hidden_state: torch.Tensor, head, hidden_act: str = None, out_dtype=None) -> torch.Tensor:
    """[..., D] bf16 tokens -> x + head.mlp(head.layernorm(x)), same shape (bf16; out_dtype=torch.float32 for parity tests).
    With autograd on and trainable head parameters (stage 3 of the reference's script trains "vision_model_head", train.py:717-720)
    the result carries a graph: gradients of layernorm / fc1 / fc2 (`_HeadFn`); the tokens come from the frozen tower body
    (train.py:703): asking for THEIR gradient raises."""
    if torch.is_grad_enabled() and (last_hidden_state.requires_grad or any(p.requires_grad for p in _head_params(head))):
        if out_dtype not in (None, last_hidden_state.dtype):
            raise NotImplementedError("siglip_head_embed: out_dtype with autograd")
        # training: the optimizer may have rewritten fc1 / fc2 behind the version counters since the last forward (DeepSpeed's
        # bf16 optimizer, `p.data.copy_`): every training forward re-reads the live weights (native.py "weight-derived caches")
        nv.note_training_forward()
        return _HeadFn.apply(last_hidden_state, head, hidden_act, *_head_params(head))
    nv.begin_inference()
    return _head_chain(last_hidden_state, head, hidden_act, out_dtype, None, True)[0]


def _head_params(head):
    return (head.layernorm.weight, head.layernorm.bias, head.mlp.fc1.weight, head.mlp.fc1.bias, head.mlp.fc2.weight, head.mlp.fc2.bias)


LAST_FP32_GRADS = None      # test hook: the fp32 gradients of the last head backward (before the cast to the parameter dtype)


class _HeadFn(torch.autograd.Function):
    """Forward = the inference chain (same kernels, same bits).  Backward is recompute-based (the reference's scripts run with
    gradient checkpointing): LayerNorm output and the pre-activation hidden layer come back from the HIP forward kernels
    (hicom_ln_stream_fwd, hicom_dense16_gemm_fwd without activation); d hidden = dY . W2 runs on the dense16 kernel as well (NT
    form); the two weight gradients dW = dY^T X contract over the TOKEN axis -- the TN form of the same kernel
    family (hicom_dense16_tn_fwd, round 4: transposed LDS fragment reads, split contraction; bf16 operands, fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, head, hidden_act, *params):
        with torch.no_grad():
            keep = {} if getattr(head, "keep_hidden_for_backward", True) else None
            out = _head_chain(x, head, hidden_act, None, None, True, keep=keep)[0]
        ctx.head, ctx.hidden_act = head, hidden_act
        # the pre-activation hidden layer (second output of fc1's epilogue: 0.4 GB fp16 at 64 frames) stays for the backward, as autograd
        # keeps it in the reference; `head.keep_hidden_for_backward = False` recomputes it instead (gradient checkpointing)
        if keep is not None and "h1" in keep:
            ctx.save_for_backward(x, keep["h1"])
        else:
            ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        global LAST_FP32_GRADS
        x, h1 = ctx.saved_tensors if len(ctx.saved_tensors) == 2 else (ctx.saved_tensors[0], None)
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("siglip_head_embed backward: the gradient w.r.t. the tower's hidden states is not built (the "
                                      "tower body is frozen in every stage of the reference's script, train.py:703); detach them")
        head = ctx.head
        ln, fc1, fc2 = head.layernorm, head.mlp.fc1, head.mlp.fc2
        hidden_act = ctx.hidden_act or getattr(getattr(head.mlp, "config", None), "hidden_act", "gelu_pytorch_tanh")
        with torch.no_grad():
            D = x.shape[-1]
            x2 = x.contiguous().view(-1, D)
            M = x2.shape[0]
            w1, w2, kpad, ld = _head_cache(head, store=False)
            inter = fc1.weight.shape[0]
            dY = d_out.contiguous().view(M, D)
            # ---- the pre-activation hidden layer: kept by the forward, else recomputed (HIP) ------------------------------------
            if h1 is None:
                n16 = torch.empty((M, D), dtype=torch.float16, device=x.device)
                nv.ln_stream(x2, ln.weight.detach(), ln.bias.detach(), n16, eps=ln.eps)
                h1 = torch.empty((M, ld), dtype=torch.float16, device=x.device)
                nv.dense16_gemm(n16, w1, fc1.bias.detach(), act=nv.ACT_NONE, out_f16=h1, n_store=kpad)
                del n16
            # the [M, inter] tensors (0.4 GB each in bf16 at 64 frames) are touched as few times as possible: one cast of the
            # pre-activation, one fused GELU, one fused GELU-backward (fp32 math inside, bf16 in / out), fp32 only in the reductions
            approx = "tanh" if hidden_act == "gelu_pytorch_tanh" else "none"
            hip_act = inter % 8 == 0 and hidden_act in _ACTS and _ACTS[hidden_act] in (nv.ACT_GELU, nv.ACT_GELU_TANH)
            if hip_act:
                a_b = nv.act_rows(h1, inter, _ACTS[hidden_act])             # HIP: GELU of the pitched fp16 rows -> dense bf16
                h1b = None
            else:
                h1b = h1[:, :inter].to(torch.bfloat16)
                a_b = torch.nn.functional.gelu(h1b, approximate=approx)
            # ---- fc2: dW2 = dY^T a, db2, d a = dY W2 --------------------------------------------------------------------------
            dYb = dY.to(torch.bfloat16)
            grads = {}
            grads["mlp.fc2.weight"] = _tn_f32(dYb, a_b)
            grads["mlp.fc2.bias"] = dY.sum(0, dtype=torch.float32)
            del a_b
            w2t = fc2.weight.detach().t().contiguous()                      # [inter, D] bf16: the NT form's "weight"
            if D % 64 == 0 and inter % 8 == 0:
                da = torch.empty((M, inter), dtype=torch.bfloat16, device=x.device)
                nv.dense16_gemm(dYb, w2t, None, y=da)                        # HIP: bf16 x bf16, fp32 accumulate
            else:
                da = torch.mm(dYb, fc2.weight.detach())
            # ---- activation, fc1 and the LayerNorm affine from ONE token contraction -------------------------------------------
            # With n = nhat gamma + beta and G = dh1^T nhat ([inter, D], the TN GEMM against the NORMALISED tokens):
            #   dW1 = G gamma + db1 (x) beta,   d gamma = sum_j W1[j, :] G[j, :],   d beta = W1^T db1
            # -- the gradients of the LayerNorm affine need neither d n = dh1 W1 (a 0.46-TFLOP GEMM for 2 x 1152 numbers) nor an
            # fp32 recomputation of nhat over all tokens.
            dh1b = nv.act_bwd_rows_(da, h1, _ACTS[hidden_act]) if hip_act else torch.ops.aten.gelu_backward(da, h1b, approximate=approx)
            del da, h1b, h1
            ones, zeros = torch.ones(D, dtype=torch.bfloat16, device=x.device), torch.zeros(D, dtype=torch.bfloat16, device=x.device)
            nhat = torch.empty((M, D), dtype=torch.bfloat16, device=x.device)
            nv.ln_stream(x2, ones, zeros, nhat, eps=ln.eps)                  # HIP: normalised tokens, no affine
            G = _tn_f32(dh1b, nhat)
            db1 = dh1b.sum(0, dtype=torch.float32)
            del dh1b, nhat
            gamma, beta = ln.weight.detach().float(), ln.bias.detach().float()
            w1f = fc1.weight.detach().float()
            grads["mlp.fc1.weight"] = G * gamma[None, :] + db1[:, None] * beta[None, :]
            grads["mlp.fc1.bias"] = db1
            grads["layernorm.weight"] = (w1f * G).sum(0)
            grads["layernorm.bias"] = (w1f * db1[:, None]).sum(0)
        LAST_FP32_GRADS = grads
        names = ("layernorm.weight", "layernorm.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
        plist = _head_params(head)
        outs = [grads[n].to(p.dtype).view(p.shape) if ctx.needs_input_grad[3 + i] else None for i, (n, p) in enumerate(zip(names, plist))]
        return (None, None, None, *outs)


def _tn_f32(a, b):
    """a^T @ b over the token rows, bf16 operands -> fp32: the TN form of the dense MFMA GEMM (hicom_dense16_tn_fwd) where the
    shapes allow (widths multiples of 8), else the library."""
    if a.shape[0] >= 64 and a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0 and a.is_contiguous() and b.is_contiguous():
        return nv.dense16_tn(a, b)
    return _mm_f32(a.t(), b)


def _mm_f32(a, b):
    """a @ b for bf16 operands with an fp32 result (vendor GEMM: fp32 accumulation; out_dtype where the build has it)."""
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.mm(a.float(), b.float())


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
    nv.begin_inference()
    out, logits = _head_chain(last_hidden_state, head, hidden_act, None, guide_embed.contiguous(), return_embed)
    return (logits, out) if return_embed else logits


def _head_chain(last_hidden_state, head, hidden_act, out_dtype, guide, want_embed, keep=None):
    from .projector import _require_bf16_cuda
    x = last_hidden_state
    _require_bf16_cuda("last_hidden_state", x)
    if guide is not None and torch.is_grad_enabled() and any(p.requires_grad for p in _head_params(head)):
        raise RuntimeError("siglip_head_scores builds no autograd graph (inference path): call it under torch.no_grad(); "
                           "siglip_head_embed is the differentiable form")
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
    pre = None
    if keep is not None and fc1.weight.shape[0] % 8 == 0:              # training forward: fc1's value before the activation as well
        pre = keep["h1"] = torch.empty((M, ld), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(a16, w1, fc1.bias.detach(), act=_ACTS[hidden_act], out_f16=hid, n_store=kpad, pre_f16=pre)
    out = torch.empty((M, D), dtype=out_dtype or x.dtype, device=x.device) if want_embed else None
    parts = torch.empty(((D + 63) // 64, M), dtype=torch.float32, device=x.device) if guide is not None else None
    nv.dense16_gemm(hid, w2, fc2.bias.detach(), N=D, K=kpad, y=out, res=x2, row_dot=(guide, parts) if guide is not None else None)
    logits = None
    if guide is not None:
        logits = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        nv.partials_sum(parts, logits.view(-1))
    return (out.view(x.shape) if want_embed else None), logits
