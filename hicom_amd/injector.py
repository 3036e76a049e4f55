"""Host orchestration of the instruction injector and the q/k/v adaptors on the HIP kernels
(reference projector.py:315-397 GuideInjector, :431-457 + :533-541 adaptors).

Every function enqueues C-ABI kernels on the current stream and returns device tensors; there is no
PyTorch arithmetic here.  Small row counts use the wave-per-column linear kernel, token-sized row
counts use the MFMA GEMMs.
"""
from __future__ import annotations

import torch

from . import native as nv


def _f32(shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


def _wb(lin):
    w = lin.weight.detach()
    if not w.is_cuda or w.dtype != torch.bfloat16:
        raise NotImplementedError("hicom_amd: projector weights must be bfloat16 on the GPU")
    return w, (lin.bias.detach() if lin.bias is not None else None)


def linear_rows(x, lin, act=nv.ACT_NONE, res=None):
    """y = act(x @ W^T + b) (+ res) for x [M, K] (bf16 or f32) -> f32 [M, N]; any M."""
    w, b = _wb(lin)
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    M = x2.shape[0]
    y = _f32((M, w.shape[0]), x2.device)
    if M <= 64 or x2.dtype != torch.float32 or res is not None or w.shape[1] % 64:
        nv.linear(x2, w, b, y, res=res, act=act)
    else:
        nv.readout_gemm(x2, w, b, y, act=act)            # MFMA path, fp32 activations split on the fly
    return y


def mlp2_rows(x, mlp, out_features=None):
    """build_mlp(depth 2): Linear -> GELU -> Linear on a few rows (ref :307-312)."""
    return linear_rows(linear_rows(x, mlp[0], act=nv.ACT_GELU), mlp[2])


def adapted_guide(inj, guide):
    """(1 - a) g + a LN(MLP(g)) when the injector has adapt_guide (ref :365 / :389); else the guide."""
    if not isinstance(inj.text2qk_proj, torch.nn.Identity):      # text_dim != qk_dim (ref :323-326, :364)
        guide = mlp2_rows(guide, inj.text2qk_proj).reshape(*guide.shape[:-1], -1)
    if isinstance(inj.guide_alpha, (int, float)):
        return guide
    g = guide.reshape(-1, guide.shape[-1]).contiguous()
    h = mlp2_rows(g, inj.guide_proj)
    out = _f32(g.shape, g.device)
    nv.row_ln(h, inj.guide_norm, out, src=g, alpha=inj.guide_alpha.detach())
    return out.reshape(guide.shape)


def inject(inj, mode, visual, guide, out=None):
    """GuideInjector.forward for `visual` [M, D] (the pooled / learnable queries).

    Returns (query, shared): shared=True means one row that every position uses ("direct").  `out` (f32 [M, D]): where the coarse /
    fine result is written (the engine's plan-owned query rows); other modes ignore it."""
    if mode in (None, "off"):
        return visual, False
    if mode == "direct":
        if guide.ndim != 1:
            raise ValueError("direct guide injection takes a [D] guide embedding")
        return adapted_guide(inj, guide).reshape(1, -1), True
    D = visual.shape[-1]
    vis = visual.reshape(-1, D).contiguous()
    if out is None or tuple(out.shape) != tuple(vis.shape) or out.dtype != torch.float32:
        out = _f32(vis.shape, vis.device)
    if mode == "coarse":
        if guide.ndim != 1:
            raise ValueError("coarse guide injection takes a [D] guide embedding")
        g = adapted_guide(inj, guide).reshape(1, -1)
        cs = mlp2_rows(g, inj.coarse_proj)                      # [1, 2D]: FiLM (scale | shift)  (ref :370-371)
        nv.row_ln(vis, inj.coarse_norm, out, mul=cs[:, :D], add=cs[:, D:])      # LN(v * (1 + scale) + shift)  (:372)
        return out, False
    if mode == "fine":
        if guide.ndim != 2:
            raise ValueError("fine guide injection takes an [L, D] guide embedding")
        g = adapted_guide(inj, guide)
        att = inj.fine_proj
        qp = linear_rows(vis, att.q_proj)
        kp = linear_rows(g, att.k_proj)
        vp = linear_rows(g, att.v_proj)
        ao = _f32(vis.shape, vis.device)
        nv.small_mha(qp, kp, vp, att.num_heads, ao)             # softmax(q k^T / sqrt(hd)) v over the L tokens (:391)
        o = linear_rows(ao, att.out_proj)
        nv.row_ln(vis, inj.fine_norm, out, add=o)               # LN(q + attn)  (:392)
        return out, False
    raise NotImplementedError(f"use_guide={mode!r}")


def _f16_weight(lin):
    """fp16 copy of an nn.Linear weight (nv.f16_weight_copy: exact above 2^-14, range checked when first built), cached on
    the module per weight state (nv.weight_stamp)."""
    w = lin.weight
    stamp = nv.weight_stamp(w)
    hit = lin.__dict__.get("_hicom_f16")
    if hit is None or hit[0] != stamp:
        # refreshed IN PLACE where the old copy fits (same device, same shape): executor plans hold its address
        old = hit[1] if (hit is not None and hit[1].device == w.device) else None
        hit = (stamp, nv.f16_weight_copy(w, out=old))
        lin.__dict__["_hicom_f16"] = hit
    return hit[1]


def adapt_stream_y(x, mlp):
    """MLP(x) over ALL tokens (the two dense GEMMs of adapt_k / adapt_v, ref :533-534): x bf16 [T,h,w,D] -> fp16 [N, D].  The
    LayerNorm and the alpha blend that follow are fused into the window-attention kernel's row loads (hicom_local_attn_adapt_fwd)."""
    D = x.shape[-1]
    x2 = x.reshape(-1, D).contiguous()
    N = x2.shape[0]
    w0, b0 = _wb(mlp[0])
    w2, b2 = _wb(mlp[2])
    if w0.shape[1] % 64 or w2.shape[1] % 64:
        raise NotImplementedError("adapt_stream: widths must be multiples of 64")
    hid = torch.empty((N, w0.shape[0]), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(x2, w0, b0, act=nv.ACT_GELU, out_f16=hid)
    y = torch.empty((N, w2.shape[0]), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(hid, _f16_weight(mlp[2]), b2, out_f16=y)
    return y


def adapt_stream(x, mlp, norm, alpha):
    """(1 - a) x + a LN(MLP(x)) over ALL tokens (adapt_k / adapt_v, ref :533-534): x bf16 [T,h,w,D] -> fp16 [T,h,w,D].
    Two dense MFMA GEMMs (hicom_dense16_gemm_fwd: raw tokens x bf16 weights, then fp16 hidden x fp16 weights) and the
    vectorised LayerNorm blend (hicom_ln_stream_fwd); no fp32 [N, D] stream is written -- the adapted stream travels as ONE
    fp16 plane (11 significand bits) into the window-attention kernel."""
    D = x.shape[-1]
    x2 = x.reshape(-1, D).contiguous()
    N = x2.shape[0]
    w0, b0 = _wb(mlp[0])
    w2, b2 = _wb(mlp[2])
    if w0.shape[1] % 64 or w2.shape[1] % 64:
        raise NotImplementedError("adapt_stream: widths must be multiples of 64")
    hid = torch.empty((N, w0.shape[0]), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(x2, w0, b0, act=nv.ACT_GELU, out_f16=hid)
    y = torch.empty((N, w2.shape[0]), dtype=torch.float16, device=x.device)
    nv.dense16_gemm(hid, _f16_weight(mlp[2]), b2, out_f16=y)
    out = torch.empty((N, D), dtype=torch.float16, device=x.device)
    nv.ln_stream(y, norm.weight.detach(), norm.bias.detach(), out, src=x2, alpha=alpha.detach(), eps=norm.eps)
    return out.reshape(x.shape)


def adapt_query(q, proj, norm, alpha):
    """(1 - a) q + a LN(q W^T)  (adapt_q, Linear without bias, ref :433,:541)."""
    q2 = q.reshape(-1, q.shape[-1]).contiguous()
    h = linear_rows(q2, proj)
    out = _f32(q2.shape, q2.device)
    nv.row_ln(h, norm, out, src=q2, alpha=alpha.detach())
    return out.reshape(q.shape)
