"""MI355X-native drop-in for the reference's `hicom/model/projector.py` (HICom's hybrid-level,
instruction-injected video-token compressor).

Same public surface as the reference, so `hicom/model/hicom_arch.py:44,97,212` can use it unchanged:

    build_vision_projector(config, delay_load=False, **kw) -> nn.Module         (ref :231-304)
    HIComProjector.forward(frames_feature, frames_embed, guide_embed, modal,
                           image_newline=None) -> Tensor[n_tok, hidden]          (ref :676-708)
    LocalCompressor / GlobalCompressor / GuideInjector / MultiheadAttention      (ref :133-646)

and the same sub-module / parameter names, i.e. the same state-dict schema, so `mm_projector.bin`
and full checkpoints load with `load_state_dict` exactly as before (ref hicom_arch.py:107-128).

What differs is everything below the signatures: the modules are parameter containers, and the
forward pass is a short sequence of hand-written HIP kernels for gfx950, called
through the C ABI in include/hicom_hip.h.  There is no PyTorch compute fallback: without
libhicom_hip.so, or on CPU tensors, forward raises.
"""
from __future__ import annotations

import math
import re
from functools import partial
from typing import Dict, Tuple

import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from . import geometry as geo
from . import native as nv

_NATIVE_GUIDE_MODES = (None, "off", "direct")


def _init_like_reference(m: nn.Module):
    """Linear: trunc_normal(std=.02), zero bias; LayerNorm: (1, 0)  (ref :155-164, 462-471, 623-632)."""
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


def build_mlp(depth: int, hidden_size: int, output_hidden_size: int) -> nn.Sequential:
    """Linear -> (GELU -> Linear)*  with the reference's Sequential indices 0, 2, ... (ref :307-312)."""
    layers = [nn.Linear(hidden_size, output_hidden_size)]
    for _ in range(1, depth):
        layers += [nn.GELU(), nn.Linear(output_hidden_size, output_hidden_size)]
    return nn.Sequential(*layers)


class _TrackedWeights:
    """Mixin of the modules that own weight-derived device caches: their parameters report `.data` accesses
    (native.TrackedParameter: an eval-mode `p.data.copy_(...)` then refreshes the caches by itself).  Re-applied after every
    _apply (.to() / .cuda() may hand out fresh nn.Parameter objects)."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        nv.track_parameters(self)
        return out


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x


class MultiheadAttention(nn.Module):
    """Parameter container for q/k/v/out projections (ref :133-153).  The attention itself runs in
    hicom_global_stream_fwd with k_proj / v_proj folded around the raw tokens."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {embed_dim} and "
                             f"`num_heads`: {num_heads}).")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.dropout = dropout
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.apply(_init_like_reference)

    def forward(self, query, key, value, attention_mask=None, logit_scale=None, logit_bias=None):
        """Reference signature (ref :166-228): [B, q, E] x [B, kv, E] -> ([B, q, E], None).  The attention weights the
        reference returns second are never formed here (every caller drops them, ref :391, :645).
          kv <= 64 keys (the text tokens of "fine" injection): projected states + hicom_small_mha_fwd;
          otherwise key and value must be the same bf16 tensor: k_proj / v_proj are folded around the raw tokens and
          the tokens are streamed once (hicom_global_stream_fwd), as in GlobalCompressor."""
        from . import injector as inj
        if query.ndim != 3 or key.ndim != 3 or value.ndim != 3:
            raise ValueError("MultiheadAttention: inputs are Batch x Time x Channel")
        if attention_mask is not None:
            raise NotImplementedError("MultiheadAttention: attention_mask has no HIP path (never passed by the reference)")
        if not torch.is_grad_enabled():
            nv.begin_inference()
        B, q_len, E = query.shape
        kv_len = key.shape[1]
        if kv_len <= 64:
            # the four projections and the clip normalisation are row-wise: ONE launch each over all batch entries; only the 64-key
            # attention itself is per entry
            dev = query.device
            qp = inj.linear_rows(query.reshape(B * q_len, E).contiguous(), self.q_proj)
            kp = inj.linear_rows(key.reshape(B * kv_len, E).contiguous(), self.k_proj)
            vp = inj.linear_rows(value.reshape(B * kv_len, E).contiguous(), self.v_proj)
            sc = None
            if logit_scale is not None:
                # clip form (ref :184-191): projected queries and keys L2-normalised over the FULL width before the heads are
                # split, logits * exp(logit_scale) (+ logit_bias: a per-row shift, softmax cancels it)
                nv.clip_query_prep(qp, None, 1, 1.0, _f32((B * q_len,), dev))
                nv.clip_query_prep(kp, None, 1, 1.0, _f32((B * kv_len,), dev))
                sc = math.exp(float(logit_scale))
            ao = _f32((B * q_len, self.embed_dim), dev)
            for b in range(B):
                nv.small_mha(qp[b * q_len:(b + 1) * q_len], kp[b * kv_len:(b + 1) * kv_len], vp[b * kv_len:(b + 1) * kv_len], self.num_heads,
                             ao[b * q_len:(b + 1) * q_len], scale=sc)
            out = inj.linear_rows(ao, self.out_proj).view(B, q_len, -1)
            return (out if getattr(self, "return_fp32", False) else out.to(query.dtype)), None
        outs = []
        for b in range(B):                      # long key streams: one pass over each entry's tokens
            q2, k2 = query[b].contiguous(), key[b].contiguous()
            if key[b].data_ptr() != value[b].data_ptr() or key.shape != value.shape:
                raise NotImplementedError("MultiheadAttention: long key streams take key is value (k_proj / v_proj folded)")
            # clip-scale (ref :184-191): queries and PROJECTED keys L2-normalised over the full width, logits * exp(logit_scale)
            # (+ logit_bias, a per-row shift that the softmax cancels): the clip form of the streamed attention
            ml, acc, _ = _stream_attention(self, k2, q2, None, 0, 0, 0, 0, 0,
                                           clip=None if logit_scale is None else float(logit_scale))
            ctx = _f32(acc.shape, acc.device)
            nv.global_combine(ml.unsqueeze(0), acc.unsqueeze(0), ctx)
            wv, bv = _linear_params(self.v_proj)
            o = _f32((q_len, self.embed_dim), q2.device)
            nv.linear(ctx, wv, bv, o, head_rows=self.num_heads, head_dim=self.head_dim)
            outs.append(inj.linear_rows(o, self.out_proj))
        out = torch.stack(outs, 0)
        return (out if getattr(self, "return_fp32", False) else out.to(query.dtype)), None


class GuideInjector(nn.Module):
    """Parameters of the instruction injector (ref :315-342).  'direct' has none."""

    def __init__(self, use_guide, text_dim, qk_dim, adapt_guide=False,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), mlp_depth=2):
        super().__init__()
        self.use_guide = use_guide
        self.text2qk_proj = build_mlp(mlp_depth, text_dim, qk_dim) if text_dim != qk_dim else nn.Identity()
        if adapt_guide:
            self.guide_proj = build_mlp(mlp_depth, qk_dim, qk_dim)
            self.guide_norm = norm_layer(qk_dim)
            self.guide_alpha = nn.Parameter(torch.zeros(1))
        else:
            self.guide_proj, self.guide_norm, self.guide_alpha = nn.Identity(), nn.Identity(), 0
        if use_guide == "coarse":
            self.coarse_proj = build_mlp(mlp_depth, qk_dim, qk_dim * 2)
            self.coarse_norm = norm_layer(qk_dim)
        elif use_guide == "fine":
            self.fine_proj = MultiheadAttention(qk_dim, num_heads=qk_dim // 128)
            self.fine_norm = norm_layer(qk_dim)
        elif use_guide != "direct":
            raise NotImplementedError(f"use_guide={use_guide!r}")

    def forward(self, visual_embed, guide_embed):
        """Reference signature (ref :344-397): visual_embed [t,h,w,d] or [n,d] -> injected queries of that shape
        (fp32: they feed the score kernels).  "direct" returns the (adapted) guide broadcast to the visual shape as a
        stride-0 view -- the reference materialises the repeat (:356, :360)."""
        from . import injector as inj
        if visual_embed.ndim not in (2, 4):
            raise ValueError("Invalid input shape for guide embedding.")
        if not torch.is_grad_enabled():
            nv.begin_inference()
        shape = visual_embed.shape
        vis = visual_embed.reshape(-1, shape[-1])
        q, shared = inj.inject(self, self.use_guide, vis, guide_embed.contiguous())
        return q.reshape(1, -1).expand(vis.shape[0], -1).reshape(shape) if shared else q.reshape(shape)


def _f32(shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


def _plain_injector(inj) -> bool:
    """No text2qk projection in front of the guide (text width == query width, ref :323-326)."""
    return not isinstance(inj, GuideInjector) or isinstance(inj.text2qk_proj, nn.Identity)


def _require_bf16_cuda(name: str, t: torch.Tensor):
    if not t.is_cuda:
        raise nv.HicomNativeError(f"{name}: hicom_amd runs on the GPU only (got a CPU tensor)")
    if t.dtype != torch.bfloat16:
        raise NotImplementedError(f"{name}: dtype {t.dtype}; the HIP path takes bfloat16 tokens and weights "
                                  "(cast the projector and its inputs to torch.bfloat16)")


def _linear_params(lin: nn.Linear):
    _require_bf16_cuda("weight", lin.weight)
    return lin.weight.detach(), (lin.bias.detach() if lin.bias is not None else None)


def _stream_attention(att, x2, q_in, pe, H, W, t0i, y0i, x0i, clip=None, kpe_t=None, kpe=None, need_scores=False, out=None):
    """Streams the tokens x2 [N, E] (bf16) once against the folded queries of q_in [nq, E]: returns the un-normalised
    online-softmax state (ml [R,2], acc [R,E]), R = nq * heads (ref :180-215 restated).

    clip = log logit_scale (a float): the clip-scale variant (ref :184-191).  Queries and keys are L2-normalised over the
    full projected width before the heads are split, so  logit_h(n) = e^ls (qhat_h . k_h(n)) / ||k(n)|| + logit_bias
    (the bias is a per-row shift: softmax cancels it).  k(n) = W_k (x_n + pos_n) + b_k never exists in memory: its norm
    comes from a dense MFMA GEMM with a row-sum-of-squares epilogue (hicom_dense16_gemm_fwd, the positional part as
    three rows of kpe_t = PE . W_k^T per token), the numerator from the usual folded queries.

    need_scores: also return the raw logits [rows_pad, N] (the backward pass reads them); without it the many-row form of the
    kernel keeps the positional marginals itself and the logit tensor is never written (third result None).
    out: callable (R, rows_pad, stride, E) -> (ml, acc, scores) buffers to fill instead of fresh tensors (the training forward's
    store, whose addresses the captured backward reads)."""
    E, nh = att.embed_dim, att.num_heads
    _require_bf16_cuda("key / value tokens", x2)
    N, dev = x2.shape[0], x2.device
    nq = q_in.shape[0]
    R = nq * nh
    rows_pad = (R + 15) // 16 * 16
    wq, bq = _linear_params(att.q_proj)
    wk, _ = _linear_params(att.k_proj)      # b_k only shifts every logit of a row: softmax cancels it
    qp = _f32((nq, E), dev)
    nv.linear(q_in, wq, bq, qp)
    qt = _f32((R, E), dev)
    inv = row_const = None
    if clip is not None:
        scale = math.exp(clip)
        row_const = torch.zeros((rows_pad,), dtype=torch.float32, device=dev)
        nv.clip_query_prep(qp, att.k_proj.bias.detach() if att.k_proj.bias is not None else None, nh, scale, row_const)
        nv.fold_query(qp, wk, nh, scale, qt)                   # qp is now qhat
        ssq = _f32(((E + 63) // 64, N), dev)
        tab = None
        if pe is not None:
            tab = (kpe_t, H, W, t0i, y0i, x0i)
        nv.dense16_gemm(x2, wk, att.k_proj.bias.detach() if att.k_proj.bias is not None else None, ssq=ssq, row_tab=tab)
        inv = _f32((N,), dev)
        nv.inv_norm(ssq, inv)
    qhi = torch.empty((rows_pad, E), dtype=torch.bfloat16, device=dev)
    qlo = torch.empty_like(qhi)
    pos_a = None
    if clip is None and R == rows_pad and (pe is None or kpe is not None):
        # fold + hi/lo split + score-side positional table in ONE launch (kpe = W_k . PE^T is weight-only and cached); every row
        # is written (no padding rows), so nothing needs zeroing
        if pe is not None:
            pos_a = _f32((rows_pad, pe.shape[0]), dev)
        nv.fold_query_split(qp, wk, kpe if pe is not None else None, nh, att.scale, qhi, qlo, pos_a)
    else:
        if clip is None:
            nv.fold_query(qp, wk, nh, att.scale, qt)
        nv.split_bf16(qt, rows_pad, qhi, qlo)
        if pe is not None:
            pos_a = torch.zeros((rows_pad, pe.shape[0]), dtype=torch.float32, device=dev)
            nv.linear(qt, pe, None, pos_a, M=R)                 # a[r, p] = qt[r] . PE[p]
    if pe is None:
        H, W = 1, N                                             # any factorisation of N: no positional terms
    nparts = nv.global_stream_nparts(N, rows_pad)
    stride = (N + 15) // 16 * 16
    part_m, part_l = _f32((nparts, rows_pad), dev), _f32((nparts, rows_pad), dev)
    part_acc = _f32((nparts, rows_pad, E), dev)
    T = N // (H * W)
    scratch = _f32((R * T * (H + W + 2),), dev) if pe is not None else None
    in_kernel = inv is None and pe is not None and nv.global_stream_has_marg(N, E, rows_pad, H, W, nparts)
    if out is not None:
        ml, acc, scores = out(R, rows_pad, stride, E)
    else:
        ml, acc = _f32((R, 2), dev), _f32((R, E), dev)
        scores = _f32((rows_pad, stride), dev) if need_scores or not in_kernel else None
    if in_kernel:
        part_marg = _f32((nparts, rows_pad, nv.global_stream_marg_width(H, W)), dev)
        nv.global_stream_marg(x2, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, scores, part_m, part_l, part_acc, part_marg, rows=R)
        nv.global_merge_marg(part_m, part_l, part_acc, part_marg, R, N, H, W, pe, t0i, y0i, x0i, scratch, ml, acc)
        return ml, acc, scores
    if inv is not None:
        nv.global_stream_clip(x2, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, inv, row_const, scores, part_m, part_l, part_acc, R)
    else:
        nv.global_stream(x2, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, scores, part_m, part_l, part_acc, rows=R)
    nv.global_merge(part_m, part_l, part_acc, R, scores, N, H, W, pe, t0i, y0i, x0i, scratch, ml, acc)
    return ml, acc, scores


class LocalCompressor(_TrackedWeights, nn.Module):
    """Windowed (t x s x s) single-head cross-attention + 2-layer readout (ref :399-559)."""

    def __init__(self, config, temporal_kernel_size=4, spatial_kernel_size=2,
                 adapt_q=False, adapt_k=False, adapt_v=False, adapt_guide=False,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), mlp_depth=2, force_use_guide=False):
        super().__init__()
        qk_dim, _ = geo.tower_dims(config.mm_vision_tower)
        enc, out = config.mm_hidden_size, config.hidden_size
        self.qk_dim = qk_dim
        self.spatial_kernel_size, self.temporal_kernel_size = spatial_kernel_size, temporal_kernel_size
        self.use_guide = getattr(config, "use_guide", None) if force_use_guide is False else force_use_guide
        if self.use_guide in (None, "off"):
            self.guide_injector = IdentityMap()
        else:
            self.guide_injector = GuideInjector(self.use_guide, qk_dim, qk_dim, adapt_guide, norm_layer, mlp_depth)
        if self.use_guide == "direct":
            adapt_q = False                     # ref :428-429
        self.adapt_q, self.adapt_k, self.adapt_v, self.adapt_guide = adapt_q, adapt_k, adapt_v, adapt_guide
        if adapt_q:
            self.q_proj = nn.Linear(qk_dim, qk_dim, bias=False)
            self.q_norm = norm_layer(qk_dim)
            self.q_alpha = nn.Parameter(torch.zeros(1))
        else:
            self.q_proj, self.q_norm, self.q_alpha = nn.Identity(), nn.Identity(), 0
        if adapt_k:
            self.k_proj = build_mlp(mlp_depth, qk_dim, qk_dim)
            self.k_norm = norm_layer(qk_dim)
            self.k_alpha = nn.Parameter(torch.zeros(1))
        else:
            self.k_proj, self.k_norm, self.k_alpha = nn.Identity(), nn.Identity(), 0
        if adapt_v:
            self.v_proj = build_mlp(mlp_depth, enc, enc)
            self.v_norm = norm_layer(enc)
            self.v_alpha = nn.Parameter(torch.zeros(1))
        else:
            self.v_proj, self.v_norm, self.v_alpha = nn.Identity(), nn.Identity(), 0
        self.readout = build_mlp(mlp_depth, enc, out)
        self.apply(_init_like_reference)
        nv.track_parameters(self)

    # -- geometry -------------------------------------------------------------------------
    def tilings(self, T: int, H: int, W: int, modal: str):
        kt = 1 if (modal == "image" or T == 1) else self.temporal_kernel_size      # ref :536
        ks = self.spatial_kernel_size
        return geo.axis_tiling(T, kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)

    @property
    def is_plain(self) -> bool:
        """True for the configurations the one-call executor covers: guide direct / off, no adaptors."""
        return (self.use_guide in _NATIVE_GUIDE_MODES and _plain_injector(self.guide_injector)
                and not (self.adapt_q or self.adapt_k or self.adapt_v or self.adapt_guide))

    @property
    def executor_ok(self) -> bool:
        """The one-call executor also covers the k / v adaptors (the second released recipe `local43_adaptkv_global32`) and -- with the
        per-window queries made in front of the call (`external_queries`) -- coarse / fine injection and the query-side adaptors.
        (Clip-scale on the local stage with adaptors or injected queries: operator by operator.)"""
        return self.use_guide in (None, "off", "direct", "coarse", "fine") and self.qk_dim % 64 == 0

    @property
    def queries_native(self) -> bool:
        """The per-window queries are something hicom_compressor_fwd has without help: the pooled queries of guide off, the guide row
        of plain direct injection."""
        return self.use_guide in _NATIVE_GUIDE_MODES and _plain_injector(self.guide_injector) and not (self.adapt_q or self.adapt_guide)

    @property
    def inject_in_call(self) -> bool:
        """coarse / fine injection with a plain injector into the plain pooled queries (ref :369-397, :542): hicom_compressor_fwd runs
        the injector itself (hicom_compressor_args.inj_l)."""
        return (self.use_guide in ("coarse", "fine") and _plain_injector(self.guide_injector) and not (self.adapt_q or self.adapt_guide))

    @property
    def external_queries(self) -> bool:
        """Everything else -- adapt_q (:541), an adapted or re-projected guide (:364-365), alone or in front of an injector: the engine
        runs pooling + adaptor + injector in front of the call and hands the rows in (f32 [windows, E])."""
        return not (self.queries_native or self.inject_in_call)

    def make_queries(self, ff, guide_embed, grid, pooled, out):
        """The per-window queries of ref :539-542 into `out` (f32 [windows | 1, E]); `pooled` f32 [*grid, E] is scratch."""
        from . import injector as inj
        E = ff.shape[-1]
        mode = self.use_guide if self.use_guide not in (None, "off") else None
        if mode == "direct":
            q, _ = inj.inject(self.guide_injector, "direct", None, guide_embed.contiguous())
            q = q.reshape(1, E)
        else:
            nv.trilinear_pool(ff, pooled)                                  # ref :539-540
            q = pooled.view(-1, E)
            if self.adapt_q:
                q = inj.adapt_query(q, self.q_proj, self.q_norm, self.q_alpha)                      # ref :541
            if mode in ("coarse", "fine"):
                q, _ = inj.inject(self.guide_injector, mode, q, guide_embed.contiguous(), out=out)  # ref :369-397
        if q.data_ptr() != out.data_ptr():
            nv.scatter_rows(q.reshape(out.shape).contiguous(), out, 0, out.shape[0])
        return out

    def _check_native(self):
        if self.use_guide not in (None, "off", "direct", "coarse", "fine"):
            raise NotImplementedError(f"LocalCompressor: use_guide={self.use_guide!r}")

    # -- attention context: [Nw, D] fp32 -----------------------------------------------------
    def window_context(self, frames_feature, frames_embed, guide_embed, modal, logit_scale, logit_bias, adapt_y=None):
        """Operator-by-operator form of ref :524-558 (everything before the readout), all variants.
        adapt_y = (y_k, y_v): the adaptor MLPs' outputs over all tokens (fp16 [N, D] or None each) when the caller already has them
        (the backward pass computes them with their intermediates): the four dense GEMMs are then not run again."""
        from . import injector as inj
        self._check_native()
        _require_bf16_cuda("frames_feature", frames_feature)
        ff = frames_feature.contiguous()
        T, H, W, D = ff.shape
        key = ff if frames_embed is None else frames_embed.contiguous()
        if key is not ff:
            _require_bf16_cuda("frames_embed", key)
            if key.shape != ff.shape:
                # e.g. the real CLIP-L tower: 768-d projected keys against 1024-d hidden-state values
                raise NotImplementedError("LocalCompressor: key width != value width (frames_embed "
                                          f"{tuple(key.shape)} vs frames_feature {tuple(ff.shape)}) has no HIP path")
        at, ay, ax = self.tilings(T, H, W, modal)
        axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
        grid = (at.nwin, ay.nwin, ax.nwin)
        nw = grid[0] * grid[1] * grid[2]
        l2norm = 0
        guide_n = None                                                # clip-scale with an injector: the guide rows L2-normalised FIRST
        if logit_scale is not None:                                   # ref :527-529, :549
            scale, bias = math.exp(float(logit_scale)), float(logit_bias)
            if frames_embed is not None:
                if self.adapt_k:
                    # the keys are normalised BEFORE the adaptor MLP (ref :527-529 in front of :533) and not again behind it
                    kn = torch.empty_like(key)
                    nv.l2norm_stream(key.view(-1, D), kn.view(-1, D))
                    key = kn
                else:
                    l2norm = 1                                        # every key row, inside the window kernel
                if self.use_guide == "direct" and not self.adapt_guide:
                    l2norm |= 2                                       # ... and the shared query (= the guide) as well
                elif self.use_guide in ("direct", "coarse", "fine"):
                    # the reference normalises guide_embed and THEN injects / adapts it (:529 before :542): the query that comes out
                    # of the injector is not normalised again
                    _require_bf16_cuda("guide_embed", guide_embed)
                    g2 = guide_embed.contiguous().reshape(-1, D)
                    guide_n = _f32(g2.shape, ff.device)
                    nv.scatter_rows(g2, guide_n, 0, g2.shape[0])      # bf16 -> f32 rows
                    nv.clip_query_prep(guide_n, None, 1, 1.0, _f32((g2.shape[0],), ff.device))   # rows / ||row||
                    guide_n = guide_n.reshape(guide_embed.shape)
        else:
            scale, bias = 1.0 / math.sqrt(self.qk_dim), 0.0            # ref :551
        value = ff
        # adaptors: the two dense GEMMs per stream, then -- unless a kernel-side normalisation is in play -- the LayerNorm + alpha
        # blend fused into the window kernel's row loads (the blended streams are never written)
        fuse_blend = (self.adapt_k or self.adapt_v) and l2norm == 0
        ky = vy = None
        have_y = adapt_y is not None and fuse_blend
        if self.adapt_k:                                               # ref :533
            if have_y:
                ky = adapt_y[0]
            elif fuse_blend:
                ky = inj.adapt_stream_y(key, self.k_proj)
            else:
                key = inj.adapt_stream(key, self.k_proj, self.k_norm, self.k_alpha)
        if self.adapt_v:                                               # ref :534
            if have_y:
                vy = adapt_y[1]
            elif fuse_blend:
                vy = inj.adapt_stream_y(ff, self.v_proj)
            else:
                value = inj.adapt_stream(ff, self.v_proj, self.v_norm, self.v_alpha)
        ctx = _f32((nw, D), ff.device)

        def attend(q, q_stride):
            if fuse_blend:
                nv.local_attn_adapt(key, ky, self.k_norm if ky is not None else None, self.k_alpha.detach() if ky is not None else None,
                                    ff, vy, self.v_norm if vy is not None else None, self.v_alpha.detach() if vy is not None else None,
                                    axes, q, q_stride, scale, bias, ctx,
                                    eps=(self.k_norm if ky is not None else self.v_norm).eps)
            else:
                nv.local_attn(key, value, axes, q, q_stride, scale, bias, l2norm, ctx)
        if self.use_guide == "direct":                                 # query := guide for every window (:352-368)
            _require_bf16_cuda("guide_embed", guide_embed)
            q, _ = inj.inject(self.guide_injector, "direct", None, guide_n if guide_n is not None else guide_embed.contiguous())
            attend(q.reshape(-1).contiguous(), 0)
            return ctx, grid
        q = _f32((*grid, D), ff.device)                                # pooled per-window query (ref :539-540)
        nv.trilinear_pool(ff, q)
        if self.adapt_q:                                               # ref :541
            q = inj.adapt_query(q, self.q_proj, self.q_norm, self.q_alpha)
        if self.use_guide in ("coarse", "fine"):
            _require_bf16_cuda("guide_embed", guide_embed)
            q, _ = inj.inject(self.guide_injector, self.use_guide, q.reshape(nw, D),
                              guide_n if guide_n is not None else guide_embed.contiguous())
        attend(q.reshape(nw, D), D)
        return ctx, grid

    def readout_f16(self):
        """fp16 copies of the two readout weights for hicom_readout16_gemm_fwd (nv.f16_weight_copy: exact above 2^-14, range
        checked), rebuilt when a weight is replaced or modified in place, on every training-mode forward and after
        hicom_amd.invalidate_weight_caches() (nv.weight_stamp) -- a weight-only cache like the reference's pos_embed buffer."""
        w0, w2 = self.readout[0].weight, self.readout[2].weight
        _require_bf16_cuda("readout weight", w0)
        stamp = nv.weight_stamp(w0, w2)
        hit = self.__dict__.get("_f16_cache")
        if hit is None or hit[0] != stamp:
            old = hit or (None, None, None)                  # refreshed in place where the shapes allow: plans keep their pointers
            hit = (stamp, nv.f16_weight_copy(w0, out=old[1]), nv.f16_weight_copy(w2, out=old[2]))
            self.__dict__["_f16_cache"] = hit
        return hit[1], hit[2]

    def readout_into(self, ctx, out, row0: int, nl_group: int):
        """out[row0 + packed(m), :] = readout(ctx[m, :]) -- both Linear layers on matrix cores."""
        w0, b0 = _linear_params(self.readout[0])
        w2, b2 = _linear_params(self.readout[2])
        if ctx.shape[1] % 64 == 0 and w0.shape[0] % 64 == 0:
            # the hot path's GEMM (one fp16 plane per activation, cached fp16 weights): 12 us per layer at 1296 rows against
            # 25 for the fp32-input form, same 2^-12 activation rounding as the one-call executor
            w0h, w2h = self.readout_f16()
            hid16 = torch.empty((ctx.shape[0], w0.shape[0]), dtype=torch.float16, device=ctx.device)
            nv.readout16_gemm(nv.to_f16(ctx), w0h, b0, act=nv.ACT_GELU, out_f16=hid16)
            nv.readout16_gemm(hid16, w2h, b2, y=out, row0=row0, nl_group=nl_group)
            return
        hid = _f32((ctx.shape[0], w0.shape[0]), ctx.device)
        nv.readout_gemm(ctx, w0, b0, hid, act=nv.ACT_GELU)
        nv.readout_gemm(hid, w2, b2, out, row0=row0, nl_group=nl_group)

    def forward(self, frames_feature, frames_embed, guide_embed, modal, logit_scale=None, logit_bias=None):
        _refuse_grad(self)
        ctx, grid = self.window_context(frames_feature, frames_embed, guide_embed, modal, logit_scale, logit_bias)
        out = torch.empty((ctx.shape[0], self.readout[2].out_features), dtype=_out_dtype(self), device=ctx.device)
        self.readout_into(ctx, out, 0, 0)
        return out.view(*grid, -1)


def _refuse_grad(module: nn.Module):
    """The stage modules have no autograd graph of their own (HIComProjector.forward has: hicom_amd/autograd.py).
    Returning a detached tensor to a training loop would silently freeze the projector, so refuse instead."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise RuntimeError(f"{type(module).__name__}.forward builds no autograd graph: call it under torch.no_grad(), or "
                           "go through HIComProjector.forward, which does")
    nv.begin_inference()


def _out_dtype(module: nn.Module) -> torch.dtype:
    return torch.float32 if getattr(module, "return_fp32", False) else torch.bfloat16


class GlobalCompressor(_TrackedWeights, nn.Module):
    """num_queries x 9-head cross-attention over all T*h*w tokens + readout (ref :562-646)."""

    def __init__(self, config, num_queries, use_pos_emb=True, adapt_guide=False,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), mlp_depth=2, force_use_guide=False):
        super().__init__()
        text_dim, hw = geo.tower_dims(config.mm_vision_tower)
        self.embed_dim = embed_dim = config.mm_hidden_size
        self.num_queries = num_queries
        self.use_pos_emb = use_pos_emb
        self.max_size = [getattr(config, "max_num_frames", 256), hw, hw]
        self.query = nn.Parameter(torch.zeros(num_queries, embed_dim))
        self.use_guide = getattr(config, "use_guide", None) if force_use_guide is False else force_use_guide
        self.adapt_guide = adapt_guide
        if self.use_guide in (None, "off"):
            self.guide_injector = IdentityMap()
        else:
            self.guide_injector = GuideInjector(self.use_guide, text_dim, embed_dim, adapt_guide, norm_layer, mlp_depth)
        self.attn_layer = MultiheadAttention(embed_dim, embed_dim // 128)
        self.readout = build_mlp(mlp_depth, embed_dim, config.hidden_size)
        self.apply(_init_like_reference)
        nv.track_parameters(self)
        self._pe_cache: Dict[Tuple, torch.Tensor] = {}
        self._cache_gen = 0          # bumped whenever a cached device table is (re)built: invalidates engine plans

    # per-axis sinusoid tables [t_cap + H + W, E] fp32 on the device (ref :57-101, :603-621: the
    # reference caches the full [T,27,27,E] sum; we keep the three separable factors)
    def pos_tables(self, t_cap: int, H: int, W: int, device) -> Tuple[torch.Tensor, int]:
        t_cap = max(t_cap, 1)
        key = (H, W, str(device))
        hit = self._pe_cache.get(key)
        if hit is None or hit[1] < t_cap:
            cap = max(t_cap, min(self.max_size[0], 4096))
            tab = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, self.embed_dim)).to(device)
            hit = (tab, cap)
            self._pe_cache[key] = hit
            self._cache_gen += 1
        return hit

    def pos_and_kpe(self, t_cap: int, H: int, W: int, device):
        """(pe [P,E], kpe [E,P] = k_proj.weight . pe^T, cap).  kpe depends on the weights only and is
        rebuilt when k_proj.weight is replaced or modified in place (tensor version counter)."""
        pe, cap = self.pos_tables(t_cap, H, W, device)
        wk = self.attn_layer.k_proj.weight
        _require_bf16_cuda("k_proj.weight", wk)
        key = ("kpe", H, W, cap, str(device))
        stamp = nv.weight_stamp(wk)
        hit = self._pe_cache.get(key)
        if hit is None or hit[1] != stamp:
            reuse = hit is not None and tuple(hit[0].shape) == (self.embed_dim, pe.shape[0])
            kpe = hit[0] if reuse else _f32((self.embed_dim, pe.shape[0]), device)      # in place: plans keep their pointers
            nv.linear(wk.detach(), pe, None, kpe)
            hit = (kpe, stamp)
            self._pe_cache[key] = hit
            if not reuse:
                self._cache_gen += 1
        return pe, hit[0], cap

    def vpe_f16(self, T: int, H: int, W: int, device, t_offset: int = 0):
        """fp16 [E, S] = v_proj.weight . pe_sel^T with pe_sel the pe rows of the slots [T frames (from t_offset) | H grid rows | W grid
        columns], zero-padded to S = 8 * (E / 64) slots -- or None when they do not fit.  By linearity of v_proj the value-side pos-emb
        of the global stage (reference projector.py:636-640 with :182, :215) is sum_s marginal[s] VPE[:, s] on top of W_v ctx: the release
        step applies it in the merge role instead of behind the token stream (csrc/merge_item.hpp).  Weight-only: cached per weight
        state like kpe and rebuilt IN PLACE (plans keep its address)."""
        E = self.embed_dim
        S = 8 * (E // 64)
        if E % 64 or T + H + W > S:
            return None
        pe, cap = self.pos_tables(t_offset + T, H, W, device)
        wv = self.attn_layer.v_proj.weight
        _require_bf16_cuda("v_proj.weight", wv)
        key = ("vpe16", T, t_offset, H, W, cap, str(device))
        stamp = nv.weight_stamp(wv) + (pe.data_ptr(),)
        hit = self._pe_cache.get(key)
        if hit is None or hit[1] != stamp:
            if hit is None:
                if sum(1 for k in self._pe_cache if isinstance(k, tuple) and k[0] == "vpe16") >= 4:      # (a few frame counts per module)
                    self._pe_cache.pop(next(k for k in self._pe_cache if isinstance(k, tuple) and k[0] == "vpe16"))
                    self._cache_gen += 1
                sel = torch.zeros((S, E), dtype=torch.float32, device=device)
                sel[:T] = pe[t_offset:t_offset + T]
                sel[T:T + H + W] = pe[cap:cap + H + W]
                hit = (torch.empty((E, S), dtype=torch.float16, device=device), None, sel, _f32((E, S), device))
                self._cache_gen += 1
            nv.linear(wv.detach(), hit[2], None, hit[3])                  # [E, S] = W_v . pe_sel^T (HIP; f32)
            nv.to_f16(hit[3], hit[0])
            hit = (hit[0], stamp, hit[2], hit[3])
            self._pe_cache[key] = hit
        return hit[0]

    def readout_over_out_proj(self):
        """C [hidden, E] f32 = readout[0].weight . attn_layer.out_proj.weight: the first global readout layer folded over
        out_proj, so that GELU(G0 (W_o o + b_o + q) + b0) (ref :226, :646, :307-312) is GELU(C o + r0) with the guide-dependent
        r0 = G0 (b_o + q) + b0 made by hicom_query_prep_fwd -- one dependent stage fewer behind the streaming kernel.
        Weight-only, cached per weight state like kpe."""
        g0, wo = self.readout[0].weight, self.attn_layer.out_proj.weight
        _require_bf16_cuda("readout weight", g0)
        stamp = nv.weight_stamp(g0, wo)
        hit = self._pe_cache.get("gc0")
        if hit is None or hit[1] != stamp:
            reuse = hit is not None and tuple(hit[0].shape) == (g0.shape[0], wo.shape[1]) and hit[0].device == g0.device
            c = hit[0] if reuse else _f32((g0.shape[0], wo.shape[1]), g0.device)          # in place: plans keep their pointers
            wot = wo.detach().t().contiguous()
            if g0.shape[1] % 64 == 0:
                nv.dense16_gemm(g0.detach(), wot, None, y=c)                   # C[n, e] = sum_k G0[n, k] W_o[k, e]: bf16 x bf16
            else:                                                              # products are exact in fp32, fp32 accumulation
                nv.linear(g0.detach(), wot, None, c)
            hit = (c, stamp)
            self._pe_cache["gc0"] = hit
            if not reuse:
                self._cache_gen += 1
        return hit[0]

    def pos_planes(self, t_cap: int, H: int, W: int, device):
        """bf16 hi / lo planes of the pe table (the fused kernel multiplies them on matrix cores), cached with it."""
        pe, cap = self.pos_tables(t_cap, H, W, device)
        key = ("planes", H, W, cap, str(device))
        hit = self._pe_cache.get(key)
        if hit is None or hit[2] != pe.data_ptr():
            hi = torch.empty(pe.shape, dtype=torch.bfloat16, device=device)
            lo = torch.empty_like(hi)
            nv.split_bf16(pe, pe.shape[0], hi, lo)
            hit = (hi, lo, pe.data_ptr())
            self._pe_cache[key] = hit
            self._cache_gen += 1
        return hit[0], hit[1]

    @property
    def is_plain(self) -> bool:
        return self.use_guide in _NATIVE_GUIDE_MODES and _plain_injector(self.guide_injector) and not self.adapt_guide

    @property
    def executor_ok(self) -> bool:
        return self.use_guide in (None, "off", "direct", "coarse", "fine")

    queries_native = is_plain

    @property
    def inject_in_call(self) -> bool:
        """coarse / fine injection of the guide into the learnable queries through a plain injector (ref :642 with :369-397):
        hicom_compressor_fwd runs the injector itself (hicom_compressor_args.inj_g)."""
        return self.use_guide in ("coarse", "fine") and _plain_injector(self.guide_injector) and not self.adapt_guide

    @property
    def external_queries(self) -> bool:
        """An adapted or re-projected guide in front of the injection: made in front of hicom_compressor_fwd, handed in as f32 rows."""
        return not (self.is_plain or self.inject_in_call)

    def make_queries(self, guide_embed, out):
        """The injected global queries (ref :642) into `out` (f32 [rows, E])."""
        q, _ = self.injected_queries(guide_embed, out=out)
        if q.data_ptr() != out.data_ptr():
            nv.scatter_rows(q.reshape(out.shape).contiguous(), out, 0, out.shape[0])
        return out

    def _check_native(self, logit_scale):
        if self.use_guide not in (None, "off", "direct", "coarse", "fine"):
            raise NotImplementedError(f"GlobalCompressor: use_guide={self.use_guide!r}")
        # (clip-scale, logit_scale given: the operator-by-operator path -- _stream_attention(clip=...) -- has it; the
        # one-call executor does not, HIComProjector.forward routes such projectors to forward_stepwise)

    def injected_queries(self, guide_embed, out=None) -> Tuple[torch.Tensor, int]:
        """[nq_eff, E] distinct injected query rows (bf16 or f32) and how many output rows they stand for.
        direct: the 32 queries are 32 copies of the (adapted) guide (ref :352-368, :642) -> one row."""
        from . import injector as inj
        if guide_embed is not None:
            _require_bf16_cuda("guide_embed", guide_embed)
            guide_embed = guide_embed.contiguous()
        if self.use_guide in (None, "off"):
            _require_bf16_cuda("global_compressor.query", self.query)
            return self.query.detach(), self.num_queries
        if self.use_guide == "direct":
            if guide_embed.ndim != 1 or guide_embed.shape[0] != self.embed_dim:
                raise ValueError("direct guide injection takes a [D] guide embedding")
        _require_bf16_cuda("global_compressor.query", self.query)
        q, shared = inj.inject(self.guide_injector, self.use_guide, self.query.detach(), guide_embed, out=out)
        return q.reshape(-1, self.embed_dim).contiguous(), self.num_queries

    def pos_kpe_t(self, t_cap: int, H: int, W: int, device):
        """kpe_t [P, E] = PE . W_k^T: the projected positional embedding per table row (clip-scale key norms); weight-only,
        cached like kpe."""
        pe, cap = self.pos_tables(t_cap, H, W, device)
        wk = self.attn_layer.k_proj.weight
        key = ("kpe_t", H, W, cap, str(device))
        stamp = nv.weight_stamp(wk) + (pe.data_ptr(),)
        hit = self._pe_cache.get(key)
        if hit is None or hit[1] != stamp:
            t = _f32((pe.shape[0], self.embed_dim), device)
            nv.linear(pe, wk.detach(), None, t)
            hit = (t, stamp)
            self._pe_cache[key] = hit
        return hit[0]

    def partial_context(self, frames_feature, q_in, t_offset: int = 0, logit_scale=None, need_scores: bool = False):
        """Streams this call's frames once: returns (ml [R,2], acc [R,E], raw logits [rows_pad, N'] or None) -- the un-normalised
        online-softmax state for R = nq_eff * num_heads folded query rows (ref :180-215 restated).  The logits are
        only guaranteed with need_scores (the backward pass): the many-row kernel does not write them otherwise."""
        ff = frames_feature.contiguous()
        _require_bf16_cuda("frames_feature", ff)
        T, H, W, E = ff.shape
        pe = None
        t0i = y0i = x0i = 0
        kpe = None
        if self.use_pos_emb:
            pe, kpe, cap = self.pos_and_kpe(t_offset + T, H, W, ff.device)
            t0i, y0i, x0i = t_offset, cap, cap + H
        clip, kpe_t = None, None
        if logit_scale is not None:
            clip = float(logit_scale)
            kpe_t = self.pos_kpe_t(t_offset + T, H, W, ff.device) if self.use_pos_emb else None
        # training forward (autograd._CompressorFn): softmax state and logits go into the per-shape store the backward reads
        store = getattr(self, "_train_store", None) if (t_offset == 0 and logit_scale is None) else None
        if store is not None:
            res = _stream_attention(self.attn_layer, ff.view(T * H * W, E), q_in, pe, H, W, t0i, y0i, x0i, clip, kpe_t, kpe, True,
                                    out=lambda R, rows_pad, stride, E_: store.buffers(R, rows_pad, stride, E_, ff.device))
            store.serial += 1
            return res
        return _stream_attention(self.attn_layer, ff.view(T * H * W, E), q_in, pe, H, W, t0i, y0i, x0i, clip, kpe_t, kpe, need_scores)

    def finish(self, ml_sets, acc_sets, q_in, out, row0: int, n_rows: int):
        """Combine shard states, apply v_proj per head, out_proj + residual, readout, and write
        n_rows output rows (broadcast when the queries are identical) at out[row0:]."""
        att = self.attn_layer
        E, nh = self.embed_dim, att.num_heads
        dev = out.device
        nq = q_in.shape[0]
        R = nq * nh
        ctx = _f32((R, E), dev)
        nv.global_combine(ml_sets, acc_sets, ctx)
        wv, bv = _linear_params(att.v_proj)
        wo, bo = _linear_params(att.out_proj)
        o = _f32((nq, E), dev)
        nv.linear(ctx, wv, bv, o, head_rows=nh, head_dim=att.head_dim)   # sum(p) = 1 carries b_v through
        qres = _f32((nq, E), dev)
        nv.scatter_rows(q_in, qres, 0, nq)                                # bf16 -> f32 residual (ref :646)
        pre = _f32((nq, E), dev)
        nv.linear(o, wo, bo, pre, res=qres)
        w0, b0 = _linear_params(self.readout[0])
        w2, b2 = _linear_params(self.readout[2])
        hid = _f32((nq, w0.shape[0]), dev)
        nv.linear(pre, w0, b0, hid, act=nv.ACT_GELU)
        tok = _f32((nq, w2.shape[0]), dev)
        nv.linear(hid, w2, b2, tok)
        nv.scatter_rows(tok, out, row0, n_rows)

    def forward_into(self, frames_feature, guide_embed, logit_scale, out, row0: int, logit_bias=None):
        self._check_native(logit_scale)
        q_in, n_rows = self.injected_queries(guide_embed)
        ml, acc, _ = self.partial_context(frames_feature, q_in, logit_scale=logit_scale)
        self.finish(ml.unsqueeze(0), acc.unsqueeze(0), q_in, out, row0, n_rows)

    def forward(self, frames_feature, frames_embed, guide_embed, modal, logit_scale=None, logit_bias=None):
        _refuse_grad(self)
        out = torch.empty((self.num_queries, self.readout[2].out_features), dtype=_out_dtype(self),
                          device=frames_feature.device)
        self.forward_into(frames_feature, guide_embed, logit_scale, out, 0, logit_bias)
        return out


class HIComProjector(nn.Module):
    """[local tokens (+ newline rows) ; global tokens] for one video / image (ref :649-708)."""

    def __init__(self, config, local_compressor=None, global_compressor=None):
        super().__init__()
        self.config = config
        use_clip_scale = (getattr(config, "use_clip_scale", "") or "").split(",")
        self.local_use_clip_scale = "local" in use_clip_scale
        self.global_use_clip_scale = "global" in use_clip_scale
        # The reference copies SigLIP's logit_scale / logit_bias out of the hub checkpoint at construction (ref :660-670) and keeps them as
        # PARAMETERS of the projector (`local_logit_scale`, ... [1]: part of its state dict, trainable under `attn_scale`,
        # train.py:730-734).  This build has no hub access: the parameters are registered here with NaN ("not given yet") and filled by
        # load_state_dict() -- a checkpoint saved by the reference carries them -- or by set_clip_logits(); a projector whose logits are
        # still NaN refuses to run.  The Python floats the plans bake in (`local_logit` / `global_logit`) follow the parameters'
        # content (_sync_clip_logits: one host read per change, none per call).
        self.local_logit_scale = self.local_logit_bias = None
        self.global_logit_scale = self.global_logit_bias = None
        for stage, on in (("local", self.local_use_clip_scale), ("global", self.global_use_clip_scale)):
            if on:
                self._register_clip_params(stage)
        self.local_logit = self.global_logit = None          # (log scale, bias) as Python floats: what the plans bake in
        self.local_compressor = local_compressor
        self.global_compressor = global_compressor
        assert local_compressor is not None or global_compressor is not None, \
            "At least one compressor should be provided."
        self.return_fp32 = False     # True: fp32 result (parity tests); default = weight dtype (bf16)
        self.use_executor = True     # dense inputs go through the one-call native executor (engine.py)
        self.graph_replay = False    # True: capture each cached plan into a hipGraph and replay it

    def _invalidate_plans(self):
        self.__dict__["_engine_params_gen"] = self.__dict__.get("_engine_params_gen", 0) + 1
        self.__dict__.pop("_engine_plans", None)
        self.__dict__.pop("_shard_plans", None)

    def _apply(self, fn, *args, **kwargs):           # .to() / .cuda() / .bfloat16() ...
        self._invalidate_plans()
        out = super()._apply(fn, *args, **kwargs)
        nv.track_parameters(self)
        return out

    def load_state_dict(self, *args, **kwargs):
        self._invalidate_plans()
        return super().load_state_dict(*args, **kwargs)

    def _register_clip_params(self, stage: str):
        ref = next(self.parameters(), None)
        kw = {} if ref is None else dict(device=ref.device, dtype=ref.dtype)
        for what in ("scale", "bias"):
            name = f"{stage}_logit_{what}"
            if not isinstance(getattr(self, name, None), nn.Parameter):
                if name in self.__dict__:
                    del self.__dict__[name]                      # (the plain `None` attribute set before registration)
                self.register_parameter(name, nn.Parameter(torch.full((1,), float("nan"), **kw), requires_grad=False))

    def set_clip_logits(self, local=None, glob=None):
        """(logit_scale, logit_bias) of the SigLIP checkpoint for the stages named in config.use_clip_scale (ref :660-670 reads them
        from AutoModel.from_pretrained), written into the projector's `*_logit_scale` / `*_logit_bias` parameters.  Tensors or numbers."""
        for stage, val in (("local", local), ("global", glob)):
            if val is None:
                continue
            self._register_clip_params(stage)
            with torch.no_grad():
                getattr(self, f"{stage}_logit_scale").fill_(float(val[0]))
                getattr(self, f"{stage}_logit_bias").fill_(float(val[1]))
        self._sync_clip_logits(force=True)

    def _clip_params(self):
        return [p for p in (self.local_logit_scale, self.local_logit_bias, self.global_logit_scale, self.global_logit_bias)
                if isinstance(p, nn.Parameter)]

    def _sync_clip_logits(self, force: bool = False):
        """`local_logit` / `global_logit` (Python floats) from the parameters, re-read only when their content may have changed
        (storage, in-place version, weights epoch: load_state_dict, .to(), an optimizer step, `p.data.copy_`)."""
        ps = self._clip_params()
        if not ps:
            return
        stamp = nv.weight_stamp(*ps)
        if not force and stamp == self.__dict__.get("_clip_stamp"):
            return

        def read(scale, bias):
            if not isinstance(scale, nn.Parameter) or not isinstance(bias, nn.Parameter):
                return None
            v = (float(scale.detach().float().cpu()), float(bias.detach().float().cpu()))
            return None if any(math.isnan(x) for x in v) else v

        new = (read(self.local_logit_scale, self.local_logit_bias), read(self.global_logit_scale, self.global_logit_bias))
        if new != (self.local_logit, self.global_logit):
            self.local_logit, self.global_logit = new
            self._invalidate_plans()
        self.__dict__["_clip_stamp"] = nv.weight_stamp(*ps)

    def _logit_args(self, stage: str):
        """(log scale, bias) as the stage modules' forward takes them (reference :524, :634), from the cached floats."""
        v = self.local_logit if stage == "local" else self.global_logit
        return (None, None) if v is None else v

    def _check_clip_logits(self):
        self._sync_clip_logits()
        if (self.local_use_clip_scale and self.local_logit is None) or (self.global_use_clip_scale and self.global_logit is None):
            raise RuntimeError("config.use_clip_scale names a stage whose SigLIP logit_scale / logit_bias have not been "
                               "given: load a checkpoint that carries `*_logit_scale` / `*_logit_bias` or call "
                               "set_clip_logits(local=(scale, bias), glob=(scale, bias)) first")

    def _layout(self, grid, modal, has_newline, is_anyres):
        return geo.pack_layout(getattr(self.config, "mm_patch_merge_type", "flat"),
                               getattr(self.config, "mm_newline_position", "one_token"),
                               modal, grid[0], grid[1], grid[2], has_newline, is_anyres)

    def _needs_grad(self, *tensors) -> bool:
        """True when autograd would expect a graph from this call (training stages 1-2 keep the projector trainable,
        ref train.py:704-712).  Inference runs under torch.inference_mode() / no_grad (ref __init__.py:107): one
        flag test."""
        if not torch.is_grad_enabled():
            return False
        from . import engine
        if any(p.requires_grad for p in engine._param_list(self)[1]):
            return True
        for t in tensors:
            for u in (t.values() if isinstance(t, dict) else (t,)):
                if isinstance(u, torch.Tensor) and u.requires_grad:
                    return True
        return False

    def forward(self, frames_feature, frames_embed, guide_embed, modal, image_newline=None, *, local_logits=None):
        """Reference signature (projector.py:676).  Extension for the producer of frames_embed:
        `local_logits` = fp32 [T,H,W] raw dot products frames_embed_n . guide from `hicom_amd.siglip_head_scores`, passed with
        frames_embed=None -- the release recipe then streams frames_feature only (half the bytes)."""
        if (self.local_compressor or self.global_compressor).readout[0].weight.dtype == torch.float16:      # (one attribute chain on the hot path)
            return self._forward_half(frames_feature, frames_embed, guide_embed, modal, image_newline, local_logits)
        self._check_clip_logits()
        if local_logits is not None:
            return self._forward_with_logits(frames_feature, frames_embed, guide_embed, modal, image_newline, local_logits)
        if self._needs_grad(frames_feature, frames_embed, guide_embed, image_newline):
            # training: an optimizer step lies between two forwards, and DeepSpeed's bf16 optimizer writes the weights
            # through `p.data.copy_` / a flat alias, which no version counter sees -> every weight-derived cache is
            # rebuilt from the live weights on each training forward (native.py "weight-derived device caches")
            nv.note_training_forward()
            from . import autograd
            return autograd.forward_with_grad(self, frames_feature, frames_embed, guide_embed, modal, image_newline)
        nv.begin_inference()           # (the first inference forward after training rebuilds the weight-derived tables)
        if self.use_executor and self._executor_covers():
            from . import engine
            if isinstance(frames_feature, dict):             # anyres image (ref :679-700): one call per segment
                return engine.run_anyres(self, frames_feature, frames_embed, guide_embed, modal, image_newline, _out_dtype(self))
            return engine.run_dense(self, frames_feature, frames_embed, guide_embed, modal, image_newline,
                                    _out_dtype(self))
        return self.forward_stepwise(frames_feature, frames_embed, guide_embed, modal, image_newline)

    def _forward_half(self, frames_feature, frames_embed, guide_embed, modal, image_newline, local_logits):
        """An fp16 projector with fp16 inputs -- the reference's inference default (`--dtype float16`,
        inference_video_mcqa_videomme.py:323; `load_mm_projector` casts the loaded weights to fp16, projector.py:53).  The kernels
        compute on bf16 tokens and weights, so the call runs on a bf16 TWIN of this module: its weights are cast once per weight state
        (a checkpoint trained in bf16 and loaded as fp16 converts back exactly; a genuinely fp16-trained weight is rounded to 8
        significand bits), the inputs are cast per call by hicom_cast16_fwd (fp16 activations are ROUNDED to bf16: 2^-9 relative -- the
        price of this width, stated here rather than hidden), the result is cast back to fp16.  Inference only."""
        import copy
        from . import engine
        if local_logits is not None:
            raise NotImplementedError("local_logits= takes a bfloat16 projector")
        if self._needs_grad(frames_feature, frames_embed, guide_embed, image_newline):
            raise NotImplementedError("hicom_amd: training takes a bfloat16 projector (the fp16 width is an inference path)")
        sig = engine.content_sig(self) + (engine.plan_sig(self)[1:3],)
        twin = self.__dict__.get("_bf16_twin")
        if twin is None or twin[1] != sig:
            with torch.no_grad():
                if twin is None:
                    saved = {k: self.__dict__.pop(k) for k in ("_bf16_twin", "_engine_plans", "_shard_plans", "_bwd_graphs", "_last_plan")
                             if k in self.__dict__}
                    try:
                        mod = copy.deepcopy(self).to(torch.bfloat16)
                    finally:
                        self.__dict__.update(saved)
                else:
                    mod = twin[0]
                    for pt, ps in zip(mod.parameters(), self.parameters()):
                        pt.copy_(ps)                                   # in place: the twin's plans keep their addresses
                mod.train(self.training)
                mod.return_fp32 = getattr(self, "return_fp32", False)
            twin = (mod, sig)
            self.__dict__["_bf16_twin"] = twin

        def down(t):
            if t is None:
                return None
            if isinstance(t, dict):
                return {k: down(v) for k, v in t.items()}
            if t.dtype == torch.bfloat16:
                return t
            if t.dtype != torch.float16 or not t.is_cuda:
                raise NotImplementedError(f"hicom_amd: an fp16 projector takes fp16 (or bf16) CUDA tensors (got {t.dtype} on {t.device})")
            return nv.cast16(t, torch.bfloat16)
        with torch.no_grad():
            out = twin[0](down(frames_feature), down(frames_embed), down(guide_embed), modal, down(image_newline))
        return out if out.dtype == torch.float32 else nv.cast16(out, torch.float16)

    def _executor_covers(self) -> bool:
        """Recipes whose token-stream work hicom_compressor_fwd runs in one call: every injection mode (coarse / fine / the query-side
        adaptors: a handful of small launches in front of the call make the query rows), optionally with k / v adaptors on the local
        stage; no clip-scale on the global stage, none on an adapted or injected local stage."""
        lc, gc = self.local_compressor, self.global_compressor
        if not all(c is None or c.executor_ok for c in (lc, gc)) or self.global_logit is not None:
            return False
        return not (lc is not None and (lc.adapt_k or lc.adapt_v or not lc.queries_native) and self.local_logit is not None)

    def _external_queries(self) -> bool:
        """Some stage's queries are made in front of the executor call (engine.run_dense does; forward_deferred refuses such recipes)."""
        return any(c is not None and c.external_queries for c in (self.local_compressor, self.global_compressor))

    def _queries_native(self) -> bool:
        """No stage injects through an injector module or adapts its queries: what the frame-sharded executor phases can run (the
        others shard operator by operator, dist.sharded_forward_stepwise)."""
        return all(c is None or c.queries_native for c in (self.local_compressor, self.global_compressor))

    def _forward_with_logits(self, frames_feature, frames_embed, guide_embed, modal, image_newline, local_logits):
        lc = self.local_compressor
        if frames_embed is not None:
            raise ValueError("local_logits replaces frames_embed: pass frames_embed=None")
        if self._needs_grad(frames_feature, guide_embed, image_newline, local_logits):
            raise RuntimeError("local_logits= is an inference path (no autograd graph): call it under torch.no_grad()")
        nv.begin_inference()
        plain = all(c is None or c.is_plain for c in (lc, self.global_compressor))
        if (lc is None or self.global_compressor is None or lc.use_guide != "direct" or not plain or self.local_logit is not None
                or self.global_logit is not None or isinstance(frames_feature, dict) or not self.use_executor):
            raise NotImplementedError("local_logits= is built for the release recipe only (local + global compressor, "
                                      "use_guide='direct', no adaptors, no clip scale, dense video input)")
        if (not isinstance(local_logits, torch.Tensor) or local_logits.dtype != torch.float32 or not local_logits.is_cuda
                or tuple(local_logits.shape) != tuple(frames_feature.shape[:-1])):
            raise ValueError("local_logits: fp32 device tensor of shape frames_feature.shape[:-1]")
        from . import engine
        return engine.run_dense(self, frames_feature, None, guide_embed, modal, image_newline, _out_dtype(self),
                                local_logits=local_logits)

    def forward_deferred(self, frames_feature, frames_embed, guide_embed, modal, image_newline=None):
        """forward() without the final join of the side stream: returns (out, event).  The local rows of `out`
        are ordered on the caller's stream as usual; its 32 global rows are complete once `event` has fired
        (`torch.cuda.current_stream().wait_event(event)` before consuming them).  A serving loop that issues
        independent videos back to back hides the latency-bound global chain behind the next video's streaming.
        Inference only (no autograd graph)."""
        self._check_clip_logits()
        if self._needs_grad(frames_feature, frames_embed, guide_embed, image_newline):
            raise RuntimeError("forward_deferred is an inference API: call it under torch.no_grad() / inference_mode(), "
                               "or use forward() for training")
        nv.begin_inference()
        if not self._executor_covers() or self._external_queries() or isinstance(frames_feature, dict):
            raise NotImplementedError("forward_deferred: dense inputs of the recipes the one-call executor covers on its own")
        # (injected queries: their buffers belong to the plan, and a deferred call's side stream may still read them when the next
        # call refills them -- such recipes run joined)
        from . import engine
        return engine.run_dense(self, frames_feature, frames_embed, guide_embed, modal, image_newline, _out_dtype(self),
                                deferred=True)

    def forward_stepwise(self, frames_feature, frames_embed, guide_embed, modal, image_newline=None):
        """Same result, one C-ABI call per operator (anyres dict inputs; also the cross-check of the
        executor in the tests)."""
        lc, gc = self.local_compressor, self.global_compressor
        if (lc is not None and gc is not None and not isinstance(frames_feature, dict) and getattr(self, "overlap_stages", True)
                and not torch.cuda.is_current_stream_capturing()):
            return self._forward_stepwise_two_streams(frames_feature, frames_embed, guide_embed, modal, image_newline)
        segments = []        # (ctx, layout) per local segment, in output order
        if lc is not None:
            if isinstance(frames_feature, dict):                                 # anyres image (ref :679-689)
                if frames_feature["base"] is not None:
                    fe = frames_embed["base"].unsqueeze(0) if frames_embed is not None else None
                    ctx, grid = lc.window_context(frames_feature["base"].unsqueeze(0), fe, guide_embed, modal,
                                                  *self._logit_args("local"))
                    segments.append((ctx, self._layout(grid, modal, image_newline is not None, False)))
                fe = frames_embed["patch"].unsqueeze(0) if frames_embed is not None else None
                ctx, grid = lc.window_context(frames_feature["patch"].unsqueeze(0), fe, guide_embed, modal,
                                              *self._logit_args("local"))
                segments.append((ctx, self._layout(grid, modal, image_newline is not None, True)))
            else:
                ctx, grid = lc.window_context(frames_feature, frames_embed, guide_embed, modal,
                                              *self._logit_args("local"))
                _keep_train_ctx(lc, ctx)
                segments.append((ctx, self._layout(grid, modal, image_newline is not None, False)))
        n_local = sum(lay.n_rows for _, lay in segments)
        n_global = gc.num_queries if gc is not None else 0
        some = segments[0][0] if segments else (frames_feature["patch"] if isinstance(frames_feature, dict)
                                                else frames_feature)
        hidden = (lc or gc).readout[2].out_features
        out = torch.empty((n_local + n_global, hidden), dtype=_out_dtype(self), device=some.device)
        row = 0
        for ctx, lay in segments:
            lc.readout_into(ctx, out, row, lay.nl_group)
            if lay.newline_rows:
                nl = image_newline.contiguous()
                first = lay.newline_rows[0]
                step = lay.newline_rows[1] - first if len(lay.newline_rows) > 1 else 1
                nv.scatter_rows(nl.view(1, -1), out, row + first, len(lay.newline_rows), row_step=step)
            row += lay.n_rows
        if gc is not None:
            gff = frames_feature["patch"].unsqueeze(0) if isinstance(frames_feature, dict) else frames_feature
            gc.forward_into(gff, guide_embed, self._logit_args("global")[0], out, row, self._logit_args("global")[1])
        return out


def _keep_train_ctx(lc, ctx):
    """The training forward (autograd._CompressorFn) asks for the window contexts of an operator-by-operator forward: the backward's
    readout gradients need them, and recomputing them is a pass over every token."""
    hold = lc.__dict__.get("_train_ctx")
    if hold is not None:
        hold.append(ctx)


def _two_stream_forward(self, frames_feature, frames_embed, guide_embed, modal, image_newline=None):
    """Dense input, both compressors: the local chain (pooled / injected queries, adaptors, window attention, readout GEMMs) on a
    side stream beside the global chain (injected queries, streamed attention, merge, small linears) on the caller's stream -- the
    two are independent until the rows are in `out` (reference :691-707), and each is a string of launch-latency-sized kernels
    around one long one.  Same operators, same results as the one-stream form (`projector.overlap_stages = False`)."""
    from . import engine
    lc, gc = self.local_compressor, self.global_compressor
    ff = frames_feature
    _require_bf16_cuda("frames_feature", ff)
    dev = ff.device
    T, H, W, _ = ff.shape
    at, ay, ax = lc.tilings(T, H, W, modal)
    lay = self._layout((at.nwin, ay.nwin, ax.nwin), modal, image_newline is not None, False)
    hidden = lc.readout[2].out_features
    out = torch.empty((lay.n_rows + gc.num_queries, hidden), dtype=_out_dtype(self), device=dev)
    res = engine._resources(dev)
    main = torch.cuda.current_stream(dev)
    res.ev_fork.record(main)
    res.ev_fork.wait(res.side)
    with torch.cuda.stream(res.side):
        ctx, _ = lc.window_context(ff, frames_embed, guide_embed, modal, *self._logit_args("local"))
        _keep_train_ctx(lc, ctx)
        lc.readout_into(ctx, out, 0, lay.nl_group)
        if lay.newline_rows:
            nl = image_newline.contiguous()
            first = lay.newline_rows[0]
            step = lay.newline_rows[1] - first if len(lay.newline_rows) > 1 else 1
            nv.scatter_rows(nl.view(1, -1), out, first, len(lay.newline_rows), row_step=step)
        res.ev_join.record(res.side)
    # caller tensors and `out` are used on the side stream: keep the allocator from recycling them under it
    for t in (ff, frames_embed, guide_embed, image_newline, out):
        if isinstance(t, torch.Tensor):
            t.record_stream(res.side)
    gc.forward_into(ff, guide_embed, self._logit_args("global")[0], out, lay.n_rows, self._logit_args("global")[1])
    res.ev_join.wait(main)
    return out


HIComProjector._forward_stepwise_two_streams = _two_stream_forward


def build_vision_projector(config, delay_load=False, **kwargs):
    """Factory with the reference's type-string grammar (ref :231-304).  'mlpNx_gelu' / 'linear'
    are plain PyTorch modules in the reference too and are returned as such."""
    projector_type = getattr(config, "mm_projector_type", "linear")
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return build_mlp(int(m.group(1)), config.mm_hidden_size, config.hidden_size)
    if projector_type == "linear":
        return nn.Linear(config.mm_hidden_size, config.hidden_size)
    lspec, gspec = geo.parse_mm_projector_type(projector_type)
    local = glob = None
    if lspec is not None:
        local = LocalCompressor(config, lspec.temporal_kernel_size, lspec.spatial_kernel_size,
                                lspec.adapt_q, lspec.adapt_k, lspec.adapt_v, lspec.adapt_guide,
                                force_use_guide=lspec.force_use_guide)
    if gspec is not None:
        glob = GlobalCompressor(config, gspec.num_queries, True, gspec.adapt_guide,
                                force_use_guide=gspec.force_use_guide)
    return HIComProjector(config, local, glob)
