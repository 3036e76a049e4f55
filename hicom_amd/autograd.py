"""Training path of the drop-in projector: autograd through hicom_compressor_fwd.

The projector is the trainable module of the reference's stages 1-2 (scripts/qwen2.5_7B/release/
directg_local43_global32.sh:54,113; hicom/train.py:704-712), called under autograd + gradient checkpointing
(:78).  `forward_with_grad` returns exactly what the inference path returns (same kernels, same bits) wrapped
in a `torch.autograd.Function`; the backward is recompute-based (nothing but the inputs is kept alive, which is what
gradient checkpointing wants):

  streaming part (HIP)   window contexts recomputed by hicom_local_attn_fwd; the global logits / softmax state by
                         hicom_global_stream_fwd + merge; the attention backward over all T*729 tokens by
                         hicom_global_stream_bwd (one more pass over frames_feature: dS and sum dS . x)
  dense weight gradients plain library GEMMs / outer products through torch (rocBLAS) in fp32 on the small
                         [Nw, 1152] / [9, 1152] tensors -- dW = dY^T X is not a kernel worth hand-writing

Scope: every injection mode (direct, coarse, fine, off) and every adaptor (adapt q / k / v / guide) of dense video / image inputs,
gradients of every projector parameter and of `image_newline`; stage 3 also trains the SigLIP head and the guide encoder
(train.py:717-726), i.e. it needs the gradients w.r.t. `frames_embed` (the head's output, key stream of the local windows) and
`guide_embed`: built for direct / coarse / fine (hicom_local_attn_bwd: one more pass over both streams, d frames_embed written as
bf16; with k / v adaptors hicom_local_attn_adapt_bwd -- the window backward through the LayerNorm blends -- and the adaptor MLPs'
backward on the dense MFMA GEMM, NT and TN forms: round 4, no token-stream sized torch algebra left).  The query chain
(adapt_q, adapt_guide, coarse / fine injection) acts on small tensors only and is differentiated as a torch graph
(_query_chain_backward).  `frames_feature` comes from the frozen tower body: asking for its gradient raises instead of returning
None silently, and so do input gradients of the guide-off recipe, clip-scale projectors and text2qk projections.
"""
from __future__ import annotations

import math

import torch

from . import native as nv


LAST_FP32_GRADS = None      # test hook: the fp32 gradients of the last backward (before the cast to the parameter dtype)


def _gelu_bwd(dy, x):
    """dy * GELU'(x) (erf form) in ONE launch: aten's own backward kernel of nn.GELU (the eight elementwise launches of the written-out
    derivative were 77 us of a 1-ms training step)."""
    return torch.ops.aten.gelu_backward(dy.contiguous(), x, approximate="none")


def _sum0(x):
    """x.sum(0); for ONE row the row itself (a view: torch's reduction of a [1, n] tensor is still a ~10-us launch, and the direct
    recipe's global tail is one row throughout)."""
    return x[0] if x.shape[0] == 1 else x.sum(0)


def _supported(proj) -> bool:
    """Every injection mode and adaptor, clip-scale on either stage (round 6; on the local stage not together with k / v adaptors); not:
    text2qk projections (text width != query width)."""
    from .projector import _plain_injector
    lc, gc = proj.local_compressor, proj.global_compressor
    for c in (lc, gc):
        if c is None:
            continue
        if c.use_guide not in ("direct", None, "off", "coarse", "fine") or not _plain_injector(c.guide_injector):
            return False
    if proj.local_logit is not None and lc is not None and (lc.adapt_k or lc.adapt_v):
        return False
    return True


def _f32_params(proj):
    """{name: fp32 copy} of every projector parameter through ONE concatenation + ONE cast (two launches instead of one per
    tensor: the backward is launch-bound, a step changes every weight, so per-tensor cached casts would be rebuilt every step)."""
    items = list(proj.named_parameters())
    flat = torch.cat([p.detach().reshape(-1) for _, p in items]).float()
    out, o = {}, 0
    for n, p in items:
        out[n] = flat[o:o + p.numel()].view(p.shape)
        o += p.numel()
    return out


def _is_plain(proj) -> bool:
    return all(c is None or c.is_plain for c in (proj.local_compressor, proj.global_compressor))


def _ln_backward(g, xhat, rstd):
    """d input of y = xhat (row-normalised) given g = d y: rstd (g - mean g - xhat mean(g xhat))."""
    return rstd * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))


def _ln_stats(x, eps):
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
    return (x - mu) * rstd, rstd


def _adaptor_recompute(x2, mlp, out=None, a16=None):
    """Forward of one adaptor MLP over all tokens WITH what its backward needs, on the inference forward's own kernels and in its
    arithmetic (the GELU fused into the first GEMM's epilogue, fp16 hidden layer): returns (h1, None, y) -- h1 = W1 x + b1 BEFORE
    the activation (fp16, the epilogue's second output), y = W2 GELU(h1) + b2 (fp16, bit-identical to the inference forward's).
    The bf16 copy of the hidden activation that dW2 = dy^T a needs is made by the backward itself (hicom_gelu_split_fwd on h1).
    out = (h1, _, y) buffers to fill (the training forward's per-shape store), else fresh tensors."""
    from . import injector as inj
    N, D = x2.shape
    w1, b1 = mlp[0].weight.detach(), mlp[0].bias.detach()
    h1, _, y = out if out is not None else (torch.empty((N, w1.shape[0]), dtype=torch.float16, device=x2.device), None,
                                            torch.empty((N, mlp[2].weight.shape[0]), dtype=torch.float16, device=x2.device))
    a16 = torch.empty_like(h1) if a16 is None else a16
    nv.dense16_gemm(x2, w1, b1, act=nv.ACT_GELU, out_f16=a16, pre_f16=h1)
    nv.dense16_gemm(a16, inj._f16_weight(mlp[2]), mlp[2].bias.detach(), out_f16=y)
    return h1, None, y


class _AdaptorStore:
    """The k / v adaptor MLPs' intermediates of the LAST training forward of one input shape (pre-activation hidden layer h1 and
    output y per adapted stream, fp16: 2 x 107 MB each at 64 frames), in buffers that keep their addresses across steps -- the
    captured backward reads them by address.  `serial` counts the forwards that filled them: a backward whose forward was not the
    last one (two forwards before the first backward) refills them from its own inputs before it reads them."""

    def __init__(self, lc, N, E, dev):
        mk = lambda: torch.empty((N, E), dtype=torch.float16, device=dev)
        self.k = (mk(), None, mk()) if lc.adapt_k else None
        self.v = (mk(), None, mk()) if lc.adapt_v else None
        # the activated hidden layer(s) between the two GEMMs of a fill: scratch kept with the store (a fresh 107-MB temporary
        # per fill before); two of them when both streams are adapted -- their layers run as paired launches
        self.a16 = tuple(mk() for _ in range(2 if (lc.adapt_k and lc.adapt_v) else 1))
        self.serial = 0

    def fill(self, lc, ff, fe):
        from . import injector as inj
        E = ff.shape[-1]
        key = fe if fe is not None else ff
        if self.k is not None and self.v is not None:
            # both adaptors: each layer of the two MLPs as ONE launch of two problems (hicom_dense16_gemm_pair_fwd), in the inference
            # forward's arithmetic (GELU fused into the first layer's epilogue whose second output is the value before the activation)
            kp, vp = lc.k_proj, lc.v_proj
            nv.dense16_gemm_pair(key.reshape(-1, E), kp[0].weight.detach(), kp[0].bias.detach(), self.a16[0],
                                 ff.reshape(-1, E), vp[0].weight.detach(), vp[0].bias.detach(), self.a16[1],
                                 act=nv.ACT_GELU, pre_k=self.k[0], pre_v=self.v[0])
            nv.dense16_gemm_pair(self.a16[0], inj._f16_weight(kp[2]), kp[2].bias.detach(), self.k[2],
                                 self.a16[1], inj._f16_weight(vp[2]), vp[2].bias.detach(), self.v[2])
        else:
            if self.k is not None:
                _adaptor_recompute(key.reshape(-1, E), lc.k_proj, self.k, a16=self.a16[0])
            if self.v is not None:
                _adaptor_recompute(ff.reshape(-1, E), lc.v_proj, self.v, a16=self.a16[0])
        self.serial += 1
        return self.serial

    @property
    def ys(self):
        return (self.k[2] if self.k is not None else None, self.v[2] if self.v is not None else None)


class _GlobalStore:
    """Softmax state (M, L), un-normalised contexts and raw logits of the global stage of the LAST training forward of one input
    shape, in buffers with stable addresses (the captured backward reads them by address instead of streaming the tokens once more);
    `serial` as in _AdaptorStore."""

    def __init__(self):
        self.bufs = None
        self.serial = 0

    def buffers(self, R, rows_pad, stride, E, dev):
        if self.bufs is None or self.bufs[0].shape[0] != R or self.bufs[2].shape != (rows_pad, stride) or self.bufs[0].device != dev:
            f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
            self.bufs = (f(R, 2), f(R, E), f(rows_pad, stride))
        return self.bufs


def _global_store(proj, ff):
    """The global stage's store of this input shape, or None (no global stage; `proj.share_global_state = False`: the backward
    streams the tokens again, as gradient checkpointing would)."""
    if proj.global_compressor is None or getattr(proj, "share_global_state", True) is False:
        return None
    stores = proj.__dict__.setdefault("_global_stores", {})
    key = (tuple(ff.shape), str(ff.device))
    st = stores.get(key)
    if st is None:
        if len(stores) >= 2:
            stores.pop(next(iter(stores)))
        st = stores[key] = _GlobalStore()
    return st


def _adaptor_store(proj, ff):
    """The store of this input shape, or None when the recipe has no k / v adaptor (or the sharing is switched off:
    `proj.share_adaptor_activations = False` recomputes the MLPs in the backward, as gradient checkpointing would)."""
    lc = proj.local_compressor
    if lc is None or not (lc.adapt_k or lc.adapt_v) or getattr(proj, "share_adaptor_activations", True) is False:
        return None
    stores = proj.__dict__.setdefault("_adaptor_stores", {})
    key = (tuple(ff.shape), str(ff.device))
    st = stores.get(key)
    if st is None:
        if len(stores) >= 2:
            stores.pop(next(iter(stores)))
        st = stores[key] = _AdaptorStore(lc, ff.shape[0] * ff.shape[1] * ff.shape[2], ff.shape[3], ff.device)
    return st


def _adaptor_mlp_backward(x2, mlp, norm, alpha, rec, coef, vec, vec_stride, axes, prefix, which, grads, want_x):
    """Backward of (1 - a) x + a LN(MLP(x)) over ALL tokens (adapt_k / adapt_v, reference projector.py:533-534) for the rank-1 upstream
    gradient coef[tok] * vec[w(tok)] the window attention leaves (hicom_local_attn_adapt_bwd), on HIP kernels: hicom_adapt_dy_fwd (LayerNorm
    backward per token -> dy bf16), the weight gradients dW = dY^T X on the TN form of the dense MFMA GEMM, d hidden = dy W2 on its NT form,
    GELU' and the bias sums as streaming kernels.  Returns d x (bf16 [N, D]) when want_x."""
    h1, abf, y = rec
    N, D = x2.shape
    dev = x2.device
    if abf is None:                                                                         # GELU(h1) as bf16: operand of dW2
        abf = torch.empty(h1.shape, dtype=torch.bfloat16, device=dev)
        nv.gelu_split(h1, None, abf)
    dy = torch.empty((N, y.shape[1]), dtype=torch.bfloat16, device=dev)
    r1 = torch.empty((N, D), dtype=torch.bfloat16, device=dev) if want_x else None
    # (the bias gradients -- column sums of dy and of d h1 -- come out of the launches that write those matrices)
    grads[f"{prefix}{which}_proj.2.bias"] = nv.adapt_dy(y, norm.weight.detach(), vec, vec_stride, coef, alpha.detach(), axes, dy, r1, eps=norm.eps,
                                                        colsum=True)
    grads[f"{prefix}{which}_proj.2.weight"] = nv.dense16_tn(dy, abf)                       # dW2[o, h] = sum_n dy[n, o] a[n, h]
    da = torch.empty((N, h1.shape[1]), dtype=torch.bfloat16, device=dev)
    nv.dense16_gemm(dy, mlp[2].weight.detach().t().contiguous(), None, y=da)                # da[n, h] = sum_o dy[n, o] W2[o, h]
    can = da.shape[1] in (1152, 768)
    db1 = nv.gelu_bwd_(da, h1, colsum=can)                                                  # dh1 = da * GELU'(h1), in place
    grads[f"{prefix}{which}_proj.0.weight"] = nv.dense16_tn(da, x2)                         # dW1[h, i] = sum_n dh1[n, h] x[n, i]
    grads[f"{prefix}{which}_proj.0.bias"] = db1 if can else nv.colsum(da)
    if not want_x:
        return None
    dx = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
    nv.dense16_gemm(da, mlp[0].weight.detach().t().contiguous(), None, y=dx, res=r1)        # dx = dh1 W1 + (1 - a) coef vec
    return dx


def _coarse_backward(inj, prefix, vis, guide, dq, grads):
    """Backward of the FiLM injection q = LN(vis (1 + scale) + shift), (scale | shift) = coarse_proj(guide) (reference
    projector.py:369-372) for vis fp32 [M, D], guide [D]: fills the injector's parameter gradients, returns (d vis, d guide)."""
    D = vis.shape[-1]
    g = guide.reshape(-1).float()
    W1, b1 = inj.coarse_proj[0].weight.detach().float(), inj.coarse_proj[0].bias.detach().float()
    W2, b2 = inj.coarse_proj[2].weight.detach().float(), inj.coarse_proj[2].bias.detach().float()
    h1 = W1 @ g + b1
    a = torch.nn.functional.gelu(h1)
    cs = W2 @ a + b2
    sc, sh = cs[:D], cs[D:]
    zhat, rstd = _ln_stats(vis * (1.0 + sc) + sh, inj.coarse_norm.eps)
    grads[prefix + "coarse_norm.weight"] = (dq * zhat).sum(0)
    grads[prefix + "coarse_norm.bias"] = dq.sum(0)
    dz = _ln_backward(dq * inj.coarse_norm.weight.detach().float(), zhat, rstd)
    dcs = torch.cat([(dz * vis).sum(0), dz.sum(0)])
    grads[prefix + "coarse_proj.2.weight"] = torch.outer(dcs, a)
    grads[prefix + "coarse_proj.2.bias"] = dcs
    dh1 = _gelu_bwd(W2.t() @ dcs, h1)
    grads[prefix + "coarse_proj.0.weight"] = torch.outer(dh1, g)
    grads[prefix + "coarse_proj.0.bias"] = dh1
    return dz * (1.0 + sc), W1.t() @ dh1


def _query_chain_backward(stage, prefix, vis, guide, d_inj, f32, grads, want_guide, vis_param=None, want_vis=False):
    """Backward of the QUERY side of a compressor stage: the adapt_q blend (reference projector.py:541), the adapt_guide blend
    (:365 / :389) and direct / coarse / fine injection (:352-397).  These act on small tensors only -- the [Nw | 32, D] queries and
    the 1..64 guide rows, never the token stream -- and are restated here as a torch graph over fp32 leaf copies of the stage's
    parameters, differentiated by torch.autograd.grad.  vis: fp32 [M, D] pooled queries / learnable queries (None for direct);
    d_inj: gradient of the injected queries ([M, D]; [1, D] for the shared direct query).  Fills `grads`, returns (d vis, d guide)."""
    F = torch.nn.functional
    inj = stage.guide_injector
    mode = stage.use_guide if stage.use_guide not in (None, "off") else None
    eps = 1e-6
    leaves = {}

    def leaf(name):
        if name not in leaves:
            leaves[name] = f32[prefix + name].detach().clone().requires_grad_(True)
        return leaves[name]

    def mlp2(x, base):
        return F.linear(F.gelu(F.linear(x, leaf(base + ".0.weight"), leaf(base + ".0.bias"))), leaf(base + ".2.weight"), leaf(base + ".2.bias"))

    def ln(x, base):
        return F.layer_norm(x, (x.shape[-1],), leaf(base + ".weight"), leaf(base + ".bias"), eps)

    with torch.enable_grad():
        v = None
        if vis is not None:
            v = vis.detach().clone().requires_grad_(vis_param is not None or want_vis)
            vq = v
            if getattr(stage, "adapt_q", False):                                   # (1 - a) q + a LN(q W^T)  (:541; Linear without bias)
                qa = leaf("q_alpha")
                vq = (1 - qa) * vq + qa * ln(F.linear(vq, leaf("q_proj.weight")), "q_norm")
        g = None
        if mode is not None:
            g = guide.detach().float().clone().requires_grad_(bool(want_guide))
            ga = g
            if getattr(stage, "adapt_guide", False):                               # (:365 / :389)
                al = leaf("guide_injector.guide_alpha")
                ga = (1 - al) * ga + al * ln(mlp2(ga, "guide_injector.guide_proj"), "guide_injector.guide_norm")
        if mode is None:
            q = vq
        elif mode == "direct":
            q = ga.reshape(1, -1)
        elif mode == "coarse":
            cs = mlp2(ga.reshape(1, -1), "guide_injector.coarse_proj")
            D = vq.shape[-1]
            q = ln(vq * (1 + cs[:, :D]) + cs[:, D:], "guide_injector.coarse_norm")
        else:                                                                      # fine (:374-397): LN(q + MHA(q, G, G))
            att = inj.fine_proj
            nh, hd = att.num_heads, att.head_dim
            P = "guide_injector.fine_proj."
            qp = F.linear(vq, leaf(P + "q_proj.weight"), leaf(P + "q_proj.bias")).view(-1, nh, hd).permute(1, 0, 2)
            kp = F.linear(ga, leaf(P + "k_proj.weight"), leaf(P + "k_proj.bias")).view(-1, nh, hd).permute(1, 0, 2)
            vp = F.linear(ga, leaf(P + "v_proj.weight"), leaf(P + "v_proj.bias")).view(-1, nh, hd).permute(1, 0, 2)
            pr = torch.softmax(torch.matmul(qp, kp.transpose(1, 2)) * att.scale, dim=-1)
            o = torch.matmul(pr, vp).permute(1, 0, 2).reshape(vq.shape[0], nh * hd)
            q = ln(vq + F.linear(o, leaf(P + "out_proj.weight"), leaf(P + "out_proj.bias")), "guide_injector.fine_norm")
        wrt = list(leaves.values())
        names = list(leaves.keys())
        extra = []
        if g is not None and want_guide:
            extra.append(g)
        if v is not None and (vis_param is not None or want_vis):
            extra.append(v)
        if not wrt and not extra:
            return None, None
        got = torch.autograd.grad(q, wrt + extra, grad_outputs=d_inj.reshape(q.shape), allow_unused=True)
    for n, gr in zip(names, got[:len(names)]):
        if gr is not None:
            grads[prefix + n] = gr
    rest = list(got[len(names):])
    d_guide = rest.pop(0) if (g is not None and want_guide) else None
    d_vis = rest.pop(0) if (v is not None and (vis_param is not None or want_vis)) else None
    if d_vis is not None and vis_param is not None:
        grads[prefix + vis_param] = d_vis
    return d_vis, d_guide


class _CompressorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, ff, fe, guide, modal, nl, names, *params):
        from . import engine
        from .projector import _out_dtype
        ctx.adapt_serial = ctx.global_serial = None
        ctx16 = None
        with torch.no_grad():
            gc_ = proj.global_compressor
            # guide off (32 learnable queries x 9 heads): operator by operator, so that the global stage's state and logits stay for the
            # backward (_GlobalStore) -- the executor keeps them in its workspace only
            # (the same for coarse / fine: 32 injected rows; in inference those recipes take the executor with their query rows made in
            # front of the call, engine.run_dense)
            many_rows = gc_ is not None and (gc_.use_guide in (None, "off") or not gc_.queries_native) and _global_store(proj, ff) is not None
            if proj.use_executor and proj._executor_covers() and not many_rows:          # (plain recipes and the k / v adaptors: one C call)
                store = _adaptor_store(proj, ff)
                if store is not None:
                    # the adaptor MLPs run here, with the intermediates their backward needs kept (as autograd keeps them in the
                    # reference); the executor gets their outputs instead of running the four GEMMs itself
                    ctx.adapt_serial = store.fill(proj.local_compressor, ff, fe)
                out = engine.run_dense(proj, ff, fe, guide, modal, nl, _out_dtype(proj), adapt_y=store.ys if store is not None else None)
                # the window contexts as the forward's readout consumed them (fp16 plane in the executor's workspace): the backward's
                # readout gradients take them instead of a pass over every token that recomputes them (55 us of a 1.0-ms release step,
                # 96 us with k / v adaptors)
                ctx16 = engine.last_window_contexts(proj, ff, modal) if getattr(proj, "share_window_contexts", True) else None
            else:                                                      # query-side adaptors / coarse / fine injection: operator by operator
                gstore = _global_store(proj, ff)
                gc = proj.global_compressor
                if gstore is not None:
                    gc.__dict__["_train_store"] = gstore               # partial_context fills it (and asks the kernel for the logits)
                lc_ = proj.local_compressor
                hold = [] if (lc_ is not None and getattr(proj, "share_window_contexts", True)) else None
                if hold is not None:
                    lc_.__dict__["_train_ctx"] = hold                  # (the forward's own fp32 window contexts: what the backward would recompute, bit for bit)
                try:
                    out = proj.forward_stepwise(ff, fe, guide, modal, nl)
                finally:
                    if gstore is not None:
                        gc.__dict__.pop("_train_store", None)
                    if hold is not None:
                        lc_.__dict__.pop("_train_ctx", None)
                if hold:
                    ctx16 = hold[0]
                    ctx16.record_stream(torch.cuda.current_stream(ff.device))     # (made on the side stream of the two-stream forward)
                if gstore is not None and gstore.bufs is not None:
                    ctx.global_serial = gstore.serial
        ctx.proj, ctx.modal, ctx.names = proj, modal, names
        ctx.save_for_backward(ff, fe, guide, nl, ctx16)
        return out

    @staticmethod
    def backward(ctx, dout):
        ff, fe, guide, nl, ctx16 = ctx.saved_tensors
        need = ctx.needs_input_grad            # (proj, ff, fe, guide, modal, nl, names, *params)
        proj = ctx.proj
        if need[1] and not _ff_grad_supported(proj):
            raise NotImplementedError("hicom_amd backward: the gradient w.r.t. frames_feature (`pure_vision_model`, reference train.py:712-715) "
                                      "is not built beside clip-scale; detach it otherwise")
        want = tuple(bool(need[7 + k]) for k in range(len(ctx.names)))
        args = (proj, ff, fe, guide, ctx.modal, nl, ctx.names, want, bool(need[2]), bool(need[3]), bool(nl is not None and need[5]))
        store = _adaptor_store(proj, ff) if ctx.adapt_serial is not None else None
        if store is not None and store.serial != ctx.adapt_serial:
            with torch.no_grad():
                store.fill(proj.local_compressor, ff, fe)          # another forward of this shape ran in between: its intermediates are not ours
        gstore = _global_store(proj, ff) if ctx.global_serial is not None else None
        if gstore is not None and (gstore.bufs is None or gstore.serial != ctx.global_serial):
            with torch.no_grad():                                  # another forward of this shape ran in between: stream again, into the store
                gc = proj.global_compressor
                gc.__dict__["_train_store"] = gstore
                try:
                    gc.partial_context(ff, gc.injected_queries(guide)[0])
                finally:
                    gc.__dict__.pop("_train_store", None)
        gb = getattr(proj, "graph_backward", None)             # None: automatic; False: always eager
        d_ff = None
        if need[1]:
            # d frames_feature (round 6): the eager backward (the tower body trains in no stage of the reference's scripts: not worth a graph)
            hold = {}
            with torch.no_grad():
                res = _backward_outputs(dout, *args, store=store, gstore=gstore, ctx16=ctx16, ff_grad=hold)
            d_ff = hold["d_ff"]
        elif (gb is None or gb) and proj.local_logit is None and proj.global_logit is None:
            # (clip-scale: the logits are TRAINABLE numbers the launches take by value -- a captured graph would keep replaying the values
            # of the step it was captured in; that recipe takes the eager backward)
            res = _graphed_backward(dout, *args, store=store, gstore=gstore, ctx16=ctx16)
        else:
            with torch.no_grad():
                res = _backward_outputs(dout, *args, store=store, gstore=gstore, ctx16=ctx16)
        flats, d_fe, d_guide, d_nl = res
        plist = dict(proj.named_parameters())
        out = [None] * len(ctx.names)
        for dt, (flat, group) in flats.items():                    # views of ONE buffer per parameter dtype
            o = 0
            for k, name in group:
                n = plist[name].numel()
                out[k] = flat[o:o + n].view(plist[name].shape)
                o += n
        return (None, d_ff, d_fe, d_guide, None, d_nl, None, *out)


def _ff_grad_supported(proj) -> bool:
    """d frames_feature: every injection mode (direct, off, coarse, fine) and the query-side adaptors -- the window backward leaves the value-side
    (and, without frames_embed, key-side) gradient per token, the pooled per-window queries send theirs back through the trilinear pooling
    (reference projector.py:539-540), the global stage's is dS^T qt + P^T dctx; with k / v adaptors the streams' input gradients come out of
    the adaptor MLPs' backward (their skip path included).  Not with clip-scale."""
    lc, gc = proj.local_compressor, proj.global_compressor
    if proj.local_logit is not None or proj.global_logit is not None:
        return False
    return True


def _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl, store=None, gstore=None, ctx16=None,
                      ff_grad=None):
    """({dtype: (flat gradient buffer, [(argument index, parameter name)])}, d frames_embed, d guide_embed, d image_newline) in
    the dtypes autograd hands on.  The parameter gradients leave as views of one buffer cast once (one concatenation + one cast
    instead of a cast per tensor)."""
    global LAST_FP32_GRADS
    grads, d_nl, d_fe, d_guide = compressor_backward(proj, ff, fe, guide, modal, nl, dout, want_fe=want_fe, want_guide=want_guide,
                                                     adaptor_saved=(store.k, store.v) if store is not None else None,
                                                     global_saved=gstore.bufs if gstore is not None else None, ctx_local16=ctx16,
                                                     ff_grad=ff_grad)
    LAST_FP32_GRADS = dict(grads)
    if d_guide is not None:
        LAST_FP32_GRADS["__guide_embed__"] = d_guide
    plist = dict(proj.named_parameters())
    by_dtype = {}
    for k, name in enumerate(names):
        if want[k] and grads.get(name) is not None:
            by_dtype.setdefault(plist[name].dtype, []).append((k, name))
    flats = {dt: (torch.cat([grads[name].reshape(-1) for _, name in group]).to(dt), group) for dt, group in by_dtype.items()}
    return (flats, d_fe, (d_guide.to(guide.dtype).reshape(guide.shape) if d_guide is not None else None),
            (d_nl.to(nl.dtype) if (want_nl and d_nl is not None) else None))


def _reads_frames_embed(proj, want_fe, want_guide, have_ctx) -> bool:
    """Whether compressor_backward reads frames_embed: for the window contexts (unless the forward kept them) and in the window-attention
    backward (input gradients, adaptors, injected / adapted queries)."""
    lc = proj.local_compressor
    if lc is None:
        return False
    mode = lc.use_guide if lc.use_guide not in (None, "off") else None
    query_params = mode in ("coarse", "fine") or lc.adapt_q or lc.adapt_guide
    return (not have_ctx) or want_fe or lc.adapt_k or lc.adapt_v or query_params or (want_guide and mode is not None) or proj.local_logit is not None


_MAX_BWD_GRAPHS = 4


def _graphed_backward(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl, store=None, gstore=None, ctx16=None):
    """The backward as a captured hipGraph (the default; `proj.graph_backward = False` turns it off).
    The eager backward is ~130 small launches behind 1.4 ms of Python at the benchmark shape; its shapes are static, so the second
    backward of a problem SHAPE is captured and later ones are one graph launch.  The captured kernels read STATIC copies of the inputs
    (frames_feature, frames_embed, guide: one device copy each per step, ~0.1 ms at 64 frames) -- round 3 keyed the graph on the callers'
    buffer addresses, which only held while the allocator handed the same blocks back.  The upstream gradient is copied into a static buffer
    too, the results are cloned out of the graph's pool (gradient accumulation adds into .grad in place: handing out the static buffers would
    alias them).  Keyed by shapes and by what is asked for; dropped when the parameters' storage or a cached device table moves
    (`engine.plan_sig`).  Falls back to the eager backward on any capture error."""
    from . import engine
    cache = proj.__dict__.setdefault("_bwd_graphs", {})
    key = (tuple(ff.shape), None if fe is None else tuple(fe.shape), None if guide is None else tuple(guide.shape), modal,
           tuple(dout.shape), dout.dtype, want, want_fe, want_guide, None if nl is None else (tuple(nl.shape), nl.dtype, want_nl),
           torch.cuda.current_stream(ff.device).cuda_stream,
           None if ctx16 is None else ctx16.dtype,
           None if store is None else id(store),          # (the captured kernels read the stores' buffers by address)
           None if gstore is None else tuple(b.data_ptr() for b in gstore.bufs))
    # what else the captured kernels read by ADDRESS: every parameter's storage and the cached device tables (pe / kpe / planes:
    # `_cache_gen` moves when one is reallocated -- T above the cached cap, a cleared cache, a device move).  A graph whose
    # signature moved is dropped, never replayed over freed or reused memory.
    sig = engine.plan_sig(proj)
    ent = cache.get(key)
    if ent is not None and ent["sig"] != sig:
        cache.pop(key)
        ent = None

    def eager():
        with torch.no_grad():
            return _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl, store=store, gstore=gstore, ctx16=ctx16)

    if ent is None:                                                # first sight of the shape: eager (also the warm-up a capture needs)
        if len(cache) >= _MAX_BWD_GRAPHS:
            cache.pop(next(iter(cache)))
        cache[key] = {"seen": 1, "sig": sig}
        return eager()
    if "graph" not in ent:
        if ent.get("failed"):
            return eager()
        try:
            gc = proj.global_compressor
            st = {"ff": ff.detach().clone(), "fe": None if fe is None else fe.detach().clone(),
                  "guide": None if guide is None else guide.detach().clone(), "dout": dout.detach().clone(),
                  "ctx16": None if ctx16 is None else ctx16.detach().clone(),
                  "tables": None if gc is None else dict(gc._pe_cache)}       # (the cached tables the kernels read stay alive with the entry)
            torch.cuda.current_stream(ff.device).synchronize()
            g = torch.cuda.CUDAGraph()
            global _ROW_HOLD
            _ROW_HOLD = st["nl_rows"] = []                             # (index tensors the captured gathers read stay alive with the entry)
            try:
                with torch.no_grad(), torch.cuda.graph(g):
                    outs = _backward_outputs(st["dout"], proj, st["ff"], st["fe"], st["guide"], modal, nl, names, want, want_fe, want_guide, want_nl,
                                             store=store, gstore=gstore, ctx16=st["ctx16"])
            finally:
                _ROW_HOLD = None
            st["store"] = (store, gstore, None if gstore is None else gstore.bufs)   # (keeps the buffers the graph reads alive with the entry)
            if engine.plan_sig(proj) != sig:                       # (the pass inside the capture reallocated a table)
                raise RuntimeError("cached device tables moved during capture")
            ent.update(graph=g, outs=outs, **st)
        except Exception as e:  # noqa: BLE001
            ent["failed"] = repr(e)
            import warnings
            warnings.warn(f"hicom_amd: capturing the backward of shape {tuple(ff.shape)} into a hipGraph failed ({e!r}); this shape keeps "
                          "the eager backward (~2x slower per training step)", RuntimeWarning, stacklevel=2)
            return eager()
    ent["ff"].copy_(ff)
    if fe is not None and _reads_frames_embed(proj, want_fe, want_guide, ctx16 is not None):
        ent["fe"].copy_(fe)                                        # (107 MB at 64 frames: only when a captured kernel reads it)
    if guide is not None:
        ent["guide"].copy_(guide)
    ent["dout"].copy_(dout)
    if ctx16 is not None:
        ent["ctx16"].copy_(ctx16)
    ent["graph"].replay()
    flats, d_fe, d_guide, d_nl = ent["outs"]
    # (image_newline -- the reference's scripts always pass it, hicom_arch.py:212, also where mm_newline_position = "no_token" leaves it
    # unused -- enters the backward by shape and dtype only: d image_newline is a sum of cotangent rows)
    return ({dt: (flat.clone(), group) for dt, (flat, group) in flats.items()}, None if d_fe is None else d_fe.clone(),
            None if d_guide is None else d_guide.clone(), None if d_nl is None else d_nl.clone())


_ROW_INDEX = {}
_ROW_HOLD = None            # while a backward is being captured: the index tensors it read (kept alive with the graph entry)


def _row_index(rows, dev):
    """Device index tensor of a packing's newline rows, kept per (rows, device): the captured backward must not make it (a host-to-device
    copy inside a stream capture); the eager pass that precedes every capture does.  The dict is a 64-entry LRU; a captured graph reads
    the tensor BY ADDRESS, so the capture also takes a reference of its own (`_ROW_HOLD` -> the graph entry, ADVICE r5: an entry evicted
    by 64 other layouts was freed under a graph that still gathered through it)."""
    key = (tuple(rows), str(dev))
    t = _ROW_INDEX.pop(key, None)
    if t is None:
        if len(_ROW_INDEX) >= 64:
            _ROW_INDEX.pop(next(iter(_ROW_INDEX)))
        t = torch.tensor(list(rows), device=dev)
    _ROW_INDEX[key] = t                                            # (re-inserted: most recently used last)
    if _ROW_HOLD is not None:
        _ROW_HOLD.append(t)
    return t


def _global_clip_backward(ff, pe, pos0, scores, ds, R, nq, nh, hd, q32, Wq, bq, Wk, bk, log_scale, grads, P):
    """Clip-scale on the GLOBAL stage under autograd (reference projector.py:184-191; `global_logit_scale` / `global_logit_bias` are trainable
    under `attn_scale`, train.py:730-733): projected queries and keys are L2-normalised over the FULL width before the heads are split,
        s[(q,h), n] = e^ls  qhat_{q,h} . k_{n,h} / ||k_n|| + lb,     k_n = W_k (x_n + pe_n) + b_k,     qhat_q = (W_q q + b_q) / ||.||
    Given dS (= p (dP - delta), out of the stream backward kernel) this fills d logit_scale / d logit_bias and d k_proj, and returns the
    gradient w.r.t. the RAW projected queries W_q q + b_q (the caller goes on from there exactly as without clip-scale).  With
    inv_n = 1 / ||k_n||, u = s / inv (the un-normalised logits) and du = dS inv:
        d qt_r = sum_n du[r,n] (x_n + pe_n)   (qt_r = e^ls W_k,h^T qhat_h: the fold),      d c_r = sum_n du[r,n]   (c_r = e^ls qhat_h . b_k,h)
        d inv_n = sum_r dS[r,n] u[r,n]   =>   d k_n (norm path) = w_n k_n,   w_n = -inv_n^2 sum_r dS[r,n] s[r,n]
        d W_k = e^ls qhat (x) d qt  +  W_k G + b_k g^T,     G = sum_n w_n x'_n x'_n^T,   g = sum_n w_n x'_n       (x' = x + pe)
        d b_k = e^ls qhat d c       +  W_k g + b_k sum_n w_n
    The token-stream sized products (k, G) are plain library GEMMs in fp32: this recipe appears in none of the reference's scripts."""
    T, H, W, E = ff.shape
    N = T * H * W
    dev = ff.device
    sc = math.exp(log_scale)
    xp = ff.view(N, E).float()
    if pe is not None:
        t0i, y0i, x0i = pos0
        xp = (xp.view(T, H, W, E) + pe[t0i:t0i + T].view(T, 1, 1, E) + pe[y0i:y0i + H].view(1, H, 1, E) + pe[x0i:x0i + W].view(1, 1, W, E)).view(N, E)
    bk_ = bk if bk is not None else torch.zeros(E, dtype=torch.float32, device=dev)
    k = torch.addmm(bk_, xp, Wk.t())                                       # [N, E]
    inv = 1.0 / k.norm(dim=-1)
    del k
    S, dS = scores[:R, :N], ds[:R, :N]
    du = dS * inv.unsqueeze(0)
    dqt = (du @ xp).view(nq, nh, E)
    dc = du.sum(1).view(nq, nh)
    wn = -(inv * inv) * (dS * S).sum(0)                                    # [N]
    xw = xp * wn.unsqueeze(1)
    G = xw.t() @ xp                                                        # [E, E]
    g = xw.sum(0)
    del xw
    qraw = q32 @ Wq.t() + bq
    nrm = qraw.norm(dim=-1, keepdim=True)
    qhat = qraw / nrm
    qh = qhat.view(nq, nh, hd)
    grads[P + "attn_layer.k_proj.weight"] = sc * torch.einsum("qhj,qhe->hje", qh, dqt).reshape(E, E) + Wk @ G + torch.outer(bk_, g)
    if bk is not None:
        grads[P + "attn_layer.k_proj.bias"] = sc * torch.einsum("qhj,qh->hj", qh, dc).reshape(E) + Wk @ g + bk_ * wn.sum()
    dqhat = sc * (torch.einsum("hje,qhe->qhj", Wk.view(nh, hd, E), dqt) + bk_.view(1, nh, hd) * dc.unsqueeze(2)).reshape(nq, E)
    grads["global_logit_scale"] = (dS * S).sum().reshape(1)                # d s / d ls = s - lb, and the logits here are kept without lb
    grads["global_logit_bias"] = torch.zeros(1, dtype=torch.float32, device=dev)   # a per-row shift: softmax cancels it
    return (dqhat - qhat * (qhat * dqhat).sum(-1, keepdim=True)) / nrm


def compressor_backward(proj, ff, fe, guide, modal, nl, dout, want_fe=False, want_guide=False, adaptor_saved=None, global_saved=None,
                        stages=("local", "global"), is_anyres=False, ctx_local16=None, ff_grad=None):
    """(fp32 gradients {parameter name: tensor} of sum(out * dout), d image_newline, d frames_embed (bf16) or None,
    d guide_embed (fp32) or None).  Restates autograd through reference projector.py:524-559 (local), :634-646 + :166-228
    (global) and mm_utils.py:92-140 (packing).  The input gradients exist for the direct recipe only.
    `stages` / `is_anyres`: one SEGMENT of an anyres dict input (reference :679-689: the base image goes through the local stage
    only, the patch grid through both, packed with the anyres layout); `dout` holds that segment's rows."""
    lc = proj.local_compressor if "local" in stages else None
    gc = proj.global_compressor if "global" in stages else None
    dev = ff.device
    d_fe = d_guide = None
    if want_fe or want_guide:
        if want_guide and not any(c is not None and c.use_guide in ("direct", "coarse", "fine") for c in (lc, gc)):
            want_guide = False         # guide off: guide_embed does not enter the forward: no gradient (None), as in the reference
        if want_guide:
            d_guide = torch.zeros(guide.shape, dtype=torch.float32, device=dev)
        if want_fe and lc is None:
            want_fe = False            # without a local stage frames_embed does not enter the forward: no gradient (None), as in the reference
        # (guide off with a local stage: frames_embed are the window keys, reference :544-551 -- d frames_embed[n] = dS_n q_w with the
        # pooled query of the token's window, the same kernel as the direct recipe's with one query row per window)
    dout = dout.float()
    f32 = _f32_params(proj)
    T, H, W, E = ff.shape
    grads = {}
    d_nl = None
    n_local = 0
    if lc is not None:
        at, ay, ax = lc.tilings(T, H, W, modal)
        grid = (at.nwin, ay.nwin, ax.nwin)
        lay = proj._layout(grid, modal, nl is not None, is_anyres)
        nw = grid[0] * grid[1] * grid[2]
        n_local = lay.n_rows
        idx = torch.arange(nw, device=dev)
        rows = idx + (idx // lay.nl_group if lay.nl_group else 0)          # packing row map of the readout store
        dY = dout[rows]
        if lay.newline_rows:
            d_nl = dout[_row_index(lay.newline_rows, dev)].sum(0)
        # k / v adaptors: ONE recomputation of the two MLPs over all tokens (with the intermediates their backward needs) serves the
        # window contexts below as well
        adapt = lc.adapt_k or lc.adapt_v
        rec_k = rec_v = None
        if adapt and adaptor_saved is not None:
            rec_k, rec_v = adaptor_saved                                   # kept by the training forward (_AdaptorStore)
        elif adapt:
            key_ = fe if fe is not None else ff
            rec_k = _adaptor_recompute(key_.reshape(-1, E), lc.k_proj) if lc.adapt_k else None
            rec_v = _adaptor_recompute(ff.reshape(-1, E), lc.v_proj) if lc.adapt_v else None
        if ctx_local16 is not None:                                         # kept by the training forward (the executor's fp16 plane,
            ctx_l = ctx_local16.float()                                     # or the operator-by-operator forward's own fp32 contexts)
        else:
            ctx_l, _ = lc.window_context(ff, fe, guide, modal, *proj._logit_args("local"),     # HIP: [Nw, E] fp32 window contexts
                                         adapt_y=(rec_k[2] if rec_k else None, rec_v[2] if rec_v else None) if adapt else None)
        W0, b0 = f32["local_compressor.readout.0.weight"], f32["local_compressor.readout.0.bias"]
        W2 = f32["local_compressor.readout.2.weight"]
        pre = torch.addmm(b0, ctx_l, W0.t())
        h = torch.nn.functional.gelu(pre)
        grads["local_compressor.readout.2.weight"] = dY.t() @ h
        grads["local_compressor.readout.2.bias"] = dY.sum(0)
        dpre = _gelu_bwd(dY @ W2, pre)
        grads["local_compressor.readout.0.weight"] = dpre.t() @ ctx_l
        grads["local_compressor.readout.0.bias"] = dpre.sum(0)
        mode = lc.use_guide if lc.use_guide not in (None, "off") else None
        plain_q = not (lc.adapt_q or lc.adapt_guide)                       # plain direct / coarse / off: the hand-written paths below
        query_params = mode in ("coarse", "fine") or lc.adapt_q or lc.adapt_guide
        # clip-scale on the local stage (reference projector.py:527-529, :549; `local_logit_scale` / `local_logit_bias` are trainable under
        # `attn_scale`, train.py:730-733): frames_embed and guide_embed enter L2-normalised (only when frames_embed is given), the logits
        # are e^ls (q . khat) + lb.  The window backward takes the normalisation of the keys itself (l2norm_key); the guide's -- a
        # handful of rows -- is differentiated here: ghat = g / ||g||, d g = (d ghat - ghat (ghat . d ghat)) / ||g||.
        clip = proj.local_logit if "local" in stages else None
        l2k = clip is not None and fe is not None
        guide_q, g_nrm = guide, None
        if l2k and mode is not None:
            g_rows = guide.detach().float().reshape(-1, E)
            g_nrm = g_rows.norm(dim=-1, keepdim=True)
            guide_q = (g_rows / g_nrm).reshape(guide.shape)

        def through_guide_norm(dgh):
            """d guide_embed from the gradient w.r.t. the normalised guide rows (identity without clip-scale)."""
            if g_nrm is None or dgh is None:
                return dgh
            gh = guide_q.reshape(-1, E)
            dgh = dgh.reshape(-1, E)
            return ((dgh - gh * (gh * dgh).sum(-1, keepdim=True)) / g_nrm).reshape(guide.shape)

        if want_fe or adapt or query_params or (want_guide and mode is not None) or clip is not None or ff_grad is not None:
            # ---- attention backward of the windows: dq per window, d key stream, d value stream -------------------------
            from . import injector as inj
            exact = all(a.nwin * a.k == a.n for a in (at, ay, ax))
            if adapt and not exact:
                raise NotImplementedError("hicom_amd backward: the adaptor gradients need an exact window "
                                          f"partition (T, H, W = {T}, {H}, {W} against kernel {at.k}, {ay.k}, {ax.k})")
            axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
            dctx_l = (dpre @ W0).contiguous()
            scale, bias = (math.exp(clip[0]), float(clip[1])) if clip is not None else (1.0 / math.sqrt(lc.qk_dim), 0.0)
            key = fe if fe is not None else ff
            pooled = None
            if mode == "direct":
                q, _ = inj.inject(lc.guide_injector, "direct", None, guide_q.contiguous())     # the (adapted) guide, one row
                q = q.reshape(-1).contiguous()
            else:
                pooled = torch.empty((*grid, E), dtype=torch.float32, device=dev)
                nv.trilinear_pool(ff, pooled)                              # HIP: the per-window query (ref :539-540)
                pooled = pooled.view(nw, E)
                q = pooled
                if lc.adapt_q:
                    q = inj.adapt_query(q, lc.q_proj, lc.q_norm, lc.q_alpha)                    # HIP (ref :541)
                if mode in ("coarse", "fine"):
                    q, _ = inj.inject(lc.guide_injector, mode, q.reshape(nw, E), guide_q.contiguous())   # HIP (ref :369-397)
            if adapt:
                # ---- k / v adaptors: window-attention backward with the blends (two streaming passes over x_k, y_k, x_v, y_v), then
                # the adaptor MLPs' backward per stream; everything token-stream sized runs on HIP kernels --------------------------
                N = T * H * W
                kx2, vx2 = key.reshape(N, E), ff.reshape(N, E)
                ds = torch.empty((N,), dtype=torch.float32, device=dev)
                pw = torch.empty_like(ds)
                sxk, sxv = torch.empty((nw, E), dtype=torch.float32, device=dev), torch.empty((nw, E), dtype=torch.float32, device=dev)
                syk = torch.empty_like(sxk) if lc.adapt_k else None
                syv = torch.empty_like(sxv) if lc.adapt_v else None
                qc = q.contiguous()
                shared_q = qc.ndim == 1 or qc.shape[0] == 1
                q_stride = 0 if shared_q else E
                nv.local_attn_adapt_bwd(key, rec_k[2] if rec_k else None, lc.k_norm if rec_k else None, lc.k_alpha.detach() if rec_k else None,
                                        ff, rec_v[2] if rec_v else None, lc.v_norm if rec_v else None, lc.v_alpha.detach() if rec_v else None,
                                        axes, qc, q_stride, scale, 0.0, dctx_l, ds, pw, sxk, syk, sxv, syv,
                                        eps=(lc.k_norm if rec_k else lc.v_norm).eps)
                L = "local_compressor."
                qf = qc.float().reshape(1 if shared_q else nw, E)
                d_kx = None
                # the key stream's input gradient is wanted for d frames_embed, or -- without frames_embed the keys are frames_feature rows
                # (reference :532) -- for d frames_feature
                want_kx = (want_fe and fe is not None) or (ff_grad is not None and fe is None)
                if ff_grad is not None and not exact:
                    raise NotImplementedError("hicom_amd backward: d frames_feature needs an exact window partition")
                if lc.adapt_k:
                    ak, gk = lc.k_alpha.detach().float(), f32[L + "k_norm.weight"]
                    dq_w = (1.0 - ak) * sxk + ak * gk * syk
                    grads[L + "k_alpha"] = (qf * (gk * syk - sxk)).sum().reshape(1)
                    grads[L + "k_norm.weight"] = ak * (qf * syk).sum(0)
                    grads[L + "k_norm.bias"] = torch.zeros(E, dtype=torch.float32, device=dev)    # a k sum_w q_w sum_n dS_n: the dS of a window sum to zero
                    d_kx = _adaptor_mlp_backward(kx2, lc.k_proj, lc.k_norm, lc.k_alpha, rec_k, ds, qc, q_stride, axes, L, "k", grads, want_kx)
                else:
                    dq_w = sxk
                    if want_kx:
                        # plain key stream beside a value adaptor: d key_n = ds_n q_w, written by the rank-1 branch of hicom_adapt_dy_fwd (alpha = 0)
                        d_kx = torch.empty((N, E), dtype=torch.bfloat16, device=dev)
                        junk = torch.empty((N, E), dtype=torch.bfloat16, device=dev)
                        nv.adapt_dy(rec_v[2], lc.v_norm.weight.detach(), qc, q_stride, ds, torch.zeros(1, dtype=torch.float32, device=dev), axes, junk, d_kx,
                                    eps=lc.v_norm.eps)
                        del junk
                if lc.adapt_v:
                    av, gv, bvv = lc.v_alpha.detach().float(), f32[L + "v_norm.weight"], f32[L + "v_norm.bias"]
                    grads[L + "v_alpha"] = (dctx_l * (gv * syv + bvv - sxv)).sum().reshape(1)
                    grads[L + "v_norm.weight"] = av * (dctx_l * syv).sum(0)
                    grads[L + "v_norm.bias"] = av * dctx_l.sum(0)
                    d_vx = _adaptor_mlp_backward(vx2, lc.v_proj, lc.v_norm, lc.v_alpha, rec_v, pw, dctx_l, E, axes, L, "v", grads, ff_grad is not None)
                elif ff_grad is not None:
                    # plain value stream beside a key adaptor: d value_n = p_n dctx_w, the rank-1 branch of hicom_adapt_dy_fwd (alpha = 0)
                    d_vx = torch.empty((N, E), dtype=torch.bfloat16, device=dev)
                    junk = torch.empty((N, E), dtype=torch.bfloat16, device=dev)
                    nv.adapt_dy(rec_k[2], lc.k_norm.weight.detach(), dctx_l, E, pw, torch.zeros(1, dtype=torch.float32, device=dev), axes, junk, d_vx,
                                eps=lc.k_norm.eps)
                    del junk
                if want_fe and fe is not None:
                    d_fe = d_kx.to(fe.dtype).view(fe.shape)
                if ff_grad is not None:
                    # d frames_feature, local share: the value stream's input gradient (through the v adaptor's MLP and its skip path) and --
                    # without frames_embed -- the key stream's
                    dff = d_vx.float() + d_kx.float() if fe is None else d_vx
                    ff_grad["d_ff"] = dff.to(ff.dtype).view(ff.shape)
            else:
                dq_w = torch.empty((nw, E), dtype=torch.float32, device=dev)
                if want_fe and fe is not None:
                    d_fe = torch.empty_like(fe)
                dls = torch.empty((nw,), dtype=torch.float32, device=dev) if clip is not None else None
                d_ffl = None
                if ff_grad is not None:
                    # d frames_feature, local share: dv_n = p_n dctx_w (rank 1 per token, written by the same window backward); without
                    # frames_embed the keys are these rows too (projector.py:532) and their gradient is added in
                    d_ffl = ff_grad["d_ff"] = torch.empty_like(ff)      # (overlapping windows: the kernel accumulates per parity class, csrc/local_attn.hip)
                nv.local_attn_bwd(key, ff, axes, q, 0 if mode == "direct" else E, scale, bias, dctx_l, dq_w, d_fe, l2norm_key=l2k, dls=dls,
                                  dvalue=d_ffl, value_is_key=(d_ffl is not None and fe is None))
                if clip is not None:
                    grads["local_logit_scale"] = dls.sum().reshape(1)       # d s_i / d ls = s_i - lb
                    grads["local_logit_bias"] = torch.zeros(1, dtype=torch.float32, device=dev)   # a shift of a window's logits: softmax cancels it
            d_pool = None                                                   # gradient of the pooled per-window queries (d frames_feature only)
            if mode == "direct" and plain_q:
                if want_guide:
                    d_guide += through_guide_norm(dq_w.sum(0)).reshape(d_guide.shape)
            elif mode == "coarse" and plain_q:
                d_pool, dg = _coarse_backward(lc.guide_injector, "local_compressor.guide_injector.", pooled, guide_q, dq_w, grads)
                if want_guide:
                    d_guide += through_guide_norm(dg).reshape(d_guide.shape)
            elif query_params or (want_guide and mode is not None):
                d_inj = dq_w.sum(0, keepdim=True) if mode == "direct" else dq_w
                d_pool, dg = _query_chain_backward(lc, "local_compressor.", pooled, guide_q, d_inj, f32, grads, want_guide,
                                                   want_vis=ff_grad is not None and mode != "direct")
                if want_guide and dg is not None:
                    d_guide += through_guide_norm(dg).reshape(d_guide.shape)
            elif mode is None:
                d_pool = dq_w                                               # guide off, no query adaptor: the pooled rows ARE the queries (:544)
            if ff_grad is not None and d_pool is not None and mode != "direct":
                # through the trilinear pooling of frames_feature to the window grid (reference :539-540: F.interpolate(size = grid, 'trilinear')):
                # its adjoint on a [1, E, t', h', w'] cotangent (an ATen library call on a window-count sized tensor), added to the value-side share
                g5 = d_pool.reshape(*grid, E).permute(3, 0, 1, 2).unsqueeze(0).contiguous().float()
                dpf = torch.ops.aten.upsample_trilinear3d_backward(g5, list(grid), [1, E, T, H, W], False, None, None, None)
                ff_grad["d_ff"] = (ff_grad["d_ff"].float() + dpf[0].permute(1, 2, 3, 0)).to(ff.dtype)
    if gc is not None:
        att = gc.attn_layer
        nh, hd = att.num_heads, att.head_dim
        # injected queries: the guide itself, one row ("direct": 32 identical output rows), or the 32 learnable queries
        # (guide off, reference stage 1: IdentityMap injector, :586-587)
        q_in, n_rows = gc.injected_queries(guide)
        nq = q_in.shape[0]
        clip_g = proj.global_logit if "global" in stages else None        # (log scale, bias): clip-scale on the global stage (reference :184-191)
        if global_saved is not None and clip_g is None:
            ml, acc, scores = global_saved                                 # kept by the training forward (_GlobalStore)
        else:
            ml, acc, scores = gc.partial_context(ff, q_in, need_scores=True,   # HIP: forward logits + softmax state, rows q*nh + h
                                                 logit_scale=None if clip_g is None else clip_g[0])
        R = ml.shape[0]
        ctxg = (acc / ml[:, 1:2]).view(nq, nh, E)                          # per-(query, head) contexts
        q32 = q_in.float()
        A = "global_compressor.attn_layer."
        Wq, bq = f32[A + "q_proj.weight"], f32[A + "q_proj.bias"]
        Wk = f32[A + "k_proj.weight"]
        Wv, bv = f32[A + "v_proj.weight"], f32[A + "v_proj.bias"]
        Wo, bo = f32[A + "out_proj.weight"], f32[A + "out_proj.bias"]
        G0, gb0 = f32["global_compressor.readout.0.weight"], f32["global_compressor.readout.0.bias"]
        G2 = f32["global_compressor.readout.2.weight"]
        o = torch.einsum("hje,qhe->qhj", Wv.view(nh, hd, E), ctxg).reshape(nq, E) + bv    # ref :182,:215 after folding
        pre = o @ Wo.t() + bo + q32                                        # out_proj + residual with the injected query (:646)
        a1 = pre @ G0.t() + gb0
        hid = torch.nn.functional.gelu(a1)
        dtok = dout[n_local:n_local + n_rows].view(n_rows // nq, nq, -1).sum(0)   # direct: the 32 global rows are copies of one row
        P = "global_compressor."
        grads[P + "readout.2.weight"] = dtok.t() @ hid
        grads[P + "readout.2.bias"] = _sum0(dtok)
        da1 = _gelu_bwd(dtok @ G2, a1)
        grads[P + "readout.0.weight"] = da1.t() @ pre
        grads[P + "readout.0.bias"] = _sum0(da1)
        dpre = da1 @ G0
        grads[P + "attn_layer.out_proj.weight"] = dpre.t() @ o
        grads[P + "attn_layer.out_proj.bias"] = _sum0(dpre)
        do = dpre @ Wo                                                     # [nq, E]
        grads[P + "attn_layer.v_proj.bias"] = _sum0(do)
        grads[P + "attn_layer.v_proj.weight"] = torch.einsum("qhj,qhe->hje", do.view(nq, nh, hd), ctxg).reshape(E, E)
        dctx = torch.einsum("hje,qhj->qhe", Wv.view(nh, hd, E), do.view(nq, nh, hd)).reshape(R, E).contiguous()
        delta = (dctx * ctxg.reshape(R, E)).sum(1).contiguous()
        # ---- attention backward over the token stream (HIP) ------------------------------------------------
        rows_pad = (R + 15) // 16 * 16
        dhi = torch.empty((rows_pad, E), dtype=torch.bfloat16, device=dev)
        dlo = torch.empty_like(dhi)
        nv.split_bf16(dctx, rows_pad, dhi, dlo)
        N = T * H * W
        pe = None
        t0i = y0i = x0i = 0
        pos_b = None
        if gc.use_pos_emb:
            pe, cap = gc.pos_tables(T, H, W, dev)
            t0i, y0i, x0i = 0, cap, cap + H
            pos_b = torch.zeros((rows_pad, pe.shape[0]), dtype=torch.float32, device=dev)
            nv.linear(dctx, pe, None, pos_b, M=R)
        nparts = nv.global_stream_nparts(N, rows_pad)
        part = torch.empty((nparts, rows_pad, E), dtype=torch.float32, device=dev)
        in_kernel = (pe is not None and nv.global_stream_has_marg(N, E, rows_pad, H, W, nparts) and ff_grad is None   # (d frames_feature reads dS)
                     and clip_g is None)                                  # (so does clip-scale: the key norms scale it per token)
        if in_kernel:
            # many rows: the stream kernel leaves the t / y / x marginals of dS per token chunk -- the [rows, N] dS tensor is never written
            pm = torch.empty((nparts, rows_pad, nv.global_stream_marg_width(H, W)), dtype=torch.float32, device=dev)
            nv.global_stream_bwd(ff.view(N, E), N, dhi, dlo, pos_b, H, W, t0i, y0i, x0i, scores, ml, delta, None, part, R, part_marg=pm)
            ybase, xbase = 16, 16 + 16 * ((H + 15) // 16)
            mT = torch.zeros((R, T), dtype=torch.float32, device=dev)
            mT.index_add_(1, nv.marg_frame_index(N, H, W, nparts, dev).reshape(-1), pm[:, :R, :8].permute(1, 0, 2).reshape(R, -1))
            mY, mX = pm[:, :R, ybase:ybase + H].sum(0), pm[:, :R, xbase:xbase + W].sum(0)
        else:
            ds = torch.empty_like(scores)
            nv.global_stream_bwd(ff.view(N, E), N, dhi, dlo, pos_b, H if pe is not None else 1, W if pe is not None else N,
                                 t0i, y0i, x0i, scores, ml, delta, ds, part, R)
            if ff_grad is not None:
                # d frames_feature, global share: d x_n = sum_r dS[r, n] qt_r + p[r, n] dctx_r (scores and values are both x in the folded
                # form; the positional terms do not depend on x), added to the local stage's share
                qp_ = (q_in.float() @ Wq.t() + bq).view(nq, nh, hd)
                qt_ = (att.scale * torch.einsum("hje,qhj->qhe", Wk.view(nh, hd, E), qp_)).reshape(R, E).contiguous()
                have = ff_grad.get("d_ff") is not None
                if not have:
                    ff_grad["d_ff"] = torch.empty_like(ff)
                if R <= 16 and E <= 1280:
                    nv.global_dx(scores, ds, ml, qt_, dctx, N, ff_grad["d_ff"].view(N, E), accumulate=have)
                else:
                    # 32 distinct queries x heads (guide off / coarse / fine): a [N, 2 R] x [2 R, E] product -- plain library GEMMs
                    pr = torch.exp(scores[:R, :N] - ml[:, 0:1]) / ml[:, 1:2]
                    dxg = ds[:R, :N].t() @ qt_ + pr.t() @ dctx
                    cur = ff_grad["d_ff"].view(N, E)
                    cur.copy_(dxg + cur.float() if have else dxg)
            if pe is not None:
                dS = ds[:R, :N].view(R, T, H, W)
                mT, mY, mX = dS.sum((2, 3)), dS.sum((1, 3)), dS.sum((1, 2))
        if clip_g is None:
            dqt = part.sum(0)[:R]                                          # sum_n dS[r, n] x_n
            if pe is not None:
                dqt = dqt + mT @ pe[t0i:t0i + T] + mY @ pe[y0i:y0i + H] + mX @ pe[x0i:x0i + W]
            dqt = dqt.view(nq, nh, E)
            # ---- through the fold: qt[q,h] = scale W_k,h^T (W_q q + b_q)[q, h-slice]  (ref :180-181,:193-197) ---------
            scale = att.scale
            qp = (q32 @ Wq.t() + bq).view(nq, nh, hd)
            grads[P + "attn_layer.k_proj.weight"] = scale * torch.einsum("qhj,qhe->hje", qp, dqt).reshape(E, E)
            grads[P + "attn_layer.k_proj.bias"] = torch.zeros(E, device=dev)   # a per-row logit shift: softmax cancels it exactly
            dqp = scale * torch.einsum("hje,qhe->qhj", Wk.view(nh, hd, E), dqt).reshape(nq, E)
        else:
            dqp = _global_clip_backward(ff, pe, (t0i, y0i, x0i), scores, ds, R, nq, nh, hd, q32, Wq, bq, Wk, f32.get(A + "k_proj.bias"),
                                        float(clip_g[0]), grads, P)
        grads[P + "attn_layer.q_proj.weight"] = dqp.t() @ q32
        grads[P + "attn_layer.q_proj.bias"] = _sum0(dqp)
        if gc.use_guide in (None, "off"):
            grads[P + "query"] = dqp @ Wq + dpre                           # the learnable queries: through q_proj and the residual
        elif gc.use_guide == "fine" or gc.adapt_guide:                     # fine injection / adapted guide: the small-tensor graph
            d_in = dqp @ Wq + dpre
            direct_g = gc.use_guide == "direct"
            _, dg = _query_chain_backward(gc, P, None if direct_g else gc.query.detach().float(), guide,
                                          d_in.sum(0, keepdim=True) if direct_g else d_in, f32, grads, want_guide,
                                          vis_param=None if direct_g else "query")
            if want_guide and dg is not None:
                d_guide += dg.reshape(d_guide.shape)
        elif gc.use_guide == "coarse":                                     # injected = LN(query (1 + scale) + shift) (:369-372)
            dvis, dg = _coarse_backward(gc.guide_injector, P + "guide_injector.", gc.query.detach().float(), guide,
                                        dqp @ Wq + dpre, grads)
            grads[P + "query"] = dvis
            if want_guide:
                d_guide += dg
        elif want_guide:
            d_guide += _sum0(dqp @ Wq + dpre)                            # direct: the injected query IS the guide (:352-368)
        # (direct: global_compressor.query does not enter the forward, ref :352-368 uses only its shape: no gradient)
    return grads, d_nl, d_fe, d_guide


class _AnyresFn(torch.autograd.Function):
    """The anyres dict input of an image (reference projector.py:679-689): base image -> local stage; patch grid -> local stage with the
    anyres packing and the global stage.  Forward = the inference path (engine.run_anyres: one C call per segment), backward = compressor_backward
    per segment on that segment's rows of the cotangent, parameter gradients summed.  Parameters and image_newline only: the
    inputs' gradients (stage 3 trains on videos) are not built for dict inputs."""

    @staticmethod
    def forward(ctx, proj, ff_base, ff_patch, fe_base, fe_patch, guide, modal, nl, names, *params):
        with torch.no_grad():
            fdict = {"base": ff_base, "patch": ff_patch}
            edict = None if fe_patch is None else {"base": fe_base, "patch": fe_patch}
            if proj.use_executor and proj._executor_covers():         # (the inference path's calls: same kernels, same bits)
                from . import engine
                from .projector import _out_dtype
                out = engine.run_anyres(proj, fdict, edict, guide, modal, nl, _out_dtype(proj))
            else:
                out = proj.forward_stepwise(fdict, edict, guide, modal, nl)
        ctx.proj, ctx.modal, ctx.names = proj, modal, names
        ctx.save_for_backward(ff_base, ff_patch, fe_base, fe_patch, guide, nl)
        return out

    @staticmethod
    def backward(ctx, dout):
        global LAST_FP32_GRADS
        ff_base, ff_patch, fe_base, fe_patch, guide, nl = ctx.saved_tensors
        need = ctx.needs_input_grad            # (proj, ff_base, ff_patch, fe_base, fe_patch, guide, modal, nl, names, *params)
        proj = ctx.proj
        lc = proj.local_compressor
        # an input gradient that actually FLOWS is refused; one that does not (guide_embed of a recipe that never reads the guide, frames_embed
        # without a local stage) is None, as in the reference and in the dense path (a stage-3 script whose text embeddings
        # carry requires_grad failed on image batches of a guide-off recipe)
        uses_guide = any(c is not None and c.use_guide in ("direct", "coarse", "fine") for c in (lc, proj.global_compressor))
        if any(need[1:3]) or (any(need[3:5]) and lc is not None) or (need[5] and uses_guide):
            raise NotImplementedError("hicom_amd backward: input gradients (frames_feature / frames_embed / guide_embed) are not built "
                                      "for anyres dict inputs; detach them")
        with torch.no_grad():
            dout = dout.float()
            total, d_nl, row = {}, None, 0
            segs = []
            if lc is not None and ff_base is not None:
                T, H, W = 1, ff_base.shape[0], ff_base.shape[1]
                at, ay, ax = lc.tilings(T, H, W, ctx.modal)
                n = proj._layout((at.nwin, ay.nwin, ax.nwin), ctx.modal, nl is not None, False).n_rows
                segs.append((ff_base, fe_base, ("local",), False, row, row + n))
                row += n
            segs.append((ff_patch, fe_patch, ("local", "global"), True, row, dout.shape[0]))
            for ff, fe, stages, anyres, r0, r1 in segs:
                g, dn, _, _ = compressor_backward(proj, ff.unsqueeze(0).contiguous(), None if fe is None else fe.unsqueeze(0).contiguous(), guide,
                                                  ctx.modal, nl, dout[r0:r1], stages=stages, is_anyres=anyres)
                for k, v in g.items():
                    total[k] = v if k not in total else total[k] + v
                if dn is not None:
                    d_nl = dn if d_nl is None else d_nl + dn
        LAST_FP32_GRADS = dict(total)
        plist = dict(proj.named_parameters())
        out = [total[name].to(plist[name].dtype).view(plist[name].shape) if (need[9 + k] and total.get(name) is not None) else None
               for k, name in enumerate(ctx.names)]
        return (None, None, None, None, None, None, None, (d_nl.to(nl.dtype) if (nl is not None and need[7] and d_nl is not None) else None), None, *out)


def forward_with_grad(proj, frames_feature, frames_embed, guide_embed, modal, image_newline):
    from .projector import _require_bf16_cuda
    some = frames_feature["patch"] if isinstance(frames_feature, dict) else frames_feature
    _require_bf16_cuda("frames_feature", some)
    if not _supported(proj):
        raise NotImplementedError("hicom_amd: the backward pass covers every injection mode and adaptor and clip-scale on the local stage "
                                  "(not together with k / v adaptors); not clip-scale on the global stage, not a text2qk projection; run "
                                  "other configurations under torch.no_grad() / inference_mode() -- forward() never returns a silently "
                                  "detached tensor")
    names, params = zip(*[(n, p) for n, p in proj.named_parameters()])
    if isinstance(frames_feature, dict):
        c = lambda t: None if t is None else t.contiguous()
        fe = frames_embed if frames_embed is not None else {"base": None, "patch": None}
        return _AnyresFn.apply(proj, c(frames_feature["base"]), c(frames_feature["patch"]), c(fe["base"]), c(fe["patch"]), c(guide_embed), modal,
                               c(image_newline), names, *params)
    ff = frames_feature.contiguous()
    fe = frames_embed.contiguous() if frames_embed is not None else None
    return _CompressorFn.apply(proj, ff, fe, guide_embed.contiguous() if guide_embed is not None else None, modal,
                               image_newline.contiguous() if image_newline is not None else None, names, *params)
