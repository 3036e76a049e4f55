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
bf16; with k / v adaptors the window backward and the adaptor chain run as fp32 tensor algebra + library GEMMs).  The query chain
(adapt_q, adapt_guide, coarse / fine injection) acts on small tensors only and is differentiated as a torch graph
(_query_chain_backward).  `frames_feature` comes from the frozen tower body: asking for its gradient raises instead of returning
None silently, and so do input gradients of the guide-off recipe, clip-scale projectors and text2qk projections.
"""
from __future__ import annotations

import math

import torch

from . import native as nv


LAST_FP32_GRADS = None      # test hook: the fp32 gradients of the last backward (before the cast to the parameter dtype)


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def _supported(proj) -> bool:
    """Every injection mode and adaptor; not: text2qk projections (text width != query width), clip-scale."""
    from .projector import _plain_injector
    lc, gc = proj.local_compressor, proj.global_compressor
    for c in (lc, gc):
        if c is None:
            continue
        if c.use_guide not in ("direct", None, "off", "coarse", "fine") or not _plain_injector(c.guide_injector):
            return False
    return proj.local_logit is None and proj.global_logit is None


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


def _mm(a, b):
    """a @ b on the vendor GEMM with bf16 operands, fp32 accumulation and an fp32 result (the token-stream GEMMs of the adaptor
    backward: 124 GFLOP each at 64 frames -- 0.15 ms like this against 1.2 ms in fp32)."""
    try:
        return torch.mm(a.to(torch.bfloat16), b.to(torch.bfloat16), out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.mm(a.float(), b.float())


class _AdaptorTape:
    """(1 - a) x + a LN(MLP(x)) over all tokens (adapt_k / adapt_v, reference projector.py:533-534) recomputed with its
    intermediates, and its backward: library GEMMs over [N, D] (dW = dY^T X, dX = dY W; bf16 operands, fp32 accumulate / result),
    LayerNorm / GELU algebra in fp32."""

    def __init__(self, x, mlp, norm, alpha):
        self.mlp, self.norm, self.alpha = mlp, norm, alpha.detach().float()
        self.x = x.reshape(-1, x.shape[-1]).float()
        self.W1, self.W2 = mlp[0].weight.detach(), mlp[2].weight.detach()
        D = self.x.shape[-1]
        if x.dtype == torch.bfloat16 and D % 64 == 0 and self.W1.shape[0] % 64 == 0:
            # recompute on the forward's own kernels (hicom_dense16_gemm_fwd: exact bf16 products / fp16 operands, fp32 accumulation,
            # fp16 results = 11 significand bits, like the forward's adapted stream)
            from . import injector as inj
            x2 = x.reshape(-1, D).contiguous()
            h1 = torch.empty((x2.shape[0], self.W1.shape[0]), dtype=torch.float16, device=x.device)
            nv.dense16_gemm(x2, self.W1, mlp[0].bias.detach(), act=nv.ACT_NONE, out_f16=h1)
            self.h1 = h1.float()
            self.a = torch.nn.functional.gelu(self.h1)
            h2 = torch.empty((x2.shape[0], self.W2.shape[0]), dtype=torch.float16, device=x.device)
            nv.dense16_gemm(self.a.half(), inj._f16_weight(mlp[2]), mlp[2].bias.detach(), out_f16=h2)
            h2 = h2.float()
        else:
            self.h1 = torch.addmm(mlp[0].bias.detach().float(), self.x, self.W1.float().t())
            self.a = torch.nn.functional.gelu(self.h1)
            h2 = torch.addmm(mlp[2].bias.detach().float(), self.a, self.W2.float().t())
        self.nhat, self.rstd = _ln_stats(h2, norm.eps)
        self.gamma = norm.weight.detach().float()
        self.n = self.nhat * self.gamma + norm.bias.detach().float()
        self.out = (1.0 - self.alpha) * self.x + self.alpha * self.n

    def backward(self, d_out, prefix, which, grads, want_x):
        """prefix 'local_compressor.', which 'k' | 'v'; returns d x (fp32 [N, D]) when want_x."""
        grads[f"{prefix}{which}_alpha"] = (d_out * (self.n - self.x)).sum().reshape(1)
        dn = self.alpha * d_out
        grads[f"{prefix}{which}_norm.weight"] = (dn * self.nhat).sum(0)
        grads[f"{prefix}{which}_norm.bias"] = dn.sum(0)
        dh2 = _ln_backward(dn * self.gamma, self.nhat, self.rstd)
        grads[f"{prefix}{which}_proj.2.weight"] = _mm(dh2.t(), self.a)
        grads[f"{prefix}{which}_proj.2.bias"] = dh2.sum(0)
        dh1 = _mm(dh2, self.W2) * _gelu_grad(self.h1)
        grads[f"{prefix}{which}_proj.0.weight"] = _mm(dh1.t(), self.x)
        grads[f"{prefix}{which}_proj.0.bias"] = dh1.sum(0)
        return (1.0 - self.alpha) * d_out + _mm(dh1, self.W1) if want_x else None


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
    dh1 = (W2.t() @ dcs) * _gelu_grad(h1)
    grads[prefix + "coarse_proj.0.weight"] = torch.outer(dh1, g)
    grads[prefix + "coarse_proj.0.bias"] = dh1
    return dz * (1.0 + sc), W1.t() @ dh1


def _query_chain_backward(stage, prefix, vis, guide, d_inj, f32, grads, want_guide, vis_param=None):
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
            v = vis.detach().clone().requires_grad_(vis_param is not None)
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
        if v is not None and vis_param is not None:
            extra.append(v)
        if not wrt and not extra:
            return None, None
        got = torch.autograd.grad(q, wrt + extra, grad_outputs=d_inj.reshape(q.shape), allow_unused=True)
    for n, gr in zip(names, got[:len(names)]):
        if gr is not None:
            grads[prefix + n] = gr
    rest = list(got[len(names):])
    d_guide = rest.pop(0) if (g is not None and want_guide) else None
    d_vis = rest.pop(0) if (v is not None and vis_param is not None) else None
    if d_vis is not None:
        grads[prefix + vis_param] = d_vis
    return d_vis, d_guide


def _to_windows(x, at, ay, ax):
    """[T, H, W, D] -> [Nw, kt ks ks, D] in the reference's window / in-window order (projector.py:473-499), exact partition."""
    D = x.shape[-1]
    return (x.reshape(at.nwin, at.k, ay.nwin, ay.k, ax.nwin, ax.k, D).permute(0, 2, 4, 1, 3, 5, 6)
            .reshape(at.nwin * ay.nwin * ax.nwin, at.k * ay.k * ax.k, D))


def _from_windows(xw, at, ay, ax):
    D = xw.shape[-1]
    return (xw.reshape(at.nwin, ay.nwin, ax.nwin, at.k, ay.k, ax.k, D).permute(0, 3, 1, 4, 2, 5, 6)
            .reshape(at.n * ay.n * ax.n, D))


def _window_attention_backward(K, V, q, scale, dctx, tilings, want_k, want_v):
    """Autograd through reference projector.py:550-553 on fp32 streams K, V [T,H,W,D] (adaptor outputs): q [D] shared or
    [Nw, D]; returns (dq [Nw, D], dK [N, D] | None, dV [N, D] | None)."""
    at, ay, ax = tilings
    Kw, Vw = _to_windows(K, at, ay, ax), _to_windows(V, at, ay, ax)
    qw = q.reshape(1, -1).expand(Kw.shape[0], -1) if q.ndim == 1 or q.shape[0] == 1 else q
    p = torch.softmax(torch.einsum("wnd,wd->wn", Kw, qw) * scale, dim=-1)
    dP = torch.einsum("wnd,wd->wn", Vw, dctx)
    ds = p * (dP - (p * dP).sum(-1, keepdim=True)) * scale
    dq = torch.einsum("wn,wnd->wd", ds, Kw)
    dK = _from_windows(ds.unsqueeze(-1) * qw.unsqueeze(1), at, ay, ax) if want_k else None
    dV = _from_windows(p.unsqueeze(-1) * dctx.unsqueeze(1), at, ay, ax) if want_v else None
    return dq, dK, dV


class _CompressorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, ff, fe, guide, modal, nl, names, *params):
        from . import engine
        from .projector import _out_dtype
        with torch.no_grad():
            if proj.use_executor and proj._executor_covers():          # (plain recipes and the k / v adaptors: one C call)
                out = engine.run_dense(proj, ff, fe, guide, modal, nl, _out_dtype(proj))
            else:                                                      # query-side adaptors / coarse / fine injection: operator by operator
                out = proj.forward_stepwise(ff, fe, guide, modal, nl)
        ctx.proj, ctx.modal, ctx.names = proj, modal, names
        ctx.save_for_backward(ff, fe, guide, nl)
        return out

    @staticmethod
    def backward(ctx, dout):
        ff, fe, guide, nl = ctx.saved_tensors
        need = ctx.needs_input_grad            # (proj, ff, fe, guide, modal, nl, names, *params)
        if need[1]:
            raise NotImplementedError("hicom_amd backward: the gradient w.r.t. frames_feature is not built (the tower body is "
                                      "frozen in every stage of the reference's script, train.py:703); detach it")
        proj = ctx.proj
        want = tuple(bool(need[7 + k]) for k in range(len(ctx.names)))
        args = (proj, ff, fe, guide, ctx.modal, nl, ctx.names, want, bool(need[2]), bool(need[3]), bool(nl is not None and need[5]))
        if getattr(proj, "graph_backward", False) and nl is None and _is_plain(proj):
            res = _graphed_backward(dout, *args)
        else:
            with torch.no_grad():
                res = _backward_outputs(dout, *args)
        flats, d_fe, d_guide, d_nl = res
        plist = dict(proj.named_parameters())
        out = [None] * len(ctx.names)
        for dt, (flat, group) in flats.items():                    # views of ONE buffer per parameter dtype
            o = 0
            for k, name in group:
                n = plist[name].numel()
                out[k] = flat[o:o + n].view(plist[name].shape)
                o += n
        return (None, None, d_fe, d_guide, None, d_nl, None, *out)


def _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl):
    """({dtype: (flat gradient buffer, [(argument index, parameter name)])}, d frames_embed, d guide_embed, d image_newline) in
    the dtypes autograd hands on.  The parameter gradients leave as views of one buffer cast once (one concatenation + one cast
    instead of a cast per tensor)."""
    global LAST_FP32_GRADS
    grads, d_nl, d_fe, d_guide = compressor_backward(proj, ff, fe, guide, modal, nl, dout, want_fe=want_fe, want_guide=want_guide)
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


_MAX_BWD_GRAPHS = 4


def _graphed_backward(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl):
    """Opt-in (`proj.graph_backward = True`): the backward of a plain recipe as a captured hipGraph.  The backward is ~130 small
    launches behind 1.4 ms of Python at the benchmark shape; its shapes are static, so the second backward with the same input
    BUFFERS (a training loop's caching allocator hands the same blocks back step after step) is captured and later ones are one
    graph launch.  Keyed by the addresses the captured kernels read (frames, guide, parameters) and by what is asked for; the
    upstream gradient is copied into a static buffer, the results are cloned out of the graph's pool (gradient accumulation adds
    into .grad in place: handing out the static buffers would alias them).  Falls back to the eager backward on any capture error."""
    from . import engine
    cache = proj.__dict__.setdefault("_bwd_graphs", {})
    key = (ff.data_ptr(), tuple(ff.shape), None if fe is None else fe.data_ptr(), None if guide is None else guide.data_ptr(), modal,
           tuple(dout.shape), dout.dtype, want, want_fe, want_guide, torch.cuda.current_stream(ff.device).cuda_stream)
    # what else the captured kernels read by ADDRESS: every parameter's storage and the cached device tables (pe / kpe / planes:
    # `_cache_gen` moves when one is reallocated -- T above the cached cap, a cleared cache, a device move).  A graph whose
    # signature moved is dropped, never replayed over freed or reused memory.
    sig = engine.plan_sig(proj)
    ent = cache.get(key)
    if ent is not None and ent["sig"] != sig:
        cache.pop(key)
        ent = None
    if ent is None:                                                # first sight: eager (also the warm-up a capture needs)
        if len(cache) >= _MAX_BWD_GRAPHS:
            cache.pop(next(iter(cache)))
        gc = proj.global_compressor
        # the entry keeps the buffers the graph will read alive: inputs and the cached tables (a recycled address must not pass
        # for the tensor the graph was captured over)
        cache[key] = {"seen": 1, "sig": sig, "refs": (ff, fe, guide, None if gc is None else dict(gc._pe_cache))}
        with torch.no_grad():
            return _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl)
    if "graph" not in ent:
        if ent.get("failed"):
            with torch.no_grad():
                return _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl)
        try:
            static_dout = dout.detach().clone()
            torch.cuda.current_stream(ff.device).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                outs = _backward_outputs(static_dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl)
            if engine.plan_sig(proj) != sig:                       # (the eager pass inside the capture reallocated a table)
                raise RuntimeError("cached device tables moved during capture")
            ent.update(graph=g, dout=static_dout, outs=outs)
        except Exception as e:  # noqa: BLE001
            ent["failed"] = repr(e)
            with torch.no_grad():
                return _backward_outputs(dout, proj, ff, fe, guide, modal, nl, names, want, want_fe, want_guide, want_nl)
    ent["dout"].copy_(dout)
    ent["graph"].replay()
    flats, d_fe, d_guide, d_nl = ent["outs"]
    return ({dt: (flat.clone(), group) for dt, (flat, group) in flats.items()}, None if d_fe is None else d_fe.clone(),
            None if d_guide is None else d_guide.clone(), None)


def compressor_backward(proj, ff, fe, guide, modal, nl, dout, want_fe=False, want_guide=False):
    """(fp32 gradients {parameter name: tensor} of sum(out * dout), d image_newline, d frames_embed (bf16) or None,
    d guide_embed (fp32) or None).  Restates autograd through reference projector.py:524-559 (local), :634-646 + :166-228
    (global) and mm_utils.py:92-140 (packing).  The input gradients exist for the direct recipe only."""
    lc, gc = proj.local_compressor, proj.global_compressor
    dev = ff.device
    d_fe = d_guide = None
    if want_fe or want_guide:
        if not any(c is not None and c.use_guide in ("direct", "coarse", "fine") for c in (lc, gc)):
            raise NotImplementedError("hicom_amd backward: gradients w.r.t. frames_embed / guide_embed need a recipe that uses them "
                                      "(use_guide = direct / coarse / fine; stage 3 of the reference's script)")
        if want_guide:
            d_guide = torch.zeros(guide.shape, dtype=torch.float32, device=dev)
        if want_fe and lc is None:
            want_fe = False            # without a local stage frames_embed does not enter the forward: no gradient (None), as in the reference
    dout = dout.float()
    f32 = _f32_params(proj)
    T, H, W, E = ff.shape
    grads = {}
    d_nl = None
    n_local = 0
    if lc is not None:
        at, ay, ax = lc.tilings(T, H, W, modal)
        grid = (at.nwin, ay.nwin, ax.nwin)
        lay = proj._layout(grid, modal, nl is not None, False)
        nw = grid[0] * grid[1] * grid[2]
        n_local = lay.n_rows
        idx = torch.arange(nw, device=dev)
        rows = idx + (idx // lay.nl_group if lay.nl_group else 0)          # packing row map of the readout store
        dY = dout[rows]
        if lay.newline_rows:
            d_nl = dout[torch.tensor(lay.newline_rows, device=dev)].sum(0)
        ctx_l, _ = lc.window_context(ff, fe, guide, modal, None, None)     # HIP: [Nw, E] fp32 window contexts
        W0, b0 = f32["local_compressor.readout.0.weight"], f32["local_compressor.readout.0.bias"]
        W2 = f32["local_compressor.readout.2.weight"]
        pre = torch.addmm(b0, ctx_l, W0.t())
        h = torch.nn.functional.gelu(pre)
        grads["local_compressor.readout.2.weight"] = dY.t() @ h
        grads["local_compressor.readout.2.bias"] = dY.sum(0)
        dpre = (dY @ W2) * _gelu_grad(pre)
        grads["local_compressor.readout.0.weight"] = dpre.t() @ ctx_l
        grads["local_compressor.readout.0.bias"] = dpre.sum(0)
        mode = lc.use_guide if lc.use_guide not in (None, "off") else None
        adapt = lc.adapt_k or lc.adapt_v
        plain_q = not (lc.adapt_q or lc.adapt_guide)                       # plain direct / coarse / off: the hand-written paths below
        query_params = mode in ("coarse", "fine") or lc.adapt_q or lc.adapt_guide
        if want_fe or adapt or query_params or (want_guide and mode is not None):
            # ---- attention backward of the windows: dq per window, d key stream, d value stream -------------------------
            from . import injector as inj
            exact = all(a.nwin * a.k == a.n for a in (at, ay, ax))
            if (want_fe or adapt) and not exact:
                raise NotImplementedError("hicom_amd backward: d frames_embed / the adaptor gradients need an exact window "
                                          f"partition (T, H, W = {T}, {H}, {W} against kernel {at.k}, {ay.k}, {ax.k})")
            axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
            dctx_l = (dpre @ W0).contiguous()
            scale = 1.0 / math.sqrt(lc.qk_dim)
            key = fe if fe is not None else ff
            pooled = None
            if mode == "direct":
                q, _ = inj.inject(lc.guide_injector, "direct", None, guide.contiguous())       # the (adapted) guide, one row
                q = q.reshape(-1).contiguous()
            else:
                pooled = torch.empty((*grid, E), dtype=torch.float32, device=dev)
                nv.trilinear_pool(ff, pooled)                              # HIP: the per-window query (ref :539-540)
                pooled = pooled.view(nw, E)
                q = pooled
                if lc.adapt_q:
                    q = inj.adapt_query(q, lc.q_proj, lc.q_norm, lc.q_alpha)                    # HIP (ref :541)
                if mode in ("coarse", "fine"):
                    q, _ = inj.inject(lc.guide_injector, mode, q.reshape(nw, E), guide.contiguous())   # HIP (ref :369-397)
            if adapt:
                tape_k = _AdaptorTape(key, lc.k_proj, lc.k_norm, lc.k_alpha) if lc.adapt_k else None
                tape_v = _AdaptorTape(ff, lc.v_proj, lc.v_norm, lc.v_alpha) if lc.adapt_v else None
                K = (tape_k.out if tape_k else key.float()).view(T, H, W, E)
                V = (tape_v.out if tape_v else ff.float()).view(T, H, W, E)
                dq_w, dK, dV = _window_attention_backward(K, V, q.float(), scale, dctx_l, (at, ay, ax),
                                                          want_fe or lc.adapt_k, lc.adapt_v)
                if tape_v:
                    tape_v.backward(dV, "local_compressor.", "v", grads, False)
                if tape_k:
                    dK = tape_k.backward(dK, "local_compressor.", "k", grads, want_fe and fe is not None)
                if want_fe and fe is not None:
                    d_fe = dK.to(fe.dtype).view(fe.shape)
            else:
                dq_w = torch.empty((nw, E), dtype=torch.float32, device=dev)
                if want_fe and fe is not None:
                    d_fe = torch.empty_like(fe)
                nv.local_attn_bwd(key, ff, axes, q, 0 if mode == "direct" else E, scale, 0.0, dctx_l, dq_w, d_fe)
            if mode == "direct" and plain_q:
                if want_guide:
                    d_guide += dq_w.sum(0)
            elif mode == "coarse" and plain_q:
                _, dg = _coarse_backward(lc.guide_injector, "local_compressor.guide_injector.", pooled, guide, dq_w, grads)
                if want_guide:
                    d_guide += dg
            elif query_params or (want_guide and mode is not None):
                d_inj = dq_w.sum(0, keepdim=True) if mode == "direct" else dq_w
                _, dg = _query_chain_backward(lc, "local_compressor.", pooled, guide, d_inj, f32, grads, want_guide)
                if want_guide and dg is not None:
                    d_guide += dg.reshape(d_guide.shape)
    if gc is not None:
        att = gc.attn_layer
        nh, hd = att.num_heads, att.head_dim
        # injected queries: the guide itself, one row ("direct": 32 identical output rows), or the 32 learnable queries
        # (guide off, reference stage 1: IdentityMap injector, :586-587)
        q_in, n_rows = gc.injected_queries(guide)
        nq = q_in.shape[0]
        ml, acc, scores = gc.partial_context(ff, q_in)                     # HIP: forward logits + softmax state, rows q*nh + h
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
        grads[P + "readout.2.bias"] = dtok.sum(0)
        da1 = (dtok @ G2) * _gelu_grad(a1)
        grads[P + "readout.0.weight"] = da1.t() @ pre
        grads[P + "readout.0.bias"] = da1.sum(0)
        dpre = da1 @ G0
        grads[P + "attn_layer.out_proj.weight"] = dpre.t() @ o
        grads[P + "attn_layer.out_proj.bias"] = dpre.sum(0)
        do = dpre @ Wo                                                     # [nq, E]
        grads[P + "attn_layer.v_proj.bias"] = do.sum(0)
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
        ds = torch.empty_like(scores)
        nparts = nv.global_stream_nparts(N, 16)
        part = torch.empty((nparts, rows_pad, E), dtype=torch.float32, device=dev)
        nv.global_stream_bwd(ff.view(N, E), N, dhi, dlo, pos_b, H if pe is not None else 1, W if pe is not None else N,
                             t0i, y0i, x0i, scores, ml, delta, ds, part, R)
        dqt = part.sum(0)[:R]                                              # sum_n dS[r, n] x_n
        if pe is not None:
            dS = ds[:R, :N].view(R, T, H, W)
            dqt = dqt + dS.sum((2, 3)) @ pe[t0i:t0i + T] + dS.sum((1, 3)) @ pe[y0i:y0i + H] + dS.sum((1, 2)) @ pe[x0i:x0i + W]
        dqt = dqt.view(nq, nh, E)
        # ---- through the fold: qt[q,h] = scale W_k,h^T (W_q q + b_q)[q, h-slice]  (ref :180-181,:193-197) ---------
        scale = att.scale
        qp = (q32 @ Wq.t() + bq).view(nq, nh, hd)
        grads[P + "attn_layer.k_proj.weight"] = scale * torch.einsum("qhj,qhe->hje", qp, dqt).reshape(E, E)
        grads[P + "attn_layer.k_proj.bias"] = torch.zeros(E, device=dev)   # a per-row logit shift: softmax cancels it exactly
        dqp = scale * torch.einsum("hje,qhe->qhj", Wk.view(nh, hd, E), dqt).reshape(nq, E)
        grads[P + "attn_layer.q_proj.weight"] = dqp.t() @ q32
        grads[P + "attn_layer.q_proj.bias"] = dqp.sum(0)
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
            d_guide += (dqp @ Wq + dpre).sum(0)                            # direct: the injected query IS the guide (:352-368)
        # (direct: global_compressor.query does not enter the forward, ref :352-368 uses only its shape: no gradient)
    return grads, d_nl, d_fe, d_guide


def forward_with_grad(proj, frames_feature, frames_embed, guide_embed, modal, image_newline):
    from .projector import _require_bf16_cuda
    some = frames_feature["patch"] if isinstance(frames_feature, dict) else frames_feature
    _require_bf16_cuda("frames_feature", some)
    if isinstance(frames_feature, dict) or not _supported(proj):
        raise NotImplementedError("hicom_amd: the backward pass covers every injection mode and adaptor on dense inputs, without "
                                  "clip-scale and without a text2qk projection; run other configurations under torch.no_grad() / "
                                  "inference_mode() -- forward() never returns a silently detached tensor")
    names, params = zip(*[(n, p) for n, p in proj.named_parameters()])
    ff = frames_feature.contiguous()
    fe = frames_embed.contiguous() if frames_embed is not None else None
    return _CompressorFn.apply(proj, ff, fe, guide_embed.contiguous() if guide_embed is not None else None, modal,
                               image_newline.contiguous() if image_newline is not None else None, names, *params)
