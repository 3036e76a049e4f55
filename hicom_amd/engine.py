"""Host side of the native executor (hicom_compressor_fwd): fills `hicom_compressor_args` from a
HIComProjector and its inputs, and owns the workspaces, the side stream and the fork/join events.
All compute happens in libhicom_hip.so; torch is used for device memory, streams and events only.

Plans (filled argument blocks + their workspace) are keyed by what determines the launch sequence --
shapes, recipe, dtype, caller stream, the state of the weights -- and NOT by the identity of the input
buffers: the pointers of frames_feature / frames_embed / guide / newline / out are patched into the cached
block on every call.  A serving loop that hands over a fresh `split` view per video (reference
hicom_arch.py:162-164) therefore pays one ctypes call per forward, not a plan build.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch

from . import native as nv
from .events import device_event


class _DeviceResources:
    def __init__(self, device):
        # high priority: the global chain is a string of small kernels that must not queue behind the
        # thousands of workgroups of the local streaming kernel
        self.side = torch.cuda.Stream(device=device, priority=-1)
        # (stream-to-stream ordering on one device: events without the system-scope fence, events.py)
        self.ev_fork = device_event()
        self.ev_join = device_event()
        self.ev_merge = device_event()
        # hipEventRecord needs created events: torch creates them lazily on first record
        cur = torch.cuda.current_stream(device)
        self.ev_fork.record(cur)
        self.ev_join.record(cur)
        self.ev_merge.record(cur)
        # injected local queries (coarse / fine / adapt_q recipes) are made on a stream of their own, beside the global stage's stream kernel
        self.inj = torch.cuda.Stream(device=device, priority=-1)
        self.ev_lq = device_event()
        self.ev_lq.record(cur)
        self.done = [torch.cuda.Event() for _ in range(16)]     # per-call completion events of deferred forwards
        for ev in self.done:
            ev.record(cur)                                      # (created: the executor records them from C)
        self.n_done = 0


_RES: Dict[Tuple[int, int], _DeviceResources] = {}


def _resources(device) -> _DeviceResources:
    """Side stream + fork/join events, one set per (device, caller stream): forwards issued on two
    different caller streams must not share events or the side queue."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream)
    if key not in _RES:
        _RES[key] = _DeviceResources(device)
    return _RES[key]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _w(lin) -> Tuple[int, Optional[int]]:
    w = lin.weight
    if not w.is_cuda or w.dtype != torch.bfloat16:
        raise NotImplementedError("hicom_amd: projector weights must be bfloat16 on the GPU "
                                  "(cast the projector with .to(torch.bfloat16).cuda())")
    return w.data_ptr(), (lin.bias.data_ptr() if lin.bias is not None else None)


def _fill_injector(j, inj, mode, guide, visual, keep):
    """hicom_injector for a plain GuideInjector (reference projector.py:369-397); the weights' addresses are plan state (plan_sig)."""
    from .projector import _require_bf16_cuda
    _require_bf16_cuda("guide_embed", guide)
    E = guide.shape[-1]
    if mode == "coarse":
        if guide.ndim != 1:
            raise ValueError("coarse guide injection takes a [D] guide embedding")
        j.mode, j.guide_rows = 1, 1
        j.c_w0, j.c_b0 = _w(inj.coarse_proj[0])
        j.c_w2, j.c_b2 = _w(inj.coarse_proj[2])
        j.c_hidden = inj.coarse_proj[0].out_features
        norm = inj.coarse_norm
    else:
        if guide.ndim != 2:
            raise ValueError("fine guide injection takes an [L, D] guide embedding")
        if guide.shape[0] > 64:
            raise NotImplementedError("fine guide injection: at most 64 text tokens (hicom_small_mha_fwd)")
        att = inj.fine_proj
        j.mode, j.guide_rows = 2, guide.shape[0]
        j.wq, j.bq = _w(att.q_proj)
        j.wk, j.bk = _w(att.k_proj)
        j.wv, j.bv = _w(att.v_proj)
        j.wo, j.bo = _w(att.out_proj)
        j.nheads = att.num_heads
        norm = inj.fine_norm
    _require_bf16_cuda("injector norm", norm.weight)
    j.ln_w, j.ln_b, j.eps = norm.weight.data_ptr(), norm.bias.data_ptr(), 1e-6      # (the eps injector.inject() runs the row LayerNorm with)
    j.guide = guide.data_ptr()
    j.visual = None if visual is None else visual.data_ptr()
    keep.append(guide)


def build_args(proj, ff, fe, guide_embed, modal, image_newline, out, layout, *, t_offset=0,
               phases=nv.PHASE_STREAM | nv.PHASE_FINISH, local_out=None, state_out=None,
               state_sets=None, state_set_stride=0, nsets=0, global_row0=None, stages=("local", "global"), local_row0=0) -> nv.CompressorArgs:
    """`stages` / `local_row0`: one SEGMENT of an anyres dict input (reference projector.py:679-689: the base image takes the local stage
    only; the patch grid both, its rows behind the base image's) written into rows of a shared output."""
    from .projector import _require_bf16_cuda
    lc = proj.local_compressor if "local" in stages else None
    gc = proj.global_compressor if "global" in stages else None
    a = nv.CompressorArgs()
    _require_bf16_cuda("frames_feature", ff)
    T, H, W, E = ff.shape
    a.ff, a.T, a.H, a.W, a.E = ff.data_ptr(), T, H, W, E
    a.fe = None
    a.phases = phases
    a.has_local, a.has_global = int(lc is not None), int(gc is not None)
    a.hidden = (lc or gc).readout[2].out_features
    keep = [ff]
    a._guide_ptr_fields = ()            # argument fields that alias the caller's guide tensor (patched per call)
    ext_l = ext_g = None
    gptr = None if guide_embed is None else guide_embed.data_ptr()
    if lc is not None:
        lc._check_native()
        if fe is not None:
            _require_bf16_cuda("frames_embed", fe)
            if fe.shape != ff.shape:
                raise ValueError("frames_embed must have the shape of frames_feature")
            a.fe = fe.data_ptr()
            keep.append(fe)
        at, ay, ax = lc.tilings(T, H, W, modal)
        a.at, a.ay, a.ax = (nv.Axis(t.n, t.k, t.nwin, t.nfull) for t in (at, ay, ax))
        ls = proj.local_logit
        a.l2norm = 0
        if ls is not None:                                            # ref :527-529, :549
            a.l_scale, a.l_bias = math.exp(ls[0]), ls[1]
            if fe is not None:
                a.l2norm = 1 | (2 if lc.use_guide == "direct" else 0)
        else:
            a.l_scale, a.l_bias = 1.0 / math.sqrt(lc.qk_dim), 0.0      # ref :551
        if lc.inject_in_call:
            # coarse / fine injection into the pooled queries, run by the executor itself (hicom_compressor_args.inj_l)
            if ls is not None:
                raise NotImplementedError("the one-call executor takes injected local queries without clip-scale")
            a.lq = None
            _fill_injector(a.inj_l, lc.guide_injector, lc.use_guide, guide_embed, None, keep)
            a._guide_ptr_fields += ("inj_l.guide",)
        elif lc.external_queries:
            # coarse / fine injection, adapt_q, an adapted guide (ref :539-542): pooling + adaptor + injector run in front of every call
            # (`_queries` below) and leave f32 rows in buffers this plan owns
            if ls is not None:
                raise NotImplementedError("the one-call executor takes injected / adapted local queries without clip-scale")
            nw = at.nwin * ay.nwin * ax.nwin
            rows = 1 if lc.use_guide == "direct" else nw
            ext_l = (torch.empty((rows, E), dtype=torch.float32, device=ff.device),
                     torch.empty((at.nwin, ay.nwin, ax.nwin, E), dtype=torch.float32, device=ff.device))
            a.lq, a.lq_dt, a.lq_stride = ext_l[0].data_ptr(), nv.DT_F32, 0 if rows == 1 else E
        elif lc.use_guide == "direct":
            g = guide_embed
            _require_bf16_cuda("guide_embed", g)
            if g.ndim != 1 or g.shape[0] != E:
                raise ValueError("direct guide injection takes a [D] guide embedding")
            a.lq, a.lq_dt, a.lq_stride = g.data_ptr(), nv.DT_BF16, 0
            a._guide_ptr_fields += ("lq",)
            keep.append(g)
        else:
            a.lq = None                                               # pooled per-window query, made natively
        a.lw0, a.lb0 = _w(lc.readout[0])
        a.lw2, a.lb2 = _w(lc.readout[2])
        w0_16, w2_16 = lc.readout_f16()                               # fp16 copies, cached per weight version
        a.lw0_f16, a.lw2_f16 = w0_16.data_ptr(), w2_16.data_ptr()
        keep += [w0_16, w2_16]
        if lc.adapt_k or lc.adapt_v:                                  # k / v adaptors (ref :431-457, :533-534)
            from . import injector as inj
            if ls is not None:
                raise NotImplementedError("the one-call executor takes k / v adaptors without clip-scale")
            for on, dst, mlp, norm, alpha in ((lc.adapt_k, a.ak, lc.k_proj, lc.k_norm, lc.k_alpha), (lc.adapt_v, a.av, lc.v_proj, lc.v_norm, lc.v_alpha)):
                if not on:
                    continue
                dst.w0, dst.b0 = _w(mlp[0])
                w2_16a = inj._f16_weight(mlp[2])
                dst.w2_f16, dst.b2 = w2_16a.data_ptr(), mlp[2].bias.data_ptr()
                _require_bf16_cuda("adaptor norm", norm.weight)
                dst.gamma, dst.beta, dst.alpha = norm.weight.data_ptr(), norm.bias.data_ptr(), alpha.data_ptr()
                a.adapt_alpha_dt, a.adapt_eps = nv._dt(alpha), norm.eps
                keep.append(w2_16a)
    if gc is not None:
        gc._check_native(None)
        if proj.global_logit is not None:
            raise NotImplementedError("the one-call executor has no clip-scale global stage (use forward_stepwise)")
        if gc.inject_in_call:
            q_in, n_rows = gc.query.detach(), gc.num_queries                # injected by the executor itself (hicom_compressor_args.inj_g)
            _require_bf16_cuda("global_compressor.query", q_in)
            _fill_injector(a.inj_g, gc.guide_injector, gc.use_guide, guide_embed, q_in, keep)
            a._guide_ptr_fields += ("inj_g.guide",)
            a.gq_dt = nv.DT_BF16
            keep.append(q_in)
        elif gc.external_queries:
            # injected through coarse / fine / an adapted guide (ref :642 with :369-397): f32 rows in a buffer this plan owns, refilled per call
            q_in, n_rows = gc.injected_queries(guide_embed)
            ext_g = torch.empty((q_in.shape[0], E), dtype=torch.float32, device=ff.device)
            q_in = ext_g
            a.gq_dt = nv.DT_F32
        else:
            q_in, n_rows = gc.injected_queries(guide_embed)
            a.gq_dt = nv.DT_BF16
            keep.append(q_in)
        att = gc.attn_layer
        a.gq, a.nq, a.nh, a.n_global_rows = q_in.data_ptr(), q_in.shape[0], att.num_heads, n_rows
        if gptr is not None and q_in.data_ptr() == gptr:
            a._guide_ptr_fields += ("gq",)
        a.wq, a.bq = _w(att.q_proj)
        a.wk, _ = _w(att.k_proj)
        a.wv, a.bv = _w(att.v_proj)
        a.wo, a.bo = _w(att.out_proj)
        a.gw0, a.gb0 = _w(gc.readout[0])
        a.gw2, a.gb2 = _w(gc.readout[2])
        a.gc0 = None
        if gc.use_guide == "direct" and q_in.shape[0] == 1 and gc.readout[0].bias is not None and att.out_proj.bias is not None:
            c0 = gc.readout_over_out_proj()                           # weight-only product: the five-launch step of the release recipe
            a.gc0 = c0.data_ptr()
            keep.append(c0)
        if gc.use_pos_emb:
            pe, kpe, cap = gc.pos_and_kpe(t_offset + T, H, W, ff.device)
            pe_hi, pe_lo = gc.pos_planes(t_offset + T, H, W, ff.device)
            a.pe, a.kpe, a.P = pe.data_ptr(), kpe.data_ptr(), pe.shape[0]
            a.pe_hi, a.pe_lo = pe_hi.data_ptr(), pe_lo.data_ptr()
            a.t_index0, a.y_index0, a.x_index0 = t_offset, cap, cap + H
            keep += [pe, kpe, pe_hi, pe_lo]
            a.vpe_f16, a.marg_slots = None, 0
            if a.gc0 and lc is not None and t_offset == 0 and state_out is None and os.environ.get("HICOM_RING_MARG", "0") == "1":
                # release step, opt-in (measured a net loss): the value-side pos-emb in the merge role (v_proj . pe^T,
                # weight-only) instead of behind the ring's token stream.  The table is only built -- and rebuilt by every training
                # step's refresh -- when the switch is on.
                vpe = gc.vpe_f16(T, H, W, ff.device)
                if vpe is not None:
                    a.vpe_f16, a.marg_slots = vpe.data_ptr(), vpe.shape[1]
                    keep.append(vpe)
        else:
            a.pe = a.kpe = a.pe_hi = a.pe_lo = None
            a.P = 0
            a.vpe_f16, a.marg_slots = None, 0
    a.out, a.out_dt, a.ldo = out.data_ptr(), nv._dt(out), out.shape[-1]
    a.local_row0 = local_row0
    a.nl_group = layout.nl_group if layout is not None else 0
    a.global_row0 = global_row0 if global_row0 is not None else (layout.n_rows if layout is not None else 0)
    a.nl_count = 0
    if layout is not None and layout.newline_rows:
        nl = image_newline
        keep.append(nl)
        a.newline, a.newline_dt = nl.data_ptr(), nv._dt(nl)
        a.nl_first = layout.newline_rows[0] + local_row0
        a.nl_step = layout.newline_rows[1] - layout.newline_rows[0] if len(layout.newline_rows) > 1 else 1
        a.nl_count = len(layout.newline_rows)
    a.local_out, a.state_out = _p(local_out), _p(state_out)
    a.state_sets, a.state_set_stride, a.nsets = _p(state_sets), state_set_stride, nsets
    a._keep = keep                      # keeps borrowed tensors alive until the call is enqueued
    a._queries = None
    if ext_l is not None or ext_g is not None:
        grid = (a.at.nwin, a.ay.nwin, a.ax.nwin) if lc is not None else None

        def queries(ff, guide, res=None, lc=lc, gc=gc, ext_l=ext_l, ext_g=ext_g, grid=grid, a=a):
            """Fills the plan's query rows for this call.  The global rows on the current stream (the stream kernel needs them first);
            the local rows -- pooling, adaptor, injector: up to ~10 launches, two of them GEMMs over all windows -- on `res.inj` beside
            the global stage's stream kernel, the executor waiting for `res.ev_lq` in front of the window kernel.  res=None (graph
            capture): everything on the current stream."""
            if ext_g is not None:
                gc.make_queries(guide, ext_g)
            a.ev_queries = None
            if ext_l is not None:
                if res is None or gc is None:
                    lc.make_queries(ff, guide, grid, ext_l[1], ext_l[0])
                else:
                    main = torch.cuda.current_stream(ff.device)
                    res.inj.wait_stream(main)          # (inputs ready; the previous call's readers of the rows are done)
                    with torch.cuda.stream(res.inj):
                        lc.make_queries(ff, guide, grid, ext_l[1], ext_l[0])
                        res.ev_lq.record(res.inj)
                    ff.record_stream(res.inj)
                    if guide is not None:
                        guide.record_stream(res.inj)
                    a.ev_queries = res.ev_lq.cuda_event
        a._queries = queries            # (the closure owns the buffers a.lq / a.gq point into)
    return a


def attach_execution(a: nv.CompressorArgs, device, main_stream=None, res=None) -> torch.Tensor:
    """Gives the argument block its OWN workspace (zero-prefixed once; the caller keeps the returned tensor alive with
    the plan) + streams/events.  Two plans never share a workspace: a deferred call's side / comm stream may still
    read its partial states while the next call -- possibly of another plan -- already streams."""
    if main_stream is None:
        main_stream = torch.cuda.current_stream(device)
    if res is None:
        res = _resources(device)
    total, prefix = nv.compressor_workspace(a)
    with torch.cuda.stream(main_stream):
        ws = torch.empty(total, dtype=torch.uint8, device=device)
        ws[:prefix].zero_()
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    a.stream_main = main_stream.cuda_stream
    a.stream_side = res.side.cuda_stream
    a.ev_fork, a.ev_join = res.ev_fork.cuda_event, res.ev_join.cuda_event
    a.ev_merge, a.defer_join = res.ev_merge.cuda_event, 0
    return ws


class _Plan:
    """A filled argument block for one (projector state, problem shape) combination and the workspace it owns, plus --
    in graph mode -- the captured hipGraph of its launch sequence and the static buffer it writes."""
    __slots__ = ("args", "ws", "rows", "hidden", "graph", "static_out", "hits", "fused", "sig", "guide_fields", "res", "fresh", "refresh",
                 "queries")

    def __init__(self, args, ws, rows, hidden, sig, res):
        self.args, self.ws, self.rows, self.hidden, self.sig, self.res = args, ws, rows, hidden, sig, res
        self.fused = nv.compressor_is_fused(args)
        self.guide_fields = args._guide_ptr_fields
        self.graph = None
        self.static_out = None
        self.hits = 0
        self.fresh = None          # weight CONTENT state (versions, epoch) the derived device caches were last built from
        self.refresh = None        # callable that rebuilds them in place
        self.queries = getattr(args, "_queries", None)     # producer of per-call query rows (coarse / fine injection, query adaptors), or None


_MAX_PLANS = 16      # per projector; plans live ON the module (they point into its cached device tables)


def _param_list(proj):
    """The parameter LIST is cached on the module (walking nn.Module.parameters() costs tens of microseconds) and
    refreshed whenever the module is moved / cast (_apply) or reloaded (load_state_dict)."""
    d = proj.__dict__
    gen = d.get("_engine_params_gen", 0)
    cached = d.get("_engine_params")
    if cached is None or cached[0] != gen:
        nv.track_parameters(proj)           # (writes through `.data` aliases move the weights epoch: native.TrackedParameter)
        cached = (gen, [p for p in proj.parameters()])
        d["_engine_params"] = cached
    return cached


def plan_sig(proj):
    """What a plan has baked in as ADDRESSES: identity of the module, generation of its parameter list, every parameter's storage
    pointer, the generation of the cached device tables (bumped when one is reallocated), the clip-scale logits."""
    gen, params = _param_list(proj)
    acc = 0
    for p in params:
        acc += p.data_ptr()
    gc = proj.global_compressor
    return (id(proj), gen, acc, 0 if gc is None else gc._cache_gen, proj.local_logit, proj.global_logit)


def content_sig(proj):
    """Weight CONTENT state: in-place version counters + the global weights epoch (training forwards,
    invalidate_weight_caches()).  When it moves, the weight-derived device caches are re-run IN PLACE and the plan stays."""
    _, params = _param_list(proj)
    acc = 0
    for p in params:
        acc += p._version
    return (acc, nv.weights_epoch())


def weights_sig(proj):
    """State of everything a plan has baked in besides the per-call pointers: identity of the module, generation of its
    parameter list, every parameter's storage pointer (`p.data = ...` swaps it without touching the version counter) and
    in-place version counter, the global weights epoch (training forwards / invalidate_weight_caches(): writes that bypass
    the version counter), the generation of the cached device tables, the clip-scale logits."""
    gen, params = _param_list(proj)
    acc = 0
    for p in params:
        acc += p.data_ptr() + p._version * 1000003
    gc = proj.global_compressor
    return (id(proj), gen, acc, nv.weights_epoch(), 0 if gc is None else gc._cache_gen, proj.local_logit, proj.global_logit)


def _evict_one(plans: dict):
    """Drops the oldest plan.  Its workspace may still be read by a deferred call's side stream: the caching allocator
    must not hand the block to another stream before that work has drained."""
    k = next(iter(plans))
    old = plans.pop(k)
    res = getattr(old, "res", None)
    ws = getattr(old, "ws", None)
    if res is not None and ws is not None:
        ws.record_stream(res.side)


def last_window_contexts(proj, ff, modal):
    """fp16 [windows, E] copy of the local stage's window contexts the LAST run_dense call of `proj` left in its plan's workspace (the A
    operand of readout GEMM 1), or None when that call did not read out through fp16 planes.  Valid until the plan's next call."""
    plan = proj.__dict__.get("_last_plan")
    lc = proj.local_compressor
    if plan is None or lc is None or not plan.args.has_local:
        return None
    off = nv.compressor_ctx16_offset(plan.args)
    if off < 0:
        return None
    T, H, W, E = ff.shape
    at, ay, ax = lc.tilings(T, H, W, modal)
    nw = at.nwin * ay.nwin * ax.nwin
    return plan.ws[off:off + nw * E * 2].view(torch.float16).view(nw, E).clone()


def run_anyres(proj, frames_feature, frames_embed, guide_embed, modal, image_newline, out_dtype):
    """HIComProjector.forward for the anyres DICT input of an image (reference projector.py:679-700) through hicom_compressor_fwd: one call
    for the base image (local stage; rows first), one for the patch grid (local stage with the anyres packing + global stage) -- two C
    calls instead of the ~30 of the operator-by-operator path (0.37 ms of host time per image at 27 x 27 + 54 x 54)."""
    lc, gc = proj.local_compressor, proj.global_compressor
    base, patch = frames_feature["base"], frames_feature["patch"]
    has_nl = image_newline is not None
    n_base = 0
    if lc is not None and base is not None:
        at, ay, ax = lc.tilings(1, base.shape[0], base.shape[1], modal)
        n_base = proj._layout((at.nwin, ay.nwin, ax.nwin), modal, has_nl, False).n_rows
    n_patch = 0
    if lc is not None:
        at, ay, ax = lc.tilings(1, patch.shape[0], patch.shape[1], modal)
        n_patch = proj._layout((at.nwin, ay.nwin, ax.nwin), modal, has_nl, True).n_rows
    n_global = gc.num_queries if gc is not None else 0
    hidden = (lc or gc).readout[2].out_features
    out = torch.empty((n_base + n_patch + n_global, hidden), dtype=out_dtype, device=patch.device)
    emb = frames_embed if frames_embed is not None else {"base": None, "patch": None}
    if n_base:
        fe = emb["base"]
        run_dense(proj, base.unsqueeze(0), None if fe is None else fe.unsqueeze(0), guide_embed, modal, image_newline, out_dtype,
                  segment=dict(stages=("local",), is_anyres=False, out=out, row0=0))
    fe = emb["patch"]
    run_dense(proj, patch.unsqueeze(0), None if fe is None else fe.unsqueeze(0), guide_embed, modal, image_newline, out_dtype,
              segment=dict(stages=tuple(s for s, c in (("local", lc), ("global", gc)) if c is not None), is_anyres=True, out=out, row0=n_base))
    return out


def run_dense(proj, ff, fe, guide_embed, modal, image_newline, out_dtype, deferred: bool = False, local_logits=None, adapt_y=None, segment=None):
    """HIComProjector.forward for a dense [T,H,W,E] input through hicom_compressor_fwd.

    adapt_y = (y_k, y_v): fp16 [T*H*W, E] outputs of the k / v adaptor MLPs the caller computed itself (the training forward keeps
    their intermediates for the backward): the executor then skips those GEMMs (hicom_adaptor.y).

    With `proj.graph_replay = True` the launch sequence of a plan is captured into a hipGraph on its second use and
    replayed afterwards (one graph launch + one device copy of the result); graph plans are additionally keyed by the
    input buffers, whose pointers the captured nodes hold.

    deferred=True returns (out, event): the main stream does not wait for the side stream's global chain (merge +
    the four small linears that produce the 32 global rows); `event` fires when they are written.  Back-to-back
    forwards of independent videos then overlap that latency-bound chain with the next video's streaming."""
    from .projector import _require_bf16_cuda
    _require_bf16_cuda("frames_feature", ff)          # fail loudly on CPU tensors before touching any stream
    stages = segment["stages"] if segment is not None else ("local", "global")
    lc = proj.local_compressor if "local" in stages else None
    gc = proj.global_compressor if "global" in stages else None
    dev = ff.device
    ff = ff.contiguous()
    fe = fe.contiguous() if fe is not None else None
    guide = guide_embed.contiguous() if guide_embed is not None else None
    nl = image_newline.contiguous() if image_newline is not None else None
    ll = local_logits.contiguous() if local_logits is not None else None
    graph = bool(getattr(proj, "graph_replay", False)) and not deferred and segment is None
    res = _resources(dev)
    seg_key = None if segment is None else (stages, segment["is_anyres"], segment["row0"])
    key = (tuple(ff.shape), None if fe is None else tuple(fe.shape), None if guide is None else tuple(guide.shape), modal,
           None if nl is None else tuple(nl.shape), out_dtype,
           torch.cuda.current_stream(dev).cuda_stream,
           (ff.data_ptr(), _p(fe), _p(guide), _p(nl), _p(ll), None if adapt_y is None else tuple(_p(t) for t in adapt_y)) if graph else None,
           ll is not None,
           None if adapt_y is None else tuple(t is not None for t in adapt_y),     # (supplied adaptor outputs: the workspace has no regions for them)
           seg_key)
    plans = proj.__dict__.setdefault("_engine_plans", {})
    plan = plans.get(key)
    sig = plan_sig(proj)
    if plan is not None and plan.sig == sig:
        fresh = content_sig(proj)
        if plan.fresh != fresh:
            # the weights changed under a plan whose addresses are all still valid (an optimizer step): re-run the producers of
            # the weight-derived tables into their existing buffers instead of rebuilding the plan
            plan.refresh()
            plan.args.reuse_queries = 0            # guide off: the folded learnable queries are weight-derived too
            sig = plan_sig(proj)                   # (a table that had to be reallocated bumps the cache generation)
            plan.fresh = fresh
    if plan is not None and plan.sig != sig:
        plans.pop(key)
        plan = None
    if plan is None:
        T, H, W, _ = ff.shape
        layout = None
        n_local = 0
        if lc is not None:
            at, ay, ax = lc.tilings(T, H, W, modal)
            layout = proj._layout((at.nwin, ay.nwin, ax.nwin), modal, nl is not None, bool(segment and segment["is_anyres"]))
            n_local = layout.n_rows
        n_global = gc.num_queries if gc is not None else 0
        hidden = (lc or gc).readout[2].out_features
        row0 = segment["row0"] if segment is not None else 0
        probe = segment["out"] if segment is not None else torch.empty((n_local + n_global, hidden), dtype=out_dtype, device=dev)
        a = build_args(proj, ff, fe, guide, modal, nl, probe, layout, global_row0=row0 + n_local, stages=stages, local_row0=row0)
        if adapt_y is not None:             # (before the workspace is sized: make_layout skips the regions of supplied outputs)
            a.ak.y = _p(adapt_y[0])
            a.av.y = _p(adapt_y[1])
        if ll is not None:
            a.local_logits = ll.data_ptr()
            if not nv.compressor_is_fused(a):
                raise NotImplementedError("local_logits=: this geometry does not run on the fused stream kernel (windows must "
                                          "partition the grid; see hicom_fused_stream_fwd in include/hicom_hip.h)")
        ws = attach_execution(a, dev, res=res)
        a._keep = None                 # the plan does not pin the caller's tensors: their pointers are patched per call
        sig = plan_sig(proj)           # (build_args may have (re)built the cached positional tables)
        if len(plans) >= _MAX_PLANS:
            _evict_one(plans)
        plan = plans[key] = _Plan(a, ws, n_local + n_global, hidden, sig, res)
        plan.fresh = content_sig(proj)
        use_gc0 = bool(a.gc0)
        use_vpe = bool(a.vpe_f16)

        def refresh(lc=lc, gc=gc, T=T, H=H, W=W, dev=dev, use_gc0=use_gc0, use_vpe=use_vpe):
            if lc is not None:
                lc.readout_f16()
                if lc.adapt_k or lc.adapt_v:
                    from . import injector as inj
                    for on, mlp in ((lc.adapt_k, lc.k_proj), (lc.adapt_v, lc.v_proj)):
                        if on:
                            inj._f16_weight(mlp[2])
            if gc is not None:
                if gc.use_pos_emb:
                    gc.pos_and_kpe(T, H, W, dev)
                    if use_vpe:
                        gc.vpe_f16(T, H, W, dev)
                if use_gc0:
                    gc.readout_over_out_proj()
        plan.refresh = refresh
    plan.hits += 1
    a = plan.args
    # per-call pointers
    a.ff = ff.data_ptr()
    if fe is not None:
        a.fe = fe.data_ptr()
    if ll is not None:
        a.local_logits = ll.data_ptr()
    a.ak.y = adapt_y[0].data_ptr() if adapt_y is not None and adapt_y[0] is not None else None
    a.av.y = adapt_y[1].data_ptr() if adapt_y is not None and adapt_y[1] is not None else None
    if guide is not None:
        gp = guide.data_ptr()
        for f in plan.guide_fields:
            obj, _, leaf = f.rpartition(".")
            setattr(getattr(a, obj) if obj else a, leaf, gp)
    if nl is not None and a.nl_count > 0:
        a.newline = nl.data_ptr()
    if plan.queries is not None and deferred:
        raise NotImplementedError("forward_deferred: recipes whose queries are injected per call (coarse / fine, query adaptors) run joined")
    if graph:
        a.ev_join, a.defer_join = res.ev_join.cuda_event, 0
        if plan.graph is None:
            plan.static_out = torch.empty((plan.rows, plan.hidden), dtype=out_dtype, device=dev)
            a.out = plan.static_out.data_ptr()
            if plan.queries is not None:
                plan.queries(ff, guide)
            nv.compressor_fwd(a)                       # warm (lazy module loads must not happen in capture)
            torch.cuda.current_stream(dev).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.Stream(device=dev)):
                # inside capture torch's current stream is the capture stream: re-point the plan at it
                a.stream_main = torch.cuda.current_stream(dev).cuda_stream
                if plan.queries is not None:
                    plan.queries(ff, guide)
                nv.compressor_fwd(a)
            plan.graph = g
        plan.graph.replay()
        proj.__dict__["_last_plan"] = plan             # (last_window_contexts reads the workspace of the call that just ran)
        return plan.static_out.clone()                 # callers own their result (no aliasing across calls)
    if segment is not None:
        if deferred:
            raise NotImplementedError("deferred segments")
        out = segment["out"]
    else:
        out = torch.empty((plan.rows, plan.hidden), dtype=out_dtype, device=dev)
    a.out = out.data_ptr()
    a.defer_join = int(bool(deferred))
    if deferred:
        ev = res.done[res.n_done % len(res.done)]      # one of a small ring of reusable events, recorded from C at
        res.n_done += 1                                # the end of the side stream's chain (no second record)
        a.ev_join = ev.cuda_event
    else:
        a.ev_join = res.ev_join.cuda_event
    if plan.queries is not None:
        plan.queries(ff, guide, res)                   # pooling / adaptor / injector launches -> the plan's query rows
    nv.compressor_fwd(a)
    proj.__dict__["_last_plan"] = plan                 # (the training forward picks the window contexts out of its workspace)
    if gc is not None and gc.use_guide in (None, "off") and gc.queries_native and not plan.fused:
        a.reuse_queries = 1                            # q_proj + fold of the learnable queries: weight-only, they stay in the plan's workspace
    if deferred:
        # the side stream is still busy with the 32 global rows: it writes `out` and reads the guide (the residual of
        # out_proj) after this call has returned
        out.record_stream(res.side)
        if guide is not None:
            guide.record_stream(res.side)
        return out, ev
    return out
