"""MI355X-native form of the step right behind the compressor: splicing the compressed visual
tokens into the LLM's input embeddings at the <image> / <video> / <audio> placeholders, with the label / attention-mask /
padding fix-ups -- reference `HIComMetaForCausalLM.prepare_inputs_labels_for_multimodal`, hicom/model/hicom_arch.py:271-373.

`prepare_inputs_labels_for_multimodal(embed_tokens, input_ids, attention_mask, past_key_values, labels, mm_features)` is
the part of that method behind `mm_features = self.encode_images_or_videos(...)` (:281) and returns the same 5-tuple
`(None, attention_mask, past_key_values, new_input_embeds, new_labels)`.

The host reads the (tiny) id tensor once and plans the new layout with vectorised integer arithmetic (the output SHAPE
depends on the ids, so the reference synchronises here too); the plan travels to the device as ONE packed upload; the rows
themselves -- tens of MB at LLM width -- are written exactly once by hicom_splice_rows_fwd from a device-pointer table, and
hicom_splice_labels_fwd builds labels / mask on the device.

Training: the reference builds `new_input_embeds` with `embed_tokens(...)` and `torch.cat` (:292-331), so gradients reach the
projector (through `mm_features`) and the embedding table.  The row placement here is a `torch.autograd.Function` with the
same two gradient paths: d mm_features[k] = the (contiguous) block of output rows feature k was placed at, and
d embed_tokens.weight = index_add of the text rows.
"""
from __future__ import annotations

from typing import List, Optional

import threading

import numpy as np
import torch

from . import native as nv

_PLAN_CACHE: dict = {}       # content hash of (ids, feature rows, embedding table) -> layout plan + pointer table (see below)
_MAX_PLANS = 8
_PLAN_LOCK = threading.Lock()   # guards the dict itself; entries are immutable apart from ONE reference store (`upload`)

IGNORE_INDEX = -100                                                   # reference hicom/constants.py:7
MODAL_INDEX_MAP = {"<image>": -200, "<video>": -201, "<audio>": -202}  # reference hicom/constants.py:30-34


class SplicePlan:
    """Integer plan of hicom_arch.py:283-372 for one batch of ids.

    segments : (sample b, first output position o, first id position p0, text ids nt, feature k, feature rows n) -- a run of
               text ids followed by the rows of feature k (k = -1, n = 0: the trailing text run)
    new_len [B] int32, Lmax
    feat_at [K, 3] int64 : (sample, first output position, rows) of every feature that was placed (rows may be 0); features a
                           text-only sample consumes without placing them (:289-299) have rows = 0 there too
    src_kind [B, Lmax] int32 : >= 0 position in the sample's ids | -1 visual row | -2 padding      (built on first use)
    src_feat [B, Lmax, 2] int32 : (feature index, row) for visual rows                             (built on first use)
    """
    __slots__ = ("new_len", "Lmax", "feat_at", "segments", "B", "_kind", "_feat")

    def _maps(self):
        kind = np.full((self.B, self.Lmax), -2, dtype=np.int32)
        feat = np.zeros((self.B, self.Lmax, 2), dtype=np.int32)
        for b, o, p0, nt, k, n in self.segments:
            if nt:
                kind[b, o:o + nt] = np.arange(p0, p0 + nt, dtype=np.int32)
            if n:
                kind[b, o + nt:o + nt + n] = -1
                feat[b, o + nt:o + nt + n, 0] = k
                feat[b, o + nt:o + nt + n, 1] = np.arange(n, dtype=np.int32)
        self._kind, self._feat = kind, feat

    @property
    def src_kind(self):
        if self._kind is None:
            self._maps()
        return self._kind

    @property
    def src_feat(self):
        if self._feat is None:
            self._maps()
        return self._feat

    def __iter__(self):                                               # (kind, feat, new_len, Lmax) = plan_layout(...)
        return iter((self.src_kind, self.src_feat, self.new_len, self.Lmax))


def plan_layout(ids: np.ndarray, feat_rows: List[int]) -> SplicePlan:
    """Layout plan by SEGMENTS (text run | feature block | text run ...): a handful of numpy slice assignments per
    placeholder instead of per-position work -- prompts carry one or two placeholders among thousands of ids.  See SplicePlan.
    ids [B, S] int64 (host), feat_rows[k] = rows of mm_features[k]."""
    B, S = ids.shape
    K = len(feat_rows)
    bb, pp = np.nonzero((ids <= -200) & (ids >= -202))                # placeholders, sample-major then position
    counts = np.bincount(bb, minlength=B)
    # a pure-text sample still consumes one feature slot, of which it takes zero rows (:289-299)
    slots = np.where(counts == 0, 1, counts)
    first = np.cumsum(slots) - slots                                  # first feature index of each sample
    if int(first[-1] + slots[-1]) > K:
        raise IndexError("list index out of range")                   # what mm_features[cur_mm_idx] raises in the reference
    start_of = np.cumsum(counts) - counts                             # index of each sample's first placeholder in bb / pp
    new_len = np.full(B, S, dtype=np.int64)
    segs = []                                                         # (b, out0, p0, n_text, k, n_rows): a text run, then feature k
    fa = np.zeros((K, 3), dtype=np.int64)
    for b in range(B):
        cur, p0 = 0, 0
        for m in range(int(start_of[b]), int(start_of[b] + counts[b])):
            p, k = int(pp[m]), int(first[b] + m - start_of[b])
            n = int(feat_rows[k])
            segs.append((b, cur, p0, p - p0, k, n))
            fa[k] = (b, cur + p - p0, n)
            cur += p - p0 + n
            p0 = p + 1
        segs.append((b, cur, p0, S - p0, -1, 0))                      # trailing text (the whole sample when it has no placeholder)
        new_len[b] = cur + S - p0
    plan = SplicePlan()
    plan.B, plan.new_len, plan.Lmax, plan.feat_at, plan.segments = B, new_len.astype(np.int32), int(new_len.max()), fa, segs
    plan._kind = plan._feat = None
    return plan


class _SpliceRows(torch.autograd.Function):
    """new_input_embeds [B, Lmax, hidden] from the device-pointer table (forward: one HIP launch) with the two gradient
    paths of the reference's embed_tokens + torch.cat (hicom_arch.py:292-331)."""

    @staticmethod
    def forward(ctx, weight, table_d, shape, text_rows, text_tok, feat_at, *feats):
        out = torch.empty(shape, dtype=weight.dtype, device=weight.device)
        nv.splice_rows(table_d, out)
        ctx.text_rows, ctx.text_tok, ctx.feat_at = text_rows, text_tok, feat_at
        ctx.wshape, ctx.frows = weight.shape, [f.shape[0] for f in feats]
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, Lmax, hidden = d_out.shape
        flat = d_out.reshape(B * Lmax, hidden)
        d_weight = None
        if ctx.needs_input_grad[0]:
            # embedding backward (:292-295, :313): every text row adds into the row of its token id
            d_weight = torch.zeros(ctx.wshape, dtype=d_out.dtype, device=d_out.device)
            if ctx.text_rows.numel():
                d_weight.index_add_(0, ctx.text_tok, flat.index_select(0, ctx.text_rows))
        d_feats = []
        for k, n in enumerate(ctx.frows):
            if not ctx.needs_input_grad[6 + k]:
                d_feats.append(None)
                continue
            b, p0, rows = (int(v) for v in ctx.feat_at[k])
            if rows == 0:                                             # consumed by a text-only sample / zero rows (cat of [0:0])
                d_feats.append(torch.zeros((n, hidden), dtype=d_out.dtype, device=d_out.device))
            else:
                d_feats.append(d_out[b, p0:p0 + rows].clone())
        return (d_weight, None, None, None, None, None, *d_feats)


def prepare_inputs_labels_for_multimodal(embed_tokens, input_ids, attention_mask, past_key_values, labels,
                                         mm_features: Optional[List[torch.Tensor]]):
    """See the module docstring.  embed_tokens: nn.Embedding (or its weight [vocab, hidden]); mm_features: the list that
    `encode_images_or_videos` returns (one [n_k, hidden] tensor per image / video), or None for text-only calls."""
    if mm_features is None or input_ids.shape[1] == 1:               # text-only / decode step (:275-279)
        return input_ids, attention_mask, past_key_values, None, labels
    weight = embed_tokens.weight if hasattr(embed_tokens, "weight") else embed_tokens
    if not weight.is_cuda:
        raise nv.HicomNativeError("splice: hicom_amd runs on the GPU only (embedding table on the CPU)")
    dev, hidden = weight.device, weight.shape[1]
    row_bytes = hidden * weight.element_size()
    feats = []
    for f in mm_features:
        if f.ndim != 2:
            raise ValueError(f"splice: visual features are [rows, hidden] tensors (got {tuple(f.shape)})")
        if f.dtype != weight.dtype or f.shape[-1] != hidden or f.device != dev:
            raise ValueError("splice: visual features must have the embedding table's dtype, width and device")
        f = f.contiguous()
        if f.data_ptr() % 16 and f.shape[0]:
            f = f.clone()                                             # (a view at an odd storage offset: 16-byte vector loads)
        feats.append(f)
    if row_bytes % 16 or weight.data_ptr() % 16 or not weight.is_contiguous():
        raise ValueError("splice: the embedding table must be contiguous with 16-byte aligned rows")
    B, S = input_ids.shape
    if labels is not None and labels.shape != input_ids.shape:
        raise AssertionError("labels and input_ids differ in shape")  # (:305)
    if labels is not None and labels.dtype != torch.int64:
        raise ValueError("splice: labels must be torch.long")
    if attention_mask is not None and attention_mask.dtype not in (torch.bool, torch.int64):
        raise ValueError("splice: attention_mask must be torch.bool or torch.long")
    ids = input_ids.detach().cpu().numpy().astype(np.int64, copy=False)   # the one host read (the output shape depends on it)
    need_maps = labels is not None or attention_mask is not None
    feat_rows = tuple(f.shape[0] for f in feats)
    feat_ptrs = tuple(f.data_ptr() for f in feats)
    # A serving loop splices the SAME prompt template around every video: the layout plan and the text half of the pointer table
    # depend on (ids, feature row counts, the embedding table) only and are kept per content hash; a call whose feature tensors
    # sit at the addresses of the cached call (the caching allocator hands the same blocks back) re-uses the uploaded table too.
    # (the key names the embedding table by identity AND address: a freed table's address can be handed to a new one of the same shape)
    ckey = (hash(ids.tobytes()), ids.shape, feat_rows, id(weight), weight.data_ptr(), weight.shape[0], row_bytes, need_maps)
    with _PLAN_LOCK:
        hit = _PLAN_CACHE.get(ckey)
    if hit is not None and (not np.array_equal(hit["ids"], ids) or hit["weight"]() is not weight):   # (hash collision / recycled id)
        hit = None
    if hit is None:
        plan = plan_layout(ids, list(feat_rows))
    else:
        plan = hit["plan"]
    new_len, Lmax = plan.new_len, plan.Lmax
    ragged = bool((new_len != new_len[0]).any())
    if ragged and attention_mask is not None and labels is None:
        # the reference's ragged branch builds the mask from `_new_labels`, which only exists when labels were given (:345,:352)
        raise UnboundLocalError("local variable '_new_labels' referenced before assignment")
    if attention_mask is not None and int(new_len.min()) < S:
        # zero-row features: the reference's left mask padding is torch.full((new_len - S,), True) (:355, :370)
        raise RuntimeError(f"Trying to create tensor with negative dimension {int(new_len.min()) - S}")
    # device-pointer table of the output rows + label map + lengths: ONE packed upload, filled segment by segment
    n = B * Lmax
    packed_d = None
    cached_upload = None if hit is None else hit["upload"]            # ONE read: (feature pointers, device table) published together
    if cached_upload is not None and cached_upload[0] == feat_ptrs and cached_upload[1].device == dev:
        packed_d = cached_upload[1]
    else:
        if hit is None:
            packed = np.zeros(n * 8 + (n * 4 + (B * 4 + 15) // 16 * 16 if need_maps else 0), dtype=np.uint8)
            table = packed[:n * 8].view(np.int64).reshape(B, Lmax)
            wptr, vocab = weight.data_ptr(), weight.shape[0]
            for b, o, p0, nt, k, nrows in plan.segments:
                if nt:
                    tk = ids[b, p0:p0 + nt]
                    if int(tk.min()) < 0 or int(tk.max()) >= vocab:
                        raise IndexError("index out of range in self")    # nn.Embedding's error for a bad id
                    table[b, o:o + nt] = wptr + tk * row_bytes
            if need_maps:
                packed[n * 8:n * 12].view(np.int32)[:] = plan.src_kind.ravel()
                packed[n * 12:n * 12 + B * 4].view(np.int32)[:] = new_len
            import weakref
            hit = {"ids": ids.copy(), "plan": plan, "packed": packed, "upload": None, "weight": weakref.ref(weight)}
            with _PLAN_LOCK:
                if len(_PLAN_CACHE) >= _MAX_PLANS:
                    _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
                _PLAN_CACHE[ckey] = hit
        # the cached host table is a TEMPLATE (text half final) and is never written again: the visual rows are re-pointed in a
        # private copy, so two threads that splice the same prompt concurrently (serving workers) cannot mix their feature pointers
        packed = hit["packed"].copy()
        table = packed[:n * 8].view(np.int64).reshape(B, Lmax)
        for b, o, p0, nt, k, nrows in plan.segments:
            if nrows:
                table[b, o + nt:o + nt + nrows] = feat_ptrs[k] + np.arange(nrows, dtype=np.int64) * row_bytes
        packed_d = torch.from_numpy(packed).to(dev)                       # (a fresh device buffer: an earlier call's table may be in flight)
        hit["upload"] = (feat_ptrs, packed_d)                             # (one reference store: readers see the pair or the old pair)
    table_d = packed_d[:n * 8].view(torch.int64).view(B, Lmax)
    needs_grad = torch.is_grad_enabled() and (weight.requires_grad or any(f.requires_grad for f in feats))
    if needs_grad:
        text = plan.src_kind >= 0
        flat_rows = np.nonzero(text.ravel())[0]
        text_rows = torch.from_numpy(flat_rows).to(dev)
        text_tok = torch.from_numpy(np.ascontiguousarray(ids[np.nonzero(text)[0], plan.src_kind[text]])).to(dev)
        new_input_embeds = _SpliceRows.apply(weight, table_d, (B, Lmax, hidden), text_rows, text_tok, plan.feat_at, *feats)
    else:
        new_input_embeds = torch.empty((B, Lmax, hidden), dtype=weight.dtype, device=dev)
        nv.splice_rows(table_d, new_input_embeds)
    new_labels = None
    new_mask = attention_mask
    if need_maps:
        map_d = packed_d[n * 8:n * 12].view(torch.int32).view(B, Lmax)
        len_d = packed_d[n * 12:n * 12 + B * 4].view(torch.int32)
        if labels is not None:
            new_labels = torch.empty((B, Lmax), dtype=labels.dtype, device=dev)
        if attention_mask is not None:
            new_mask = torch.empty((B, Lmax), dtype=attention_mask.dtype, device=dev)
        nv.splice_labels(labels.contiguous() if labels is not None else None,
                         attention_mask.contiguous() if attention_mask is not None else None, map_d, len_d, S, IGNORE_INDEX,
                         new_labels, new_mask if attention_mask is not None else None)
    return None, new_mask, past_key_values, new_input_embeds, new_labels
