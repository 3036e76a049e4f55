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

import numpy as np
import torch

from . import native as nv

IGNORE_INDEX = -100                                                   # reference hicom/constants.py:7
MODAL_INDEX_MAP = {"<image>": -200, "<video>": -201, "<audio>": -202}  # reference hicom/constants.py:30-34


class SplicePlan:
    """Integer plan of hicom_arch.py:283-372 for one batch of ids.

    src_kind [B, Lmax] int32 : >= 0 position in the sample's ids | -1 visual row | -2 padding
    src_feat [B, Lmax, 2] int32 : (feature index, row) for visual rows
    new_len [B] int32, Lmax
    feat_at [K, 3] int64 : (sample, first output position, rows) of every feature that was placed (rows may be 0); features a
                           text-only sample consumes without placing them (:289-299) have rows = 0 there too
    """
    __slots__ = ("src_kind", "src_feat", "new_len", "Lmax", "feat_at")

    def __iter__(self):                                               # (kind, feat, new_len, Lmax) = plan_layout(...)
        return iter((self.src_kind, self.src_feat, self.new_len, self.Lmax))


def plan_layout(ids: np.ndarray, feat_rows: List[int]) -> SplicePlan:
    """Vectorised (numpy cumsum / repeat) layout plan; see SplicePlan.  ids [B, S] int64 (host), feat_rows[k] = rows of
    mm_features[k]."""
    B, S = ids.shape
    K = len(feat_rows)
    rows_k = np.asarray(feat_rows, dtype=np.int64)
    is_mm = (ids == -200) | (ids == -201) | (ids == -202)
    nmm = is_mm.sum(axis=1)
    # a pure-text sample still consumes one feature slot, of which it takes zero rows (:289-299)
    slots = np.where(nmm == 0, 1, nmm)
    first = np.cumsum(slots) - slots                                  # first feature index of each sample
    if int(first[-1] + slots[-1]) > K:
        raise IndexError("list index out of range")                   # what mm_features[cur_mm_idx] raises in the reference
    # feature index of every placeholder: first[b] + its rank among the sample's placeholders
    rank = np.cumsum(is_mm, axis=1) - 1
    k_of = np.where(is_mm, first[:, None] + rank, 0)
    lens = np.where(is_mm, rows_k[k_of] if K else 0, 1).astype(np.int64)   # output rows each input position expands to
    new_len = lens.sum(axis=1)
    Lmax = int(new_len.max())
    start = np.cumsum(lens, axis=1) - lens                            # first output position of each input position
    flat_lens = lens.ravel()
    total = int(flat_lens.sum())
    src = np.repeat(np.arange(B * S, dtype=np.int64), flat_lens)      # input position (flattened) of every output row
    b_of = src // S
    p_of = src - b_of * S
    row0 = np.cumsum(new_len) - new_len                               # first output row (flattened, unpadded) of each sample
    within = np.arange(total, dtype=np.int64) - row0[b_of]            # output position inside the sample
    dest = b_of * Lmax + within
    mm_row = is_mm.ravel()[src]
    r_of = within - start.ravel()[src]                                # row inside the feature for visual rows
    plan = SplicePlan()
    kind = np.full(B * Lmax, -2, dtype=np.int32)
    kind[dest] = np.where(mm_row, -1, p_of).astype(np.int32)
    feat = np.zeros((B * Lmax, 2), dtype=np.int32)
    feat[dest, 0] = np.where(mm_row, k_of.ravel()[src], 0)
    feat[dest, 1] = np.where(mm_row, r_of, 0)
    plan.src_kind, plan.src_feat = kind.reshape(B, Lmax), feat.reshape(B, Lmax, 2)
    plan.new_len, plan.Lmax = new_len.astype(np.int32), Lmax
    fa = np.zeros((K, 3), dtype=np.int64)
    bb, pp = np.nonzero(is_mm)
    kk = k_of[bb, pp]
    fa[kk, 0], fa[kk, 1], fa[kk, 2] = bb, start[bb, pp], rows_k[kk] if K else 0
    plan.feat_at = fa
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
    plan = plan_layout(ids, [f.shape[0] for f in feats])
    src_kind, src_feat, new_len, Lmax = plan
    ragged = bool((new_len != new_len[0]).any())
    if ragged and attention_mask is not None and labels is None:
        # the reference's ragged branch builds the mask from `_new_labels`, which only exists when labels were given (:345,:352)
        raise UnboundLocalError("local variable '_new_labels' referenced before assignment")
    if attention_mask is not None and int(new_len.min()) < S:
        # zero-row features: the reference's left mask padding is torch.full((new_len - S,), True) (:355, :370)
        raise RuntimeError(f"Trying to create tensor with negative dimension {int(new_len.min()) - S}")
    # device-pointer table of the output rows + label map + lengths: ONE packed upload
    text = src_kind >= 0
    tok = ids[np.nonzero(text)[0], src_kind[text]]
    if tok.size and (int(tok.min()) < 0 or int(tok.max()) >= weight.shape[0]):
        raise IndexError("index out of range in self")                # nn.Embedding's error for a bad id
    n = B * Lmax
    need_maps = labels is not None or attention_mask is not None
    packed = np.zeros(n * 8 + (n * 4 + (B * 4 + 15) // 16 * 16 if need_maps else 0), dtype=np.uint8)
    table = packed[:n * 8].view(np.int64).reshape(B, Lmax)
    table[text] = weight.data_ptr() + tok * row_bytes
    vis = src_kind == -1
    if vis.any():
        fbase = np.array([f.data_ptr() for f in feats], dtype=np.int64)
        table[vis] = fbase[src_feat[..., 0][vis]] + src_feat[..., 1][vis].astype(np.int64) * row_bytes
    if need_maps:
        packed[n * 8:n * 12].view(np.int32)[:] = src_kind.ravel()
        packed[n * 12:n * 12 + B * 4].view(np.int32)[:] = new_len
    packed_d = torch.from_numpy(packed).to(dev)
    table_d = packed_d[:n * 8].view(torch.int64).view(B, Lmax)
    needs_grad = torch.is_grad_enabled() and (weight.requires_grad or any(f.requires_grad for f in feats))
    if needs_grad:
        flat_rows = np.nonzero(text.ravel())[0]
        text_rows = torch.from_numpy(flat_rows).to(dev)
        text_tok = torch.from_numpy(np.ascontiguousarray(tok)).to(dev)
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
