"""MI355X-native form of the step right behind the compressor: splicing the compressed visual
tokens into the LLM's input embeddings at the <image> / <video> / <audio> placeholders, with the label / attention-mask /
padding fix-ups -- reference `HIComMetaForCausalLM.prepare_inputs_labels_for_multimodal`, hicom/model/hicom_arch.py:271-373.

`prepare_inputs_labels_for_multimodal(embed_tokens, input_ids, attention_mask, past_key_values, labels, mm_features)` is
the part of that method behind `mm_features = self.encode_images_or_videos(...)` (:281) and returns the same 5-tuple
`(None, attention_mask, past_key_values, new_input_embeds, new_labels)`.

The host reads the (tiny) id tensor once and plans the new layout with integer arithmetic (the output SHAPE depends on the
ids, so the reference synchronises here too); the rows themselves -- tens of MB at LLM width -- are written exactly once by
hicom_splice_rows_fwd from a device-pointer table, and hicom_splice_labels_fwd builds labels / mask on the device.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import native as nv

IGNORE_INDEX = -100                                                   # reference hicom/constants.py:7
MODAL_INDEX_MAP = {"<image>": -200, "<video>": -201, "<audio>": -202}  # reference hicom/constants.py:30-34


def plan_layout(ids: np.ndarray, feat_rows: List[int]):
    """Integer plan of hicom_arch.py:283-372.  ids [B, S] int64 (host), feat_rows[k] = rows of mm_features[k].
    Returns (src_kind [B, Lmax] int32: >= 0 position in the sample's ids | -1 visual row | -2 padding,
             src_feat [B, Lmax, 2] int32: (feature index, row) for visual rows, new_len [B], Lmax)."""
    B, S = ids.shape
    mm_vals = list(MODAL_INDEX_MAP.values())
    per_sample = []
    cur = 0
    for b in range(B):
        row = ids[b]
        is_mm = np.isin(row, mm_vals)
        kinds, feats = [], []
        if not is_mm.any():
            # pure text: the sample still consumes one feature slot, of which it takes zero rows (:289-299)
            if cur >= len(feat_rows):
                raise IndexError("list index out of range")          # what mm_features[cur_mm_idx] raises in the reference
            kinds = list(range(S))
            feats = [(0, 0)] * S
            cur += 1
        else:
            for p in range(S):
                if is_mm[p]:
                    if cur >= len(feat_rows):
                        raise IndexError("list index out of range")
                    n = feat_rows[cur]
                    kinds += [-1] * n
                    feats += [(cur, r) for r in range(n)]
                    cur += 1
                else:
                    kinds.append(p)
                    feats.append((0, 0))
        per_sample.append((kinds, feats))
    new_len = np.array([len(k) for k, _ in per_sample], dtype=np.int32)
    Lmax = int(new_len.max())
    src_kind = np.full((B, Lmax), -2, dtype=np.int32)
    src_feat = np.zeros((B, Lmax, 2), dtype=np.int32)
    for b, (kinds, feats) in enumerate(per_sample):
        src_kind[b, :len(kinds)] = kinds
        src_feat[b, :len(kinds)] = feats
    return src_kind, src_feat, new_len, Lmax


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
    feats = [f.contiguous() for f in mm_features]
    for f in feats:
        if f.dtype != weight.dtype or f.shape[-1] != hidden or f.device != dev:
            raise ValueError("splice: visual features must have the embedding table's dtype, width and device")
    B, S = input_ids.shape
    if labels is not None and labels.shape != input_ids.shape:
        raise AssertionError("labels and input_ids differ in shape")  # (:305)
    ids = input_ids.detach().cpu().numpy().astype(np.int64)          # the one host read (the output shape depends on it)
    src_kind, src_feat, new_len, Lmax = plan_layout(ids, [f.shape[0] for f in feats])
    ragged = bool((new_len != new_len[0]).any())
    if ragged and attention_mask is not None and labels is None:
        # the reference's ragged branch builds the mask from `_new_labels`, which only exists when labels were given (:345,:352)
        raise UnboundLocalError("local variable '_new_labels' referenced before assignment")
    # device-pointer table of the output rows
    row_bytes = hidden * weight.element_size()
    tok = np.where(src_kind >= 0, ids[np.arange(B)[:, None], np.maximum(src_kind, 0)], 0)
    if (tok[src_kind >= 0] < 0).any() or (tok[src_kind >= 0] >= weight.shape[0]).any():
        raise IndexError("index out of range in self")                # nn.Embedding's error for a bad id
    table = np.zeros((B, Lmax), dtype=np.int64)
    text = src_kind >= 0
    table[text] = weight.data_ptr() + tok[text].astype(np.int64) * row_bytes
    fbase = np.array([f.data_ptr() for f in feats] + [0], dtype=np.int64)
    vis = src_kind == -1
    table[vis] = fbase[src_feat[..., 0][vis]] + src_feat[..., 1][vis].astype(np.int64) * row_bytes
    table_d = torch.from_numpy(table).to(dev, non_blocking=False)
    new_input_embeds = torch.empty((B, Lmax, hidden), dtype=weight.dtype, device=dev)
    nv.splice_rows(table_d, new_input_embeds)
    new_labels = None
    new_mask = attention_mask
    if labels is not None or attention_mask is not None:
        map_d = torch.from_numpy(np.ascontiguousarray(src_kind)).to(dev)
        len_d = torch.from_numpy(new_len).to(dev)
        if labels is not None:
            new_labels = torch.empty((B, Lmax), dtype=labels.dtype, device=dev)
            if labels.dtype != torch.int64:
                raise ValueError("splice: labels must be torch.long")
        if attention_mask is not None:
            if attention_mask.dtype not in (torch.bool, torch.int64):
                raise ValueError("splice: attention_mask must be torch.bool or torch.long")
            new_mask = torch.empty((B, Lmax), dtype=attention_mask.dtype, device=dev)
        nv.splice_labels(labels.contiguous() if labels is not None else None,
                         attention_mask.contiguous() if attention_mask is not None else None, map_d, len_d, S, IGNORE_INDEX,
                         new_labels, new_mask if attention_mask is not None else None)
    for f in feats:
        del f
    return None, new_mask, past_key_values, new_input_embeds, new_labels
