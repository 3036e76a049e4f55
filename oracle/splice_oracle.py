"""CPU restatement of the reference's splice of visual tokens into the LLM input embeddings
(`HIComMetaForCausalLM.prepare_inputs_labels_for_multimodal`, hicom/model/hicom_arch.py:283-372, i.e. the part behind
`mm_features = self.encode_images_or_videos(...)`).  TEST INFRASTRUCTURE: imported by tests/ only.

Plain loops over samples and segments, each step citing the reference line it follows.  Pinned by
tests/golden/golden_splice_v1.npz (made by tests/golden/make_golden_splice.py, which runs the reference's own method on a
stub model)."""
from __future__ import annotations

import torch

IGNORE_INDEX = -100                                  # hicom/constants.py:7
MM_TOKENS = (-200, -201, -202)                        # hicom/constants.py:30-34 MODAL_INDEX_MAP values


def splice(embed_weight, input_ids, attention_mask, labels, mm_features):
    new_embeds, new_labels = [], ([] if labels is not None else None)
    cur = 0
    for b, ids in enumerate(input_ids):                                            # :287
        is_mm = torch.zeros_like(ids, dtype=torch.bool)
        for t in MM_TOKENS:
            is_mm |= ids == t
        if int(is_mm.sum()) == 0:                                                  # :290 pure text: takes mm_features[cur][0:0]
            _ = mm_features[cur]
            new_embeds.append(embed_weight[ids])                                   # :292-295
            if labels is not None:
                new_labels.append(labels[b])                                       # :297-298
            cur += 1                                                               # :299
            continue
        segs, lsegs = [], []
        cur_ids, cur_labels = ids, (labels[b] if labels is not None else None)
        pos = torch.where(is_mm)[0]
        while pos.numel() > 0:                                                     # :309
            f = mm_features[cur]
            start = int(pos[0])
            segs += [embed_weight[cur_ids[:start]], f]                             # :313-314
            if labels is not None:
                lsegs += [cur_labels[:start], torch.full((f.shape[0],), IGNORE_INDEX, dtype=labels.dtype)]   # :316-317
                cur_labels = cur_labels[start + 1:]                                # :318
            cur += 1                                                               # :320
            cur_ids = cur_ids[start + 1:]                                          # :321
            m = torch.zeros_like(cur_ids, dtype=torch.bool)
            for t in MM_TOKENS:
                m |= cur_ids == t
            pos = torch.where(m)[0]                                                # :322
        if cur_ids.numel() > 0:                                                    # :324
            segs.append(embed_weight[cur_ids])
            if labels is not None:
                lsegs.append(cur_labels)
        new_embeds.append(torch.cat(segs, dim=0))                                  # :330-331
        if labels is not None:
            new_labels.append(torch.cat(lsegs, dim=0))                             # :333-334
    lens = [e.shape[0] for e in new_embeds]
    Lmax = max(lens)
    S = input_ids.shape[1]
    if any(l != lens[0] for l in lens):                                            # :337 ragged: right-pad
        out = torch.stack([torch.cat([e, torch.zeros((Lmax - e.shape[0], e.shape[1]), dtype=e.dtype)]) for e in new_embeds])   # :341-344
        out_labels = None
        if labels is not None:
            out_labels = torch.stack([torch.cat([l, torch.full((Lmax - l.shape[0],), IGNORE_INDEX, dtype=l.dtype)]) for l in new_labels])  # :346-352
        out_mask = attention_mask
        if attention_mask is not None:                                             # :354-363 (needs labels in the reference)
            rows = []
            for mrow, l in zip(attention_mask, lens):
                rows.append(torch.cat([torch.full((l - S,), True, dtype=attention_mask.dtype), mrow,
                                       torch.full((Lmax - l,), False, dtype=attention_mask.dtype)]))
            out_mask = torch.stack(rows)
    else:
        out = torch.stack(new_embeds)                                              # :365
        out_labels = torch.stack(new_labels) if labels is not None else None       # :366-367
        out_mask = attention_mask
        if attention_mask is not None:                                             # :369-372
            out_mask = torch.cat([torch.full((attention_mask.shape[0], Lmax - S), True, dtype=attention_mask.dtype), attention_mask], dim=1)
    return out_mask, out, out_labels
