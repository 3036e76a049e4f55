"""Generates tests/golden/golden_splice_v1.npz by running the REFERENCE's own
`HIComMetaForCausalLM.prepare_inputs_labels_for_multimodal` (hicom/model/hicom_arch.py:271-373, imported in place through
oracle/ref_shim.py) on a stub model: an nn.Embedding as `get_model().embed_tokens` and a fixed list of "compressed
token" tensors as the result of `encode_images_or_videos`.  Build container only:  python tests/golden/make_golden_splice.py"""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hicom_amd import synth            # noqa: E402

VOCAB, HIDDEN = 40, 16
IMG, VID = -200, -201

# name -> (input_ids rows, feature row counts, with_labels, mask dtype or None)
SPLICE_CASES = {
    "S1_video_and_text_ragged": ([[3, 4, VID, 5, 6, 7, 8], [9, 10, 11, 12, 13, 14, 15]], [5, 3], True, "long"),
    "S2_single_video_equal": ([[1, 2, VID, 3]], [6], True, "bool"),
    "S3_image_and_video_one_sample": ([[IMG, 4, 5, VID, 6], [7, 8, IMG, 9, 10]], [2, 4, 7], True, "long"),
    "S4_generate_no_labels": ([[1, VID, 2, 3, 4]], [5], False, "bool"),
    "S5_placeholder_first_and_last": ([[VID, 2, 3], [4, 5, IMG]], [3, 3], True, "long"),
    "S6_no_mask_no_labels": ([[1, 2, VID]], [4], False, None),
}


def build(name):
    rows, nfeat, with_labels, mdt = SPLICE_CASES[name]
    ids = torch.tensor(rows, dtype=torch.long)
    B, S = ids.shape
    weight = torch.from_numpy(synth.normal_like((VOCAB, HIDDEN), synth.seed_of(name + ":emb")))
    feats = [torch.from_numpy(synth.normal_like((n, HIDDEN), synth.seed_of(f"{name}:feat{k}"))) for k, n in enumerate(nfeat)]
    labels = None
    if with_labels:
        labels = torch.where(ids >= 0, ids + 100, torch.full_like(ids, -100))
        labels[:, 0] = -100
    mask = None
    if mdt is not None:
        mask = torch.ones((B, S), dtype=torch.bool if mdt == "bool" else torch.long)
        mask[-1, -1] = 0
    return ids, weight, feats, labels, mask


def main():
    from oracle import ref_shim
    ref_shim.load()
    arch = importlib.import_module("hicom.model.hicom_arch")

    class Stub(arch.HIComMetaForCausalLM):
        def __init__(self, weight, feats):
            self._emb = torch.nn.Embedding(VOCAB, HIDDEN)
            self._emb.weight.data.copy_(weight)
            self._feats = feats

        def get_model(self):
            return SimpleNamespace(embed_tokens=self._emb)

        def get_vision_tower(self):
            return object()

        def encode_images_or_videos(self, images, guided_input=None):
            return self._feats

        @property
        def device(self):
            return torch.device("cpu")

    blobs = {}
    for name in SPLICE_CASES:
        ids, weight, feats, labels, mask = build(name)
        with torch.no_grad():
            r_ids, r_mask, _, r_emb, r_lab = Stub(weight, feats).prepare_inputs_labels_for_multimodal(ids, mask, None, labels, images=[0])
        assert r_ids is None
        blobs[name + "/embeds"] = r_emb.numpy().astype(np.float32)
        if r_lab is not None:
            blobs[name + "/labels"] = r_lab.numpy()
        if r_mask is not None:
            blobs[name + "/mask"] = r_mask.numpy().astype(np.int64)
        print(name, tuple(r_emb.shape), None if r_lab is None else tuple(r_lab.shape), None if r_mask is None else tuple(r_mask.shape))
    path = os.path.join(HERE, "golden_splice_v1.npz")
    np.savez_compressed(path, **blobs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
