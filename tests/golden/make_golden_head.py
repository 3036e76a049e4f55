"""Generates tests/golden/golden_head_v1.npz: the SigLIP head projection of reference hicom/model/encoder.py:284-286,

    image_embeds = head.layernorm(x);  image_embeds = x + head.mlp(image_embeds)

evaluated in float32 on HF transformers' SiglipMultiheadAttentionPoolingHead (the module the reference calls; the
so400m dimensions 1152 / 4304 / gelu_pytorch_tanh / eps 1e-6 are HF config values) with synthetic weights and tokens from
hicom_amd.synth.  Build container only:  python tests/golden/make_golden_head.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hicom_amd import synth   # noqa: E402

D, INTER, TOKENS = 1152, 4304, 64


def head_state_dict(tag="head"):
    shapes = {"head.layernorm.weight": (D,), "head.layernorm.bias": (D,), "head.mlp.fc1.weight": (INTER, D), "head.mlp.fc1.bias": (INTER,),
              "head.mlp.fc2.weight": (D, INTER), "head.mlp.fc2.bias": (D,)}
    sd = {}
    for k, shp in shapes.items():
        s = synth.seed_of(tag + ":" + k)
        if k.endswith("layernorm.weight"):
            sd[k] = synth.round_to_bf16(1.0 + synth.normal_like(shp, s, 0.1))
        elif k.endswith("bias"):
            sd[k] = synth.normal_like(shp, s, 0.05)
        else:
            sd[k] = synth.normal_like(shp, s, 0.02)
    return sd


def tokens(tag="head"):
    # SigLIP's last_hidden_state is post-layernorm output: O(1) values with a few large channels
    x = synth.normal_like((TOKENS, D), synth.seed_of(tag + ":x"), 1.0)
    x[:, 7] *= 12.0
    return synth.round_to_bf16(x)


def sample_positions(n, count=512):
    k = np.arange(count, dtype=np.int64)
    return (k * 2654435761 + 12345) % n


def cotangent(tag="head"):
    return synth.normal_like((TOKENS, D), synth.seed_of(tag + ":cotangent"), 1.0)


def main():
    from transformers import SiglipVisionConfig
    from transformers.models.siglip.modeling_siglip import SiglipMultiheadAttentionPoolingHead
    cfg = SiglipVisionConfig(hidden_size=D, intermediate_size=INTER, num_attention_heads=16, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    head = SiglipMultiheadAttentionPoolingHead(cfg).float().eval()
    sd = head_state_dict()
    missing, unexpected = head.load_state_dict({k[len("head."):]: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(m.startswith(("probe", "attention")) for m in missing), (missing, unexpected)
    x = torch.from_numpy(tokens())
    with torch.no_grad():
        image_embeds = head.layernorm(x)                     # encoder.py:284
        image_embeds = x + head.mlp(image_embeds)            # encoder.py:285
    blobs = {"out": image_embeds.numpy().astype(np.float32)}
    # stage 3 of the reference's script trains "vision_model_head" (train.py:717-720; release scripts :175): gradients of the six
    # parameters those two lines touch, for loss = sum(out * R), by the reference modules' own fp32 autograd
    head.train()
    for p_ in head.parameters():
        p_.requires_grad_(True)
    out = x + head.mlp(head.layernorm(x))
    R = torch.from_numpy(cotangent())
    (out * R).sum().backward()
    for k in sd:
        mod = head
        for part in k[len("head."):].split("."):
            mod = getattr(mod, part)
        v = mod.grad.numpy().astype(np.float32).reshape(-1)            # 512 sampled entries + (sum, abs-sum, max) per parameter
        blobs["grad/" + k + "/samples"] = v[sample_positions(v.size)]
        blobs["grad/" + k + "/sums"] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum(), np.abs(v).max()])
    path = os.path.join(HERE, "golden_head_v1.npz")
    np.savez_compressed(path, **blobs)
    print("wrote", path, os.path.getsize(path), "bytes; absmax", float(image_embeds.abs().max()))


if __name__ == "__main__":
    main()
