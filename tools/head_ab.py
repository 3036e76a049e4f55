"""Dev tool: head projection time (ln_stream + 2 dense16 GEMMs) with and without the padded hidden pitch, one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hicom_amd import encoder, native as nv

D, inter = 1152, 4304
head = torch.nn.Module()
head.layernorm = torch.nn.LayerNorm(D, eps=1e-6)
head.mlp = torch.nn.Module()
head.mlp.fc1, head.mlp.fc2 = torch.nn.Linear(D, inter), torch.nn.Linear(inter, D)
head = head.to(torch.bfloat16).cuda()
x = torch.randn(64, 729, D, device="cuda").to(torch.bfloat16)

def t(n=20):
    with torch.no_grad():
        for _ in range(5):
            encoder.siglip_head_embed(x, head)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            encoder.siglip_head_embed(x, head)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

orig = encoder._head_cache
for rep in range(3):
    for pad in (0, 192):
        head.__dict__.pop("_hicom_f16", None)
        def patched(h, pad=pad):
            fc1, fc2 = h.mlp.fc1, h.mlp.fc2
            hit = h.__dict__.get("_hicom_f16")
            if hit is None:
                kpad = 4352
                hit = (None, nv.f16_weight_copy(fc1.weight), nv.f16_weight_copy(fc2.weight, kpad + pad), kpad, kpad + pad)
                h.__dict__["_hicom_f16"] = hit
            return hit[1], hit[2], hit[3], hit[4]
        encoder._head_cache = patched
        print(f"pad {pad}: {t():.3f} ms")
