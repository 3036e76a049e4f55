"""Dev tool: the SigLIP head projection's training step (stage 3 of the reference's script, train.py:717-720) at 64 x 729 tokens."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd.encoder import siglip_head_embed
dev = torch.device("cuda", 0)
D, inter = 1152, 4304
gen = torch.Generator(device=dev).manual_seed(99)
head = torch.nn.Module()
head.layernorm = torch.nn.LayerNorm(D, eps=1e-6)
head.mlp = torch.nn.Module()
head.mlp.fc1, head.mlp.fc2 = torch.nn.Linear(D, inter), torch.nn.Linear(inter, D)
head = head.to(torch.bfloat16).to(dev).train()
x = torch.randn(64, 729, D, device=dev, generator=gen).to(torch.bfloat16)
cot = torch.randn(x.shape, device=dev, generator=gen).to(torch.bfloat16)
def step():
    for p in head.parameters(): p.grad = None
    siglip_head_embed(x, head).backward(cot)
for keep in (True, False):
    head.keep_hidden_for_backward = keep
    for _ in range(3): step()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3)
    print(f"head training step, keep_hidden_for_backward={keep}: {min(ts) * 1e3:.2f} ms")
