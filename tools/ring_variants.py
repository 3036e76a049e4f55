"""Dev tool (round 3): the ring kernel at the benchmark shape, frames_embed stream vs precomputed local logits (HIP events,
3 rotating input sets)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hicom_amd import native as nv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 27; E = 1152; R = 9; kt, ks = 4, 3
dev = "cuda"
sets = [(torch.randn(T, H, W, E, device=dev).bfloat16(), torch.randn(T, H, W, E, device=dev).bfloat16()) for _ in range(3)]
g = torch.randn(E, device=dev).bfloat16()
llogs = [(fe.view(-1, E).float() @ g.float()).contiguous() for _, fe in sets]
nw = (T // kt) * 81
qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (torch.randn(16, E, device=dev) * 1e-4).bfloat16()
qhi[R:] = g; qlo[R:] = 0
pos_a = torch.randn(16, T + H + W, device=dev) * 0.1
pe = torch.randn(T + H + W, E, device=dev); pe_hi = pe.bfloat16(); pe_lo = (pe - pe_hi.float()).bfloat16()
nparts = nv.fused_stream_nparts(nw)
pm, pl, pa = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, E, device=dev)
c16 = torch.empty(nw, E, device=dev, dtype=torch.float16)
def run(i, variant):
    ff, fe = sets[i % 3]
    nv.fused_stream(ff, fe if variant == "fe" else None, kt, ks, qhi, qlo, R, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, T, T + H, pm, pl, pa, None,
                    ctx_f16=c16, local_logits=llogs[i % 3] if variant == "llog" else None)
for variant in ("fe", "llog", "fe", "llog"):
    for i in range(20): run(i, variant)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200): run(i, variant)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 200 * 1e3
    byt = T * 729 * E * 2 * (2 if variant == "fe" else 1)
    print(f"T={T} {variant:4s}: {us:7.2f} us per launch, {byt / us / 1e6:6.2f} TB/s of token bytes")
