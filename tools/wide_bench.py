"""Dev tool: the wide global stream kernel alone (288 folded rows, 64 x 27 x 27 tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hicom_amd import native as nv
T, H, W, E, R = 64, 27, 27, 1152, 288
N = T * H * W
g = torch.Generator(device="cuda").manual_seed(1)
ff = torch.randn(N, E, device="cuda", generator=g).to(torch.bfloat16)
qhi = (torch.randn(R, E, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
qlo = (torch.randn(R, E, device="cuda", generator=g) * 1e-4).to(torch.bfloat16)
pos_a = torch.randn(R, 64 + H + W, device="cuda", generator=g) * 0.1
nparts = nv.global_stream_nparts(N, R)
scores = torch.empty(R, (N + 15) // 16 * 16, device="cuda")
pm, pl = torch.empty(nparts, R, device="cuda"), torch.empty(nparts, R, device="cuda")
pacc = torch.empty(nparts, R, E, device="cuda")
pmarg = torch.empty(nparts, R, nv.global_stream_marg_width(H, W), device="cuda")
forms = {
    "logits written (marginals taken from them afterwards)": lambda: nv.global_stream(ff, N, qhi, qlo, pos_a, H, W, 0, 64, 64 + H, scores, pm, pl, pacc, rows=R),
    "marginals in the kernel + logits written": lambda: nv.global_stream_marg(ff, N, qhi, qlo, pos_a, H, W, 0, 64, 64 + H, scores, pm, pl, pacc, pmarg, rows=R),
    "marginals in the kernel, no logit tensor": lambda: nv.global_stream_marg(ff, N, qhi, qlo, pos_a, H, W, 0, 64, 64 + H, None, pm, pl, pacc, pmarg, rows=R),
}
flops = 4.0 * R * N * E
for name, f in forms.items():
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: nparts {nparts}, {dt * 1e6:.1f} us, {flops / dt / 1e12:.0f} TFLOP/s useful (hi/lo MFMAs: x1.5), checksum {float(pacc.double().sum()):.6e}")
