"""Dev tool (round 3): ring kernel local contexts per window against torch, both variants (frames_embed stream / precomputed logits)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hicom_amd import native as nv, synth, geometry as geo
T, H, W, kt, ks = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 6, 6, 4, 3)))
E, R = 1152, 9
x = synth.synth_inputs(T, H, W, E, tag="dbgring")
bf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).cuda()
ff, fe, g = bf(x["ff"]), bf(x["fe"]), bf(x["g"])
N = T * H * W
nw = (T // kt) * (H // ks) * (W // ks)
qhi = torch.zeros((16, E), dtype=torch.bfloat16, device="cuda"); qlo = torch.zeros_like(qhi)
qhi[:R] = (torch.randn(R, E, device="cuda") * 0.05).to(torch.bfloat16)
qhi[R:] = g
# reference local contexts
idx = torch.arange(N, device="cuda").view(T // kt, kt, H // ks, ks, W // ks, ks).permute(0, 2, 4, 1, 3, 5).reshape(nw, -1)
llog = (fe.view(N, E).float() @ g.float())
logit = llog[idx] / math.sqrt(E)
p = torch.softmax(logit, -1)
ref = torch.einsum("wk,wke->we", p, ff.view(N, E).float()[idx])
for nparts in sorted({nw, max(1, nw // 3), nv.fused_stream_nparts(nw)}):
    for variant in ("fe", "llog"):
        pm, pl, pa = (torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, E, device="cuda"))
        ctx = torch.full((nw, E), float("nan"), device="cuda")
        nv.fused_stream(ff, fe if variant == "fe" else None, kt, ks, qhi, qlo, R, 1 / math.sqrt(E), 0.0, None, None, None, 0, T, T + H, pm, pl, pa, ctx,
                        local_logits=llog.contiguous() if variant == "llog" else None)
        torch.cuda.synchronize()
        err = (ctx - ref).abs().max(1).values
        print(f"nparts {nparts:3d} {variant:4s}: max err {float(err.max()):.3e}; bad windows {[int(i) for i in torch.nonzero(err > 1e-3).flatten()[:16]]}")
