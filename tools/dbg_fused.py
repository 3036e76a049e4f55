"""Dev tool: per-window error of the fused kernel's local contexts against hicom_local_attn_fwd."""
import math, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv, geometry as geo, synth
T, H, W, kt, ks = [int(x) for x in (sys.argv[1:6] or (8, 6, 6, 4, 3))]
E, R = 1152, 9
x = synth.synth_inputs(T, H, W, E, tag="dbg")
bf = lambda a: torch.from_numpy(a).cuda().to(torch.bfloat16)
ff, fe, g = bf(x["ff"]), bf(x["fe"]), bf(x["g"])
qt = torch.from_numpy(synth.normal_like((R, E), 81, 0.05)).cuda()
qhi = torch.zeros((16, E), dtype=torch.bfloat16, device="cuda"); qlo = torch.zeros_like(qhi)
nv.split_bf16(qt, 16, qhi, qlo); qhi[R:] = g
axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (geo.axis_tiling(T, kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)))
nw = (T // kt) * (H // ks) * (W // ks)
ref = torch.empty((nw, E), device="cuda"); nv.local_attn(fe, ff, axes, g, 0, 1 / math.sqrt(E), 0.0, 0, ref)
for nparts in sorted({nv.fused_stream_nparts(nw), max(1, (nw + 15) // 16), min(nw, 3), nw}):
    wpw = (nw + nparts - 1) // nparts
    if (nparts - 1) * wpw >= nw: continue
    per_t = (H // ks) * (W // ks)
    if ((wpw + per_t - 2) // per_t + 1) * kt > 8 or wpw > 32: continue
    pm, pl, pa = torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, E, device="cuda")
    ctx = torch.full((nw, E), float("nan"), device="cuda")
    nv.fused_stream(ff, fe, kt, ks, qhi, qlo, R, 1 / math.sqrt(E), 0.0, None, None, None, 0, T, T + H, pm, pl, pa, ctx)
    torch.cuda.synchronize()
    err = (ctx - ref).abs().amax(dim=1).cpu().numpy()
    print("nparts", nparts, "wpw", wpw, "per-window max err:", np.array2string(err, precision=3))
    bad = int(np.nanargmax(np.where(np.isnan(err), np.inf, err)))
    e = (ctx[bad] - ref[bad]).abs().cpu().numpy()
    print("   worst window", bad, "nan", int(np.isnan(e).sum()), "err by 144-slice:", np.array2string(e.reshape(8, 144).max(1), precision=3),
          "ratio ctx/ref median", float((ctx[bad] / ref[bad]).median()))
