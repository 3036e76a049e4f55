"""Dev tool: global context of the fused kernel (+ plain merge) against the two-kernel path, per row / channel block."""
import math, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv, geometry as geo, synth
T, H, W, kt, ks = [int(x) for x in (sys.argv[1:6] or (12, 9, 6, 4, 3))]
E, R = 1152, 9
x = synth.synth_inputs(T, H, W, E, tag="dbg")
bf = lambda a: torch.from_numpy(a).cuda().to(torch.bfloat16)
f32 = lambda s: torch.empty(s, dtype=torch.float32, device="cuda")
ff, fe, g = bf(x["ff"]), bf(x["fe"]), bf(x["g"])
N = T * H * W
qt = torch.from_numpy(synth.normal_like((R, E), 81, 0.05)).cuda()
qhi = torch.zeros((16, E), dtype=torch.bfloat16, device="cuda"); qlo = torch.zeros_like(qhi)
nv.split_bf16(qt, 16, qhi, qlo)
cap = T + 2
pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
pos_a = torch.zeros((16, pe.shape[0]), dtype=torch.float32, device="cuda"); nv.linear(qt, pe, None, pos_a, M=R)
pe_hi = torch.empty(pe.shape, dtype=torch.bfloat16, device="cuda"); pe_lo = torch.empty_like(pe_hi); nv.split_bf16(pe, pe.shape[0], pe_hi, pe_lo)
stride = (N + 15) // 16 * 16
np0 = nv.global_stream_nparts(N, 16)
s0 = f32((16, stride)); pm0, pl0, pa0 = f32((np0, 16)), f32((np0, 16)), f32((np0, 16, E))
nv.global_stream(ff, N, qhi, qlo, pos_a, H, W, 0, cap, cap + H, s0, pm0, pl0, pa0, rows=R)
ml, ref = f32((R, 2)), f32((R, E))
nv.global_merge(pm0, pl0, pa0, R, s0, N, H, W, pe, 0, cap, cap + H, f32((R * T * (H + W + 2),)), ml, ref, normalize=True)
# value-side pos part alone (for the error pattern): ref_nopos
ref_np = f32((R, E)); nv.global_merge(pm0, pl0, pa0, R, None, N, H, W, None, 0, 0, 0, None, ml, ref_np, normalize=True)
nw = (T // kt) * (H // ks) * (W // ks)
qhi_f = qhi.clone(); qhi_f[R:] = g
for nparts in sorted({nv.fused_stream_nparts(nw), min(nw, 3), nw}):
    wpw = (nw + nparts - 1) // nparts
    pm, pl, pa = f32((nparts, 16)), f32((nparts, 16)), f32((nparts, 16, E))
    ctx = f32((nw, E))
    nv.fused_stream(ff, fe, kt, ks, qhi_f, qlo, R, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, cap, cap + H, pm, pl, pa, ctx)
    acc = f32((R, E)); nv.global_merge(pm, pl, pa, R, None, N, H, W, None, 0, 0, 0, None, ml, acc, normalize=True)
    torch.cuda.synchronize()
    err = (acc - ref).abs()
    print("nparts", nparts, "wpw", wpw, "max err %.3e" % float(err.max()), " pos part magnitude %.3e" % float((ref - ref_np).abs().max()),
          " err/row", np.array2string(err.amax(1).cpu().numpy(), precision=2))
