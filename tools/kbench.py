"""Dev tool: per-kernel timings (HIP events, back-to-back launches) at the C2 shape."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv, geometry as geo
dev = "cuda"
T, H, W, E, HID = 64, 27, 27, 1152, 896
ff = torch.randn(T, H, W, E, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(E, device=dev).bfloat16()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
which = sys.argv[1:] or ["gemm", "local", "stream", "merge", "lin"]
if "gemm" in which:
    x = torch.randn(1296, E, device=dev); w0 = torch.randn(HID, E, device=dev).bfloat16() * 0.02; b0 = torch.zeros(HID, device=dev).bfloat16()
    w2 = torch.randn(HID, HID, device=dev).bfloat16() * 0.02
    hid = torch.empty(1296, HID, device=dev); out = torch.empty(1296, HID, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: nv.readout_gemm(x, w0, b0, hid, act=1)); t2 = timeit(lambda: nv.readout_gemm(hid, w2, b0, out))
    print("gemm1 %.1f us (%.0f TF incl hi/lo)  gemm2 %.1f us" % (t1, 2 * 2 * 1296 * HID * E / t1 / 1e6, t2))
if "local" in which:
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (geo.axis_tiling(T, 4), geo.axis_tiling(H, 3), geo.axis_tiling(W, 3)))
    ctx = torch.empty(1296, E, device=dev)
    t = timeit(lambda: nv.local_attn(fe, ff, axes, g, 0, 1 / math.sqrt(E), 0.0, 0, ctx))
    print("local_attn %.1f us  %.2f TB/s" % (t, 2 * ff.numel() * 2 / t / 1e6))
if "stream" in which:
    N = T * H * W
    qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (qhi.float() * 1e-3).bfloat16(); qhi[9:] = 0; qlo[9:] = 0
    pos_a = torch.randn(16, 64 + 54, device=dev) * 0.1
    for nparts in (256, 512, 768):
        scores = torch.empty(16, (N + 15) // 16 * 16, device=dev)
        pm, pl, pacc = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, E, device=dev)
        t = timeit(lambda: nv.global_stream(ff, N, qhi, qlo, pos_a, H, W, 0, 64, 64 + H, scores, pm, pl, pacc, rows=9))
        print("global_stream nparts=%d %.1f us  %.2f TB/s" % (nparts, t, ff.numel() * 2 / t / 1e6))

if "fused" in which or len(sys.argv) == 1:
    N = T * H * W
    qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (qhi.float() * 1e-3).bfloat16(); qhi[9:] = g; qlo[9:] = 0
    pos_a = torch.randn(16, 64 + 54, device=dev) * 0.1
    nw = 1296
    for nparts in sorted({nv.fused_stream_nparts(nw)}):
        pe = torch.randn(64 + 54, E, device=dev); pe_hi = pe.bfloat16(); pe_lo = (pe - pe_hi.float()).bfloat16()
        pm, pl, pacc = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, E, device=dev)
        ctx = torch.empty(nw, E, device=dev)
        t = timeit(lambda: nv.fused_stream(ff, fe, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, 64, 64 + H, pm, pl, pacc, ctx))
        print("fused_stream nparts=%d %.1f us  %.2f TB/s (inputs only)" % (nparts, t, 2 * ff.numel() * 2 / t / 1e6))
        t = timeit(lambda: nv.fused_stream(ff, fe, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, None, None, None, 0, 64, 64 + H, pm, pl, pacc, ctx))
        print("fused_stream (no pos) nparts=%d %.1f us" % (nparts, t))

if "pgemm" in which or len(sys.argv) == 1:
    x = torch.randn(1296, E, device=dev); w0 = (torch.randn(HID, E, device=dev) * 0.02).bfloat16(); b0 = torch.zeros(HID, device=dev).bfloat16()
    w2 = (torch.randn(HID, HID, device=dev) * 0.02).bfloat16()
    ah = torch.empty(1296, E, device=dev, dtype=torch.bfloat16); al = torch.empty_like(ah); nv.split_bf16(x, 1296, ah, al)
    hh = torch.empty(1296, HID, device=dev, dtype=torch.bfloat16); hl = torch.empty_like(hh); out = torch.empty(1296, HID, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: nv.planes_gemm(ah, al, w0, b0, act=1, out_hi=hh, out_lo=hl)); t2 = timeit(lambda: nv.planes_gemm(hh, hl, w2, b0, y=out))
    print("planes gemm1 %.1f us (%.0f TF incl hi/lo)  gemm2 %.1f us" % (t1, 2 * 2 * 1296 * HID * E / t1 / 1e6, t2))

if "r16" in which:
    x = torch.randn(1296, E, device=dev); w0 = (torch.randn(HID, E, device=dev) * 0.02).bfloat16(); b0 = torch.zeros(HID, device=dev).bfloat16()
    w2 = (torch.randn(HID, HID, device=dev) * 0.02).bfloat16()
    a16 = nv.to_f16(x); w0_16 = nv.to_f16(w0); w2_16 = nv.to_f16(w2)
    h16 = torch.empty(1296, HID, device=dev, dtype=torch.float16); out = torch.empty(1296, HID, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: nv.readout16_gemm(a16, w0_16, b0, act=1, out_f16=h16)); t2 = timeit(lambda: nv.readout16_gemm(h16, w2_16, b0, y=out))
    print("readout16 gemm1 %.1f us (%.0f TF)  gemm2 %.1f us" % (t1, 2 * 1296 * HID * E / t1 / 1e6, t2))
    nparts = 216
    pm, pl, pacc = torch.randn(nparts, 16, device=dev), torch.rand(nparts, 16, device=dev) + 0.5, torch.randn(nparts, 16, E, device=dev)
    wv = (torch.randn(E, E, device=dev) * 0.02).bfloat16(); bv = torch.zeros(E, device=dev).bfloat16()
    po = torch.empty(E // 64, E, device=dev); pre = torch.empty(E, device=dev); hid = torch.empty(HID, device=dev)
    t = timeit(lambda: nv.merge_vproj(pm, pl, pacc, 9, wv, po))
    print("merge_vproj %.1f us" % t)
    gw0 = (torch.randn(HID, E, device=dev) * 0.02).bfloat16()
    t1 = timeit(lambda: nv.readout16_gemm(a16, w0_16, b0, act=1, out_f16=h16, aux=dict(xs=po, xb=bv, w=wv, b=bv, res=bv, y=pre)))
    t2 = timeit(lambda: nv.readout16_gemm(h16, w2_16, b0, y=out, aux=dict(xs=pre.view(1, -1), w=gw0, b=b0, act=1, y=hid)))
    print("readout16 gemm1+aux(out_proj) %.1f us  gemm2+aux(readout0) %.1f us" % (t1, t2))
    tok = torch.empty(32, HID, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: nv.linear_to_rows(hid.view(1, -1), w2, b0, tok, 0, 32))
    print("linear_to_rows (M=1) %.1f us" % t)

if "aux" in which:
    # the GEMV role alone: a one-tile GEMM carries it
    a1 = nv.to_f16(torch.randn(96, 64, device=dev)); w1 = nv.to_f16(torch.randn(64, 64, device=dev)); o1 = torch.empty(96, 64, device=dev, dtype=torch.float16)
    wv = (torch.randn(E, E, device=dev) * 0.02).bfloat16(); bv = torch.zeros(E, device=dev).bfloat16()
    po = torch.randn(E // 64, E, device=dev); pre = torch.empty(E, device=dev); hid = torch.empty(HID, device=dev)
    gw0 = (torch.randn(HID, E, device=dev) * 0.02).bfloat16()
    t0 = timeit(lambda: nv.readout16_gemm(a1, w1, None, out_f16=o1))
    t1 = timeit(lambda: nv.readout16_gemm(a1, w1, None, out_f16=o1, aux=dict(xs=po, xb=bv, w=wv, b=bv, res=bv, y=pre)))
    t2 = timeit(lambda: nv.readout16_gemm(a1, w1, None, out_f16=o1, aux=dict(xs=po[:1], xb=bv, w=wv, b=bv, res=bv, y=pre)))
    t3 = timeit(lambda: nv.readout16_gemm(a1, w1, None, out_f16=o1, aux=dict(xs=pre.view(1, -1), w=gw0, act=1, y=hid)))
    print("one-tile gemm %.1f us | + aux out_proj (18 parts) %.1f | (1 part) %.1f | + aux readout0 %.1f" % (t0, t1, t2, t3))
    x1 = torch.randn(1, E, device=dev); y1 = torch.empty(1, E, device=dev)
    t4 = timeit(lambda: nv.linear(x1, wv, bv, y1))
    print("linear_rows M=1 1152x1152: %.1f us" % t4)
