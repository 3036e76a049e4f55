"""Dev tool: phase timeline of fused_ring_kernel from in-kernel s_memtime stamps.

Builds a second library with -DHICOM_TRACE (never the product one), runs the C2 shape, and prints for every
phase the distribution over workgroups (ticks = shader cycles; clocks of different CUs are not synchronised).
Usage on the GPU box:  python tools/fused_trace.py
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_trace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    bn.build(extra_flags=("-DHICOM_TRACE",), lib_path=LIB, verbose=False)
    os.environ["HICOM_NATIVE_LIB"] = LIB
    sys.exit(subprocess.call([sys.executable, *sys.argv]))
import numpy as np, torch
from hicom_amd import native as nv
dev = "cuda"
T, H, W, E = int(os.environ.get("T", 64)), 27, 27, 1152
ff = torch.randn(T, H, W, E, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(E, device=dev).bfloat16()
qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (qhi.float() * 1e-3).bfloat16(); qhi[9:] = g; qlo[9:] = 0
pos_a = torch.randn(16, T + 54, device=dev) * 0.1
nw = (T // 4) * 81
nparts = nv.fused_stream_nparts(nw)
wpw = (nw + nparts - 1) // nparts
pe = torch.randn(T + 54, E, device=dev); pe_hi = pe.bfloat16(); pe_lo = (pe - pe_hi.float()).bfloat16()
pm, pl, pacc = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, E, device=dev)
chi = torch.empty(nw, E, device=dev, dtype=torch.bfloat16); clo = torch.empty_like(chi)
run = lambda: nv.fused_stream(ff, fe, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, T, T + H, pm, pl, pacc, None, chi, clo)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
print("kernel (events, incl. launch) %.1f us, nparts %d, windows/wg %d" % (a.elapsed_time(b) * 1e3, nparts, wpw))
buf = np.zeros(1024 * 3 * 256, dtype=np.uint64)
L = nv.lib()
L.hicom_debug_fused_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.hicom_debug_fused_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(1024, 3, 256)[:nparts].astype(np.int64)
ntile = (wpw * 36 + 15) // 16
def stat(name, d):
    print("%-40s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (name, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max()))
for who, label in ((0, "compute wave 0 (marginals)"), (1, "compute wave 2")):
    c = tr[:, who, who::2]          # both stamp macros advance the slot counter: wave 0 owns the even slots, wave 2 the odd ones
    print("---- %s: 1 + 5 stamps per tile" % label)
    names = ["wait at [A]", "reads+scores ([A]->arrive B)", "wait at [B]", "softmax+marg+P.x", "completion -> arrive next [A]"]
    agg = {n: [] for n in names}
    for t in range(ntile):
        b0 = 1 + 5 * t
        seg = [c[:, b0 + 1] - c[:, b0], c[:, b0 + 2] - c[:, b0 + 1], c[:, b0 + 3] - c[:, b0 + 2], c[:, b0 + 4] - c[:, b0 + 3]]
        if t + 1 < ntile:
            seg.append(c[:, b0 + 5] - c[:, b0 + 4])
        for n, d in zip(names, seg):
            agg[n].append(d)
    for n in names:
        stat(n + " (all tiles)", np.concatenate(agg[n]))
    stat("tile 0: wait at [A] (first data)", c[:, 2] - c[:, 1])
    stat("loop total (first [A] arrive -> last P.x)", c[:, 1 + 5 * (ntile - 1) + 4] - c[:, 1])
c0 = tr[:, 0, 0::2]
e = 1 + 5 * ntile
stat("PROLOGUE: kernel entry -> tables / operands done (wave 0, before [P])", c0[:, 0] - tr[:, 0, 2 * (e + 2)])
stat("PROLOGUE: kernel entry -> loader 0 has issued its prologue requests", tr[:, 2, 0] - tr[:, 0, 2 * (e + 2)])
for idx, what in ((241, "operand loads issued"), (242, "window tables written"), (243, "pos-emb slot tables built"), (244, "score-side pos table staged")):
    stat("PROLOGUE: wave 0 entry -> " + what, tr[:, 0, idx] - tr[:, 0, 2 * (e + 2)])
for l in range(4):
    stat("PROLOGUE: wave 0 entry -> first instruction of loader %d (wave %d)" % (l, 8 + l), tr[:, 2, 240 + l] - tr[:, 0, 2 * (e + 2)])
stat("PROLOGUE: kernel entry -> first [A] arrive (wave 0)", c0[:, 1] - tr[:, 0, 2 * (e + 2)])
stat("PROLOGUE: kernel entry -> tile 0 data (past [A])", c0[:, 2] - tr[:, 0, 2 * (e + 2)])
stat("TAIL: last P.x issued -> pos-emb tiles done", c0[:, e] - c0[:, e - 1])
stat("TAIL: pos-emb done -> state stores issued", c0[:, e + 1] - c0[:, e])
stat("WHOLE: entry -> state stores issued", c0[:, e + 1] - tr[:, 0, 2 * (e + 2)])
c = tr[:, 2]
print("---- loader 0: 1 + 6 stamps per tile")
names = ["addr -> landed (vmcnt wait)", "wait at [A]", "issue ff(t+1)", "wait at [B]", "issue fe(t+2)", "next addr math"]
agg = {n: [] for n in names}
for t in range(ntile):
    b0 = 1 + 6 * t
    seg = [c[:, b0 + 1] - c[:, b0], c[:, b0 + 2] - c[:, b0 + 1], c[:, b0 + 3] - c[:, b0 + 2], c[:, b0 + 4] - c[:, b0 + 3], c[:, b0 + 5] - c[:, b0 + 4]]
    if t + 1 < ntile:
        seg.append(c[:, b0 + 6] - c[:, b0 + 5])
    for n, d in zip(names, seg):
        agg[n].append(d)
for n in names:
    stat(n + " (all tiles)", np.concatenate(agg[n]))
stat("tile 0: vmcnt wait (first data)", c[:, 2] - c[:, 1])
stat("loader total", c[:, 1 + 6 * (ntile - 1) + 5] - c[:, 0])

print("---- loader 0, per tile (incl. the pe tiles that follow the %d token tiles): landed-wait | wait[A] | issue ff | wait[B] | issue fe" % ntile)
for t in range(ntile - 3, ntile + 3):
    b0 = 1 + 6 * t
    if (c[:, b0 + 5] <= 0).all():
        break
    seg = [c[:, b0 + 1] - c[:, b0], c[:, b0 + 2] - c[:, b0 + 1], c[:, b0 + 3] - c[:, b0 + 2], c[:, b0 + 4] - c[:, b0 + 3], c[:, b0 + 5] - c[:, b0 + 4]]
    print("tile %2d: " % t + "  ".join("%6.0f" % np.median(x) for x in seg) + ("   next-addr %6.0f" % np.median(c[:, b0 + 6] - c[:, b0 + 5]) if (c[:, b0 + 6] > 0).all() else ""))

# wall-clock view (s_memrealtime, 100 MHz, chip-wide): when does each workgroup start and finish?
rt0, rt1, xcc = tr[:, 0, 250], tr[:, 0, 251], tr[:, 0, 252]
base = rt0.min()
st, en = (rt0 - base) / 100.0, (rt1 - base) / 100.0
print("---- wall clock (us from the first workgroup's entry)")
print("entry : p50 %.2f  p90 %.2f  max %.2f" % (np.percentile(st, 50), np.percentile(st, 90), st.max()))
print("exit  : min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f   (kernel ends with the LAST: max - p50 = %.2f us)" % (en.min(), np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max(), en.max() - np.percentile(en, 50)))
dur = en - st
print("duration per workgroup: p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), dur.max()))
for x in range(8):
    sel = xcc == x
    if sel.any():
        print("  XCC %d: %3d workgroups, exit p50 %.2f max %.2f" % (x, sel.sum(), np.percentile(en[sel], 50), en[sel].max()))
late = np.argsort(en)[-8:]
print("latest workgroups:", [(int(i), round(float(en[i]), 2), int(xcc[i])) for i in late])

# is the per-XCC pattern stable from launch to launch?  five more launches (rotating nothing: the same inputs), per-XCC median exit time
for rep in range(5):
    run(); torch.cuda.synchronize()
    assert L.hicom_debug_fused_trace(buf.ctypes.data, buf.nbytes) == 0
    t2 = buf.reshape(1024, 3, 256)[:nparts].astype(np.int64)
    r0, r1, xc = t2[:, 0, 250], t2[:, 0, 251], t2[:, 0, 252]
    e2 = (r1 - r0.min()) / 100.0
    print("launch %d: per-XCC median exit" % rep, " ".join("%d:%.1f" % (x, np.percentile(e2[xc == x], 50)) for x in range(8) if (xc == x).any()), " last %.1f" % e2.max(),
          " blocks->xcc of blocks 0..7:", [int(xc[i]) for i in range(8)])
