"""Dev tool: phase timeline of fused_stream_kernel from in-kernel s_memtime stamps.

Builds a second library with -DHICOM_TRACE (never the product one), runs the C2 shape, and prints for every
phase the distribution over workgroups.  Usage on the GPU box:  python tools/fused_trace.py
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_trace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    subprocess.check_call([bn.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           "-DHICOM_TRACE", "-o", LIB, *bn.sources()])
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
marg = torch.empty(nparts, 9, wpw, 12, device=dev)
pm, pl, pacc = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, E, device=dev)
chi = torch.empty(nw, E, device=dev, dtype=torch.bfloat16); clo = torch.empty_like(chi)
run = lambda: nv.fused_stream(ff, fe, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, pos_a, 0, T, T + H, pm, pl, pacc, marg, None, chi, clo)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
print("kernel (events, incl. launch) %.1f us, nparts %d, windows/wg %d" % (a.elapsed_time(b) * 1e3, nparts, wpw))
buf = np.zeros(1024 * 128, dtype=np.uint64)
L = nv.lib()
L.hicom_debug_fused_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.hicom_debug_fused_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(1024, 128)[:nparts].astype(np.int64)
ntile = (wpw * 36 + 15) // 16
n = 3 + 5 * ntile + 2
t0 = tr[:, 0].min()
span = tr[:, n - 1].max() - t0
print("stamps per wg %d (tiles %d); span first start -> last end: %d ticks" % (n, ntile, span))
tick_us = float(os.environ.get("TICK_US", 1.0))      # printed in raw s_memtime ticks (~core clock cycles; not synchronised across CUs)
def stat(name, d):
    print("%-34s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f ticks" % (name, d.mean() * tick_us, np.percentile(d, 10) * tick_us,
          np.percentile(d, 50) * tick_us, np.percentile(d, 90) * tick_us, d.max() * tick_us))
full = tr[(tr[:, n - 1] > 0)]
stat("wg start (after first wg)", full[:, 0] - t0)
stat("prologue: issue requests", full[:, 1] - full[:, 0])
stat("prologue: tables", full[:, 2] - full[:, 1])
for t in range(ntile):
    base = 3 + 5 * t
    prev = full[:, base - 1]
    stat("tile %d wait data  (-> [A])" % t, full[:, base] - prev)
    stat("tile %d scores     ([A]->[B])" % t, full[:, base + 1] - full[:, base])
    stat("tile %d softmax    ([B]->[C])" % t, full[:, base + 2] - full[:, base + 1])
    stat("tile %d P.x        ([C]->   )" % t, full[:, base + 3] - full[:, base + 2])
    stat("tile %d completion          " % t, full[:, base + 4] - full[:, base + 3])
stat("final drain", full[:, n - 2] - full[:, n - 3])
stat("epilogue issue", full[:, n - 1] - full[:, n - 2])
stat("wg total", full[:, n - 1] - full[:, 0])
stat("wg end (after first start)", full[:, n - 1] - t0)
