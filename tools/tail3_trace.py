"""Dev tool: timeline of the fused tail launch (readout_tail_kernel) of a release step from in-kernel s_memrealtime stamps: tile workgroups
[0, 200) (GEMM 1 tile, then the same tile of GEMM 2), role workgroups [200, 254).  Builds a second library with -DHICOM_TRACE (never the product one).
Usage on the GPU box:  python tools/tail3_trace.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_trace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    bn.build(extra_flags=("-DHICOM_TRACE",), lib_path=LIB, verbose=False)
    os.environ["HICOM_NATIVE_LIB"] = LIB
    os.environ["HICOM_TAIL_LAUNCHES"] = "3"        # (the fused tail launch is opt-in)
    sys.exit(subprocess.call([sys.executable, *sys.argv]))
import numpy as np, torch
import bench
from hicom_amd import native as nv
dev = torch.device("cuda", 0)
m = bench.make_projector(bench.release_config(896, 64), dev)
sets = [(torch.randn(64, 27, 27, 1152, device=dev).bfloat16(), torch.randn(64, 27, 27, 1152, device=dev).bfloat16(), torch.randn(1152, device=dev).bfloat16()) for _ in range(3)]
L = nv.lib()
L.hicom_debug_r16_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
agg = []
with torch.no_grad():
    for i in range(300):
        a, b, g = sets[i % 3]
        m(a, b, g, "video", None)
    for rep in range(20):
        for i in range(7):
            a, b, g = sets[(rep + i) % 3]
            m(a, b, g, "video", None)
        torch.cuda.synchronize()
        buf = np.zeros(512 * 16, dtype=np.uint64)
        assert L.hicom_debug_r16_trace(buf.ctypes.data, buf.nbytes) == 0
        agg.append(buf.reshape(512, 16).astype(np.int64))
tr = np.stack(agg)                      # [rep, block, 16]
N1, NR = 200, 54
def show(name, x):
    x = x[np.isfinite(x)]
    if x.size == 0:
        return
    print("  %-44s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, x.min(), *np.percentile(x, [10, 50, 90]), x.max()))
t0 = np.where(tr[:, :, 0] > 0, tr[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]
def grp(title, sl, keys, live_key):
    print(title)
    live = tr[:, sl, live_key] > tr[:, sl, 0]
    for k, nm in keys:
        show(nm, ((tr[:, sl, k] - t0) / 100.0)[live])
T = slice(0, N1)
grp("tile workgroups, GEMM 1 phase (us from the launch's first entry):", T, ((0, "entry"), (1, "first stage landed"), (2, "main loop done"), (5, "stores issued")), 2)
grp("tile workgroups, GEMM 2 phase:", T, ((14, "row block published (poll matched)"), (9, "first stage landed"), (10, "main loop done"), (13, "stores issued"), (7, "exit")), 10)
grp("role workgroups:", slice(N1, N1 + NR), ((0, "entry"), (1, "every weight load requested"), (12, "merge items done"), (13, "gate passed"), (2, "x + weights landed"),
                                             (3, "first layer done, granules stored"), (4, "hand-off complete"), (5, "second layer done"), (7, "exit (after the sweep)")), 7)
live = tr[:, T, 10] > tr[:, T, 0]
d = lambda k1, k0: ((tr[:, T, k1] - tr[:, T, k0]) / 100.0)[live]
show("GEMM 1: first stage -> main loop done", d(2, 1))
show("GEMM 1: main loop done -> stores issued", d(5, 2))
show("stores issued -> poll matched", d(14, 5))
show("poll matched -> first stage landed", d(9, 14))
show("GEMM 2: first stage -> main loop done", d(10, 9))
show("GEMM 2: main loop done -> stores issued", d(13, 10))
print("launch: first entry -> last exit %.2f us (median over reps)" % np.median((tr[:, :, 7].max(axis=1) - t0[:, 0]) / 100.0))
