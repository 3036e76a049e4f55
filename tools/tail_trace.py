"""Dev tool: timeline of the LAST readout16 launch of a release step (GEMM 2 + its role) from in-kernel s_memrealtime stamps.
Builds a second library with -DHICOM_TRACE (never the product one).  Usage on the GPU box:  python tools/tail_trace.py [which]
which = 2 (default): GEMM 2's launch; 1: GEMM 1's (stops the step after it by running the role launch directly is not possible -- the
trace buffer holds the last launch, so `1` sets HICOM_TAIL_LAUNCHES=5-free... see below)."""
import ctypes, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_trace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    bn.build(extra_flags=("-DHICOM_TRACE",), lib_path=LIB, verbose=False)
    os.environ["HICOM_NATIVE_LIB"] = LIB
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
tr = np.stack(agg)                      # [rep, block, 8]
n_gemm = 200
def show(name, x):
    x = x[np.isfinite(x)]
    print("  %-44s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, *np.percentile(x, [10, 50, 90]), x.max()))
t0 = np.where(tr[:, :, 0] > 0, tr[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]      # first entry of the launch
rel = lambda k, sl: ((tr[:, sl, k] - t0) / 100.0)[tr[:, sl, 0] > 0]
tiles = slice(0, n_gemm)
live = tr[:, :n_gemm, 2] > tr[:, :n_gemm, 0]
print("tile workgroups (us from the launch's first entry):")
show("entry", rel(0, tiles))
for k, nm in ((1, "first stage landed"), (2, "main loop done"), (5, "stores issued (exit)")):
    x = ((tr[:, :n_gemm, k] - t0) / 100.0)[live]
    show(nm, x)
roles = slice(n_gemm, n_gemm + 80)
used = tr[0, n_gemm:n_gemm + 80, 0] > 0
print("role workgroups: %d" % used.sum())
for k, nm in ((0, "entry"), (6, "role entered (kernargs read)"), (8, "counter requested"), (9, "x requested"), (10, "biases requested"), (11, "layer-1 rows requested"), (1, "every load requested"), (2, "x landed"), (3, "first layer done, granules stored"), (4, "hand-off complete"),
              (5, "second layer done"), (7, "exit")):
    x = ((tr[:, n_gemm:n_gemm + 80, k] - t0) / 100.0)[:, used]
    show(nm, x.ravel())

xc = tr[:, :, 14]
have = xc >= 1000
blk = np.arange(tr.shape[1])[None, :].repeat(tr.shape[0], 0)
off = ((xc - 1000) - blk) % 8
print("XCC of workgroup b minus b, mod 8 (launches x workgroups):", {int(k): int((off[have] == k).sum()) for k in range(8)})
print("  per launch (first 8 launches), offset of block 0 / consistent within the launch:", [(int(off[r][have[r]][0]), bool((off[r][have[r]] == off[r][have[r]][0]).all())) for r in range(min(8, tr.shape[0]))])
