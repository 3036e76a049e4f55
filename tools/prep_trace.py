"""Dev tool: timeline of query_prep_kernel inside the release step, from in-kernel s_memrealtime stamps (-DHICOM_TRACE build)."""
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
L.hicom_debug_prep_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
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
        buf = np.zeros(512 * 8, dtype=np.uint64)
        assert L.hicom_debug_prep_trace(buf.ctypes.data, buf.nbytes) == 0
        agg.append(buf.reshape(512, 8).astype(np.int64))
tr = np.stack(agg)
nq, nr, nf, npos = 72, 56, 81, 18
t0 = np.where(tr[:, :, 0] > 0, tr[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]
def show(name, x):
    print("  %-40s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, *np.percentile(x, [10, 50, 90]), x.max()))
for label, sl, names in (("q_proj", slice(0, nq), ((0, "entry"), (1, "loads requested"), (2, "dots done"), (3, "granules stored"), (7, "exit"))),
                         ("r0", slice(nq, nq + nr), ((0, "entry"), (7, "exit"))),
                         ("fold-w", slice(nq + nr, nq + nr + nf), ((0, "entry"), (1, "weights requested"), (2, "granules swept"), (7, "exit"))),
                         ("fold-pos", slice(nq + nr + nf, nq + nr + nf + npos), ((0, "entry"), (1, "weights requested"), (2, "granules swept"), (7, "exit")))):
    print(label)
    for k, nm in names:
        show(nm, ((tr[:, sl, k] - t0) / 100.0).ravel())
# per-block medians over the repetitions: are the late fold workgroups always the same ones?
med = np.median((tr[:, :, :] - t0[:, :, None]) / 100.0, axis=0)          # [block, stamp]
fold = slice(nq + nr, nq + nr + nf)
order = np.argsort(med[fold, 2])
print("fold-w blocks by median 'granules swept' (block: head, slab -> swept, exit):")
for i in list(order[:4]) + list(order[-12:]):
    print("   block %3d: head %d slab %d  swept %.2f exit %.2f" % (i, i // 9, i % 9, med[nq + nr + i, 2], med[nq + nr + i, 7]))
print("q_proj blocks by median 'granules stored' (latest 8):")
oq = np.argsort(med[:nq, 3])
for i in oq[-8:]:
    print("   block %3d (head %d): loads %.2f stored %.2f" % (i, i // 8, med[i, 1], med[i, 3]))
