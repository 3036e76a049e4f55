"""Dev tool: where a tile iteration of the many-row stream kernel goes (in-kernel cycle sums per phase, -DHICOM_WTRACE build).
Usage on the GPU box:  python tools/wide_trace.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_wtrace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    bn.build(extra_flags=("-DHICOM_WTRACE",), lib_path=LIB, verbose=False)
    os.environ["HICOM_NATIVE_LIB"] = LIB
    sys.exit(subprocess.call([sys.executable, *sys.argv]))
import numpy as np, torch
from hicom_amd import native as nv
T, H, W, E, R = 64, 27, 27, 1152, 288
N = T * H * W
g = torch.Generator(device="cuda").manual_seed(1)
ff = torch.randn(N, E, device="cuda", generator=g).to(torch.bfloat16)
qhi = (torch.randn(R, E, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
qlo = (torch.randn(R, E, device="cuda", generator=g) * 1e-4).to(torch.bfloat16)
pos_a = torch.randn(R, 64 + H + W, device="cuda", generator=g) * 0.1
nparts = nv.global_stream_nparts(N, R)
pm, pl = torch.empty(nparts, R, device="cuda"), torch.empty(nparts, R, device="cuda")
pacc = torch.empty(nparts, R, E, device="cuda")
pmarg = torch.empty(nparts, R, nv.global_stream_marg_width(H, W), device="cuda")
f = lambda: nv.global_stream_marg(ff, N, qhi, qlo, pos_a, H, W, 0, 64, 64 + H, None, pm, pl, pacc, pmarg, rows=R)
for _ in range(5): f()
torch.cuda.synchronize()
buf = np.zeros(512 * 8 * 12, dtype=np.uint64)
L = nv.lib()
L.hicom_debug_wide_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.hicom_debug_wide_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(512, 8, 12)[:nparts].astype(np.float64)
names = {1: "DMA issue (loading waves)", 2: "score MFMAs issued", 3: "softmax quarter (LDS reads, DPP, exp, P written)", 4: "wait for the LDS writes",
         5: "barrier 1", 6: "P read, rescale, marginals, P.x", 7: "publish partials", 8: "wait vmcnt(0) (DMA of tile + 2)", 9: "barrier 2"}
tiles = tr[:, :, 0]
for label, ws in (("waves 0-3 (issue the DMA)", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print("----", label, " cycles per tile (mean over workgroups and waves)")
    tot = 0.0
    for i in range(1, 10):
        v = (tr[:, ws, i] / tiles[:, ws]).mean()
        tot += v
        print("  %-52s %7.0f" % (names[i], v))
    print("  %-52s %7.0f" % ("sum", tot))
