"""Dev tool: the readout GEMM launch ALONE, back to back on the same operands (cache-hot), with the in-kernel stamps of tools/tail_trace.py:
how long is a tile's K loop when nothing has to come from beyond the caches?  (In the step: first stage landed 2.2 us, main loop done 6.4-7.0.)"""
import ctypes, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = "/tmp/libhicom_trace.so"
if "HICOM_NATIVE_LIB" not in os.environ:
    from hicom_amd import build_native as bn
    bn.build(extra_flags=("-DHICOM_TRACE", *[f for f in os.environ.get("R16_DEV_FLAGS", "").split() if f]), lib_path=LIB + os.environ.get("R16_DEV_TAG", ""), verbose=False)
    LIB = LIB + os.environ.get("R16_DEV_TAG", "")
    os.environ["HICOM_NATIVE_LIB"] = LIB
    sys.exit(subprocess.call([sys.executable, *sys.argv]))
import numpy as np, torch
from hicom_amd import native as nv
dev = "cuda"
L = nv.lib()
L.hicom_debug_r16_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
for (M, N, K, tag) in ((1296, 896, 1152, "GEMM 1 shape"), (1296, 896, 896, "GEMM 2 shape"), (648, 3584, 3584, "hidden 3584, GEMM 2 shape")):
    a = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.02).half(); b = torch.zeros(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for mode in ("hot", "cold"):
        agg = []
        for rep in range(12):
            if mode == "cold":
                big.fill_(rep)                                   # 512 MB through every cache
            for _ in range(1 if mode == "cold" else 6):
                nv.readout16_gemm(a, w, b, out_f16=out, act=1)
            torch.cuda.synchronize()
            buf = np.zeros(512 * 16, dtype=np.uint64)
            assert L.hicom_debug_r16_trace(buf.ctypes.data, buf.nbytes) == 0
            agg.append(buf.reshape(512, 16).astype(np.int64))
        tr = np.stack(agg[2:])
        live = tr[:, :, 2] > tr[:, :, 0]
        t0 = np.where(tr[:, :, 0] > 0, tr[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]
        f = lambda k: np.median(((tr[:, :, k] - t0) / 100.0)[live])
        print("%-28s %-4s tiles %3d  first stage landed %5.2f  main loop done %5.2f  stores issued %5.2f us   (K loop %.2f us = %.0f GB/s per CU)" % (
            tag, mode, int(live[0].sum()), f(1), f(2), f(5), f(2) - f(1), ((96 + (64 if N == 896 else 128)) * K * 2) / ((f(2) - f(1)) * 1e-6) / 1e9))
