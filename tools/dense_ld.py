"""Dev tool: dense16 GEMM time against the leading dimensions of A / W (L2 channel spread of the row pitch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hicom_amd import native as nv

def run(M, N, K, pad, n=10):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(M, K + pad, device="cuda", generator=g) * 0.5).to(torch.float16)
    w = (torch.randn(N, K + pad, device="cuda", generator=g) * 0.03).to(torch.float16)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    f = lambda: nv.dense16_gemm(a, w, b, K=K, y=y)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K} ld={K + pad}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TFLOP/s")

print("tile", os.environ.get("HICOM_DENSE16_TILE", "auto"))
for pad in (0, 64, 128, 192):
    run(46656, 1152, 4352, pad)
for pad in (0, 64, 128):
    run(46656, 4352, 1152, pad)
