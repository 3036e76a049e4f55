"""Dev tool: planes GEMM time vs K (and epilogue kind) -- separates the per-stage cost from the fixed cost."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv
dev = "cuda"
M, N = 1296, 896
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("lib", nv.LIB_PATH)
for K in (64, 128, 256, 512, 896, 1152, 2304):
    ah = torch.randn(M, K, device=dev).bfloat16(); al = (torch.randn(M, K, device=dev) * 1e-3).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev).bfloat16()
    hh = torch.empty(M, N, device=dev, dtype=torch.bfloat16); hl = torch.empty_like(hh); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: nv.planes_gemm(ah, al, w, b, act=1, out_hi=hh, out_lo=hl))
    t2 = timeit(lambda: nv.planes_gemm(ah, al, w, b, y=out))
    t3 = timeit(lambda: nv.planes_gemm(ah, None, w, b, y=out))
    print("K=%4d  gelu+planes %.1f us   plain bf16 %.1f us   no-lo %.1f us" % (K, t1, t2, t3))
