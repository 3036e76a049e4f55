import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hicom_amd import native as nv
M, N, K = 46656, 4352, 1152
g = torch.Generator(device="cuda").manual_seed(1)
a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.float16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.float16)
b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
o = torch.empty(M, N, dtype=torch.float16, device="cuda")
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("y bf16, no act      %.3f ms" % t(lambda: nv.dense16_gemm(a, w, b, y=y)))
    print("y bf16, tanh gelu   %.3f ms" % t(lambda: nv.dense16_gemm(a, w, b, act=nv.ACT_GELU_TANH, y=y)))
    print("o fp16, no act      %.3f ms" % t(lambda: nv.dense16_gemm(a, w, b, out_f16=o)))
    print("o fp16, tanh gelu   %.3f ms" % t(lambda: nv.dense16_gemm(a, w, b, act=nv.ACT_GELU_TANH, out_f16=o)))
    print("o fp16, erf gelu    %.3f ms" % t(lambda: nv.dense16_gemm(a, w, b, act=nv.ACT_GELU, out_f16=o)))
