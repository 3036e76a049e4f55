"""Dev tool: the readout GEMM launch alone in a loop (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv
dev = "cuda"
M, N, K = 1296, 896, 1152
a = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.02).half(); b = torch.zeros(N, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    nv.readout16_gemm(a, w, b, out_f16=out, act=1)
torch.cuda.synchronize()
