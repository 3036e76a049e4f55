"""Dev tool: dense16 GEMM at the head-projection / adaptor shapes, both tile variants (HICOM_DENSE16_TILE=128|256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hicom_amd import native as nv

def run(M, N, K, act, with_res, dt=torch.float16, n=10):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(dt)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if with_res else None
    npad = (N + 63) // 64 * 64
    o16 = torch.empty(M, npad, dtype=torch.float16, device="cuda") if not with_res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if with_res else None
    f = lambda: nv.dense16_gemm(a, w, b, act=act, out_f16=o16, n_store=npad if o16 is not None else None, y=y, res=res)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / n
    # sampled check against fp64
    rows = torch.randint(0, M, (64,), device="cuda")
    ref = a[rows].double() @ w.double().t() + b.double()
    if act == nv.ACT_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    got = (o16[rows][:, :N] if o16 is not None else y[rows]).double()
    if with_res:
        ref = ref + res[rows].double()
    err = float((got - ref).abs().max())
    print(f"M={M} N={N} K={K} act={act} res={with_res}: {dt_s*1e3:.3f} ms  {2.0*M*N*K/dt_s/1e12:.0f} TFLOP/s  max err {err:.2e} (scale {float(ref.abs().max()):.2f})")

print("tile", os.environ.get("HICOM_DENSE16_TILE", "auto"))
run(46656, 4304, 1152, nv.ACT_GELU_TANH, False)
run(46656, 1152, 4352, nv.ACT_NONE, True)
run(46656, 1152, 1152, nv.ACT_GELU, False, dt=torch.bfloat16)
run(46656, 1152, 1152, nv.ACT_NONE, False)
