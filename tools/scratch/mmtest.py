import torch, time
a = torch.randn(4304, 46656, device="cuda").bfloat16(); b = torch.randn(46656, 1152, device="cuda").bfloat16()
try:
    c = torch.mm(a, b, out_dtype=torch.float32); print("out_dtype ok", c.dtype)
except Exception as e:
    print("out_dtype failed:", type(e).__name__, str(e)[:200])
for f, name in ((lambda: torch.mm(a, b), "bf16 out"), (lambda: torch.mm(a.float(), b.float()), "fp32")):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 3 * 1e3, "ms")
try:
    f = lambda: torch.mm(a, b, out_dtype=torch.float32)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); print("out_dtype f32", (time.perf_counter() - t0) / 3 * 1e3, "ms")
except Exception as e:
    pass
