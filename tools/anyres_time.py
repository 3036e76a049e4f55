import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
m = bench.make_projector(bench.release_config(896, 32), dev)
g = torch.randn(1152, device=dev).bfloat16()
nl = torch.randn(896, device=dev).bfloat16()
for ph, pw in ((54, 54), (27, 54), (81, 54)):
    fd = {"base": torch.randn(27, 27, 1152, device=dev).bfloat16(), "patch": torch.randn(ph, pw, 1152, device=dev).bfloat16()}
    ed = {"base": torch.randn(27, 27, 1152, device=dev).bfloat16(), "patch": torch.randn(ph, pw, 1152, device=dev).bfloat16()}
    with torch.no_grad():
        for _ in range(20): out = m(fd, ed, g, "image", nl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): out = m(fd, ed, g, "image", nl)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("anyres base 27x27 + patch %dx%d -> %d tokens: host %.1f us, wall %.1f us per forward" % (ph, pw, out.shape[0], (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
x = torch.randn(1, 27, 27, 1152, device=dev).bfloat16(); y = torch.randn(1, 27, 27, 1152, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(20): out = m(x, y, g, "image", nl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = m(x, y, g, "image", nl)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print("dense single image -> %d tokens: wall %.1f us" % (out.shape[0], (t2 - t0) / 200 * 1e6))
plans = m.__dict__.get("_engine_plans", {})
for k, pl in plans.items():
    print("plan", k[0], k[3], "fused" if pl.fused else "generic path")
