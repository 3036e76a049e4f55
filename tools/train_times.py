"""Dev tool: forward / training-step wall times of a recipe at C2 (bench.py's `secondary` / `neighbours.train_step` numbers, one recipe)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["adaptkv", "release"]
R = {"off": ("local43_global32", None), "coarse": ("local43_global32", "coarse"), "fine": ("local43_global32", "fine"),
     "adaptkv": ("local43_adaptkv_global32", "direct"), "release": ("local43_global32_coarse", "direct")}
gen = torch.Generator(device=dev).manual_seed(3)
ff = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).bfloat16(); fe = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).bfloat16()
def best(fn, n=5, reps=3):
    for _ in range(2): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    return min(ts) * 1e3
for name in which:
    ptype, guide = R[name]
    g = torch.randn(64, 1152, device=dev).bfloat16() if guide == "fine" else torch.randn(1152, device=dev).bfloat16()
    cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
    m = bench.make_projector(cfg, dev)
    with torch.no_grad():
        out = m(ff, fe, g, "video", None)
        fwd = best(lambda: m(ff, fe, g, "video", None), n=10)
    m.train()
    cot = torch.randn(out.shape, device=dev).to(out.dtype)
    def step(inputs=False):
        fe_, g_ = (fe.detach().requires_grad_(True), g.detach().requires_grad_(True)) if inputs else (fe, g)
        m.zero_grad(set_to_none=True)
        m(ff, fe_, g_, "video", None).backward(cot)
    tr = best(step, n=3)
    tri = best(lambda: step(True), n=3) if guide in ("direct", "coarse", "fine") else float("nan")
    for e in m.__dict__.get("_bwd_graphs", {}).values():
        print("   backward graph:", "captured" if "graph" in e else ("FAILED " + str(e.get("failed"))[:300]))
    print(f"{name}: forward {fwd * 1e3:.0f} us, train step {tr:.2f} ms, with input grads {tri:.2f} ms")
