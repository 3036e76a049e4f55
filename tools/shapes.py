"""Dev tool: release-recipe forward time at BASELINE.json's other single-GPU shapes (C4, one GPU's share of C5)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
for T, hidden in ((64, 896), (32, 3584), (128, 896), (64, 3584), (256, 896)):
    ff = torch.randn(T, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
    m = bench.make_projector(bench.release_config(hidden, max(64, T)), dev)
    res = []
    with torch.no_grad():
        for fn in (lambda: m(ff, fe, g, "video", None), lambda: m.forward_deferred(ff, fe, g, "video", None)[0]):
            for _ in range(5): out = fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 100
            for _ in range(n): out = fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / n)
    byts = 2 * ff.numel() * 2
    print("T=%3d H=%4d  joined %7.1f us  deferred %7.1f us  %9.0f tok/s  inputs alone at 8 TB/s: %.1f us" %
          (T, hidden, res[0] * 1e6, res[1] * 1e6, out.shape[0] / res[1], byts / 8e6))
    del ff, fe, m
