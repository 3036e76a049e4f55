"""Dev tool: host cost per dense call (short bursts so the launch queue never back-pressures)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
m = bench.make_projector(bench.release_config(896, 64), dev)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(20): m.forward_deferred(ff, fe, g, "video", None, next_guide=g)
    torch.cuda.synchronize()
    for name, fn in (("forward", lambda: m(ff, fe, g, "video", None)), ("forward_deferred", lambda: m.forward_deferred(ff, fe, g, "video", None)),
                     ("forward_deferred + next_guide", lambda: m.forward_deferred(ff, fe, g, "video", None, next_guide=g))):
        ts = []
        for rep in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            ts.append((time.perf_counter() - t0) / 8)
            torch.cuda.synchronize()
        print("%-32s host %.1f us per call (min of 8 bursts of 8)" % (name, min(ts) * 1e6))
