"""Dev tool: whole-forward time of the non-release recipes at the C2 shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench, hicom_amd
from types import SimpleNamespace
dev = torch.device("cuda", 0)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
gl = torch.randn(64, 1152, device=dev).bfloat16()
for name, ptype, guide in (("direct", "local43_global32_coarse", "direct"), ("off", "local43_global32", None), ("coarse", "local43_global32", "coarse"),
                           ("fine", "local43_global32", "fine"), ("adaptkv", "local43_adaptkv_global32", "direct")):
    cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
    m = bench.make_projector(cfg, dev)
    gg = gl if guide == "fine" else g
    with torch.no_grad():
        for _ in range(3): out = m(ff, fe, gg, "video", None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): out = m(ff, fe, gg, "video", None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-8s %8.1f us/forward  %10.0f tok/s  out %s" % (name, dt * 1e6, out.shape[0] / dt, tuple(out.shape)))
