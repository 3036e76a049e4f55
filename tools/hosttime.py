"""Dev tool: host-side cost of one projector call (no device sync inside the loop)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
cfg = bench.release_config(896, 64)
dev = torch.device("cuda", 0)
m = bench.make_projector(cfg, dev)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
for mode in ("eager", "graph"):
    m.graph_replay = mode == "graph"
    with torch.no_grad():
        for _ in range(10): m(ff, fe, g, "video", None)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): m(ff, fe, g, "video", None)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, "host us/call %.1f   wall us/call %.1f" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
from hicom_amd import engine
import cProfile, pstats
m.graph_replay = False
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for _ in range(200): m(ff, fe, g, "video", None)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
