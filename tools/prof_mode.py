"""Dev tool: host-side profile (cProfile) of one recipe's forward."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
name = sys.argv[1]
ptype, guide = {"off": ("local43_global32", None), "coarse": ("local43_global32", "coarse"), "fine": ("local43_global32", "fine"),
                "adaptkv": ("local43_adaptkv_global32", "direct")}[name]
dev = torch.device("cuda", 0)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff)
g = torch.randn(64, 1152, device=dev).bfloat16() if guide == "fine" else torch.randn(1152, device=dev).bfloat16()
cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
m = bench.make_projector(cfg, dev)
with torch.no_grad():
    for _ in range(3): m(ff, fe, g, "video", None)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): m(ff, fe, g, "video", None)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
