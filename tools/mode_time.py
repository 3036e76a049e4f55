"""Dev: wall time of one non-release recipe's forward at C2 (best of 5 x 20 calls).  usage: mode_time.py off|coarse|fine|adaptkv"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
MODES = {"off": ("local43_global32", None), "coarse": ("local43_global32", "coarse"), "fine": ("local43_global32", "fine"),
         "adaptkv": ("local43_adaptkv_global32", "direct")}
dev = torch.device("cuda", 0)
for name in sys.argv[1:]:
    ptype, guide = MODES[name]
    ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff)
    g = torch.randn(64, 1152, device=dev).bfloat16() if guide == "fine" else torch.randn(1152, device=dev).bfloat16()
    cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
    m = bench.make_projector(cfg, dev)
    with torch.no_grad():
        for _ in range(10): m(ff, fe, g, "video", None)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): m(ff, fe, g, "video", None)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
    print(f"{name}: {best * 1e6:.0f} us")
