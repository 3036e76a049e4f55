"""Dev tool: pure host cost per sharded step (short bursts, so the launch queue never back-pressures)."""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import torch.distributed as dist
from hicom_amd.dist import sharded_forward
from hicom_amd import native as nv
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
cfg = bench.release_config(896, 64); m = bench.make_projector(cfg, dev)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
a = torch.empty(1 << 20, device=dev); b = torch.empty(1 << 20, device=dev)
with torch.no_grad():
    for _ in range(10): sharded_forward(m, ff, fe, g, 64, deferred=True)
    torch.cuda.synchronize()
    for name, fn in (("sharded deferred", lambda: sharded_forward(m, ff, fe, g, 64, deferred=True)), ("plain", lambda: m(ff, fe, g, "video", None)),
                     ("one all_gather_into_tensor", lambda: dist.all_gather_into_tensor(a, b))):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            ts.append((time.perf_counter() - t0) / 8)
            torch.cuda.synchronize()
        print("%-28s host %.1f us per call (min of 5 bursts of 8)" % (name, min(ts) * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    for rep in range(10):
        torch.cuda.synchronize(); pr.enable()
        for _ in range(8): sharded_forward(m, ff, fe, g, 64, deferred=True)
        pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumtime').print_stats(22)
dist.destroy_process_group()
