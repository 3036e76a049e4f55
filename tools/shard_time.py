"""Dev tool: step time of sharded_forward (world size 1, RCCL) vs the plain forward at the C2 shape."""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import torch.distributed as dist
from hicom_amd.dist import sharded_forward
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
n = 200
with torch.no_grad():
    for name, fn in (("plain", lambda: m(ff, fe, g, "video", None)), ("sharded(world=1)", lambda: sharded_forward(m, ff, fe, g, 64)), ("sharded deferred", lambda: sharded_forward(m, ff, fe, g, 64, deferred=True)[0])):
        for _ in range(10): out = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ref = m(ff, fe, g, "video", None); torch.cuda.synchronize()
        print("equal to plain:", torch.equal(out, ref), end="  ")
        print("%-18s %7.1f us/step (host enqueue %.1f us)" % (name, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6))
dist.destroy_process_group()
import cProfile, pstats
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port + 1}", rank=0, world_size=1, device_id=dev)
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for _ in range(200): sharded_forward(m, ff, fe, g, 64, deferred=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
dist.destroy_process_group()
