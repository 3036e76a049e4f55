"""Dev tool: where one training step of the release recipe spends its time (torch profiler: host ops + device kernels)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev).train()
m.graph_backward = os.environ.get("GRAPH_BWD", "0") == "1"
gen = torch.Generator(device=dev).manual_seed(1)
ff = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).to(torch.bfloat16)
fe = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).to(torch.bfloat16)
g = torch.randn(1152, device=dev, generator=gen).to(torch.bfloat16)
cot = None

def step():
    global cot
    m.zero_grad(set_to_none=True)
    o = m(ff, fe, g, "video", None)
    if cot is None:
        cot = torch.randn(o.shape, device=dev, generator=gen).to(o.dtype)
    o.backward(cot)

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"train step {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
# forward only / backward only
t0 = time.perf_counter()
outs = []
for _ in range(20):
    outs.append(m(ff, fe, g, "video", None))
torch.cuda.synchronize()
print(f"forward (training mode) {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
t0 = time.perf_counter()
for o in outs:
    o.backward(cot)
torch.cuda.synchronize()
print(f"backward {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
if m.graph_backward:
    print("captured:", "graph" in next(iter(m.__dict__.get("_bwd_graphs", {}).values()), {}))
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
