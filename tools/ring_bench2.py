"""Dev: does the ring-alone time depend on what ran before?  (a) ring only from a cold start: per-batch times; (b) 2000 release steps, then ring only."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from hicom_amd import native as nv
dev = torch.device("cuda", 0)
m = bench.make_projector(bench.release_config(896, 64), dev)
T, H, W, E = 64, 27, 27, 1152
sets = [(torch.randn(T, H, W, E, device=dev).bfloat16(), torch.randn(T, H, W, E, device=dev).bfloat16(), torch.randn(E, device=dev).bfloat16()) for _ in range(3)]
g = sets[0][2]
qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (qhi.float() * 1e-3).bfloat16(); qhi[9:] = g; qlo[9:] = 0
pos_a = torch.randn(16, T + 54, device=dev) * 0.1
nw = 1296; nparts = nv.fused_stream_nparts(nw)
pe = torch.randn(T + 54, E, device=dev); pe_hi = pe.bfloat16(); pe_lo = (pe - pe_hi.float()).bfloat16()
pm, pl = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev)
p16 = torch.empty(nparts, 16, E, device=dev, dtype=torch.float16); c16 = torch.empty(nw, E, device=dev, dtype=torch.float16)
zero = torch.zeros(E, dtype=torch.int64, device=dev)
i = [0]
def ring():
    a, b, _ = sets[i[0] % 3]; i[0] += 1
    nv.fused_stream(a, b, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, T, T + H, pm, pl, None, None, ctx_f16=c16, zero=zero, part_ctx_f16=p16)
def step():
    a, b, gg = sets[i[0] % 3]; i[0] += 1
    m(a, b, gg, "video", None)
def batches(n):
    out = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ring()
        b.record(); torch.cuda.synchronize()
        out.append(round(a.elapsed_time(b) * 100, 1))
    return out
with torch.no_grad():
    ring(); torch.cuda.synchronize()
    print("cold start, ring only:", batches(30))
    for _ in range(4000): step()
    torch.cuda.synchronize()
    print("after 4000 release steps (0.3 s), ring only:", batches(30))
    import time; time.sleep(1.0)
    print("after 1 s idle, ring only:", batches(12))
