"""Dev tool: the ring kernel alone, launched as the release step launches it (fp16 window contexts, normalised fp16 partial contexts,
the zeroed accumulators, value-side pos-emb), rotating three input sets (HBM, not Infinity Cache); HIP events over batches of 10."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv
dev = "cuda"
T, H, W, E = int(os.environ.get("T", 64)), 27, 27, 1152
sets = [(torch.randn(T, H, W, E, device=dev).bfloat16(), torch.randn(T, H, W, E, device=dev).bfloat16()) for _ in range(3)]
g = torch.randn(E, device=dev).bfloat16()
qhi = (torch.randn(16, E, device=dev) * 0.05).bfloat16(); qlo = (qhi.float() * 1e-3).bfloat16(); qhi[9:] = g; qlo[9:] = 0
pos_a = torch.randn(16, T + 54, device=dev) * 0.1
nw = (T // 4) * 81
nparts = nv.fused_stream_nparts(nw)
pe = torch.randn(T + 54, E, device=dev); pe_hi = pe.bfloat16(); pe_lo = (pe - pe_hi.float()).bfloat16()
pm, pl = torch.empty(nparts, 16, device=dev), torch.empty(nparts, 16, device=dev)
p16 = torch.empty(nparts, 16, E, device=dev, dtype=torch.float16)
c16 = torch.empty(nw, E, device=dev, dtype=torch.float16)
zero = torch.zeros(E, dtype=torch.int64, device=dev)
i = [0]
def run():
    a, b = sets[i[0] % 3]; i[0] += 1
    nv.fused_stream(a, b, 4, 3, qhi, qlo, 9, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, T, T + H, pm, pl, None, None, ctx_f16=c16, zero=zero, part_ctx_f16=p16)
for _ in range(300): run()
torch.cuda.synchronize()
res = []
for rep in range(40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    res.append(a.elapsed_time(b) * 100)
res.sort()
print("ring alone (%s): median %.2f us  min %.2f  frac of 8 TB/s %.3f" % (os.environ.get("HICOM_NATIVE_LIB", "product"), res[len(res) // 2], res[0], 3359232 * T / (res[len(res) // 2] * 1e-6) / 8e12))
