"""Dev tool (round 5): the readout GEMM launch with each role, cold caches (a 400 MB copy between launches), timed from the rocprofv3
kernel trace:   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/tail_bench.py   then   python3 tools/tail_bench.py --report <dir>
Variants run in a fixed order, REPS launches each; the report groups the readout16 kernel durations by position."""
import csv
import glob
import os
import sys

REPS = 60
VARIANTS = ["gemm1", "gemm1+aux1", "gemm1+merge", "gemm2", "gemm2+aux2", "gemm2+chain", "chain alone (1 tile)", "aux2 alone (1 tile)", "aux1 alone (1 tile)", "merge alone (1 tile)"]

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import statistics as st
    f = glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "readout16_gemm_kernel" in r["Kernel_Name"]]
    d = d[-REPS * len(VARIANTS):]
    for i, v in enumerate(VARIANTS):
        x = sorted(d[i * REPS:(i + 1) * REPS])
        print(f"  {v:28s} median {st.median(x):6.2f}  min {x[0]:6.2f}  p90 {x[int(len(x) * 0.9)]:6.2f} us")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
E, H, NW, nh, nparts = 1152, 896, 1296, 9, 216
r = lambda *s: torch.randn(*s, device=dev, generator=g)
ctx16 = nv.to_f16(r(NW, E))
hid16 = nv.to_f16(r(NW, H))
w0_16, w2_16 = nv.to_f16(r(H, E) * 0.02), nv.to_f16(r(H, H) * 0.02)
b0, b2 = (r(H) * 0.02).bfloat16(), (r(H) * 0.02).bfloat16()
hid_out = torch.empty(NW, H, dtype=torch.float16, device=dev)
out = torch.empty(NW + 32, H, dtype=torch.bfloat16, device=dev)
pm, pl = r(nparts, 16) * 3, torch.rand(nparts, 16, device=dev, generator=g) + 0.5
p16 = r(nparts, 16, E).to(torch.float16)
wv, bv = (r(E, E) * 0.02).bfloat16(), (r(E) * 0.02).bfloat16()
ofx = torch.zeros(E, dtype=torch.int64, device=dev)
ofx_in = torch.round(r(E).double() * 2.0 ** 36).to(torch.int64)
c0, r0 = r(H, E) * 0.03, r(H) * 0.1
gw2, gb2 = (r(H, H) * 0.03).bfloat16(), (r(H) * 0.02).bfloat16()
hid_g = torch.empty(H, device=dev)
state = nv.r16_chain_state(H, dev)
state1 = nv.r16_chain_state(H, dev)
a_s, w_s, o_s = nv.to_f16(r(96, 128)), nv.to_f16(r(64, 128)), torch.empty(96, 64, dtype=torch.float16, device=dev)
big_a, big_b = torch.empty(100 << 20, dtype=torch.float32, device=dev), torch.empty(100 << 20, dtype=torch.float32, device=dev)
aux1 = dict(x_fixed=ofx_in, xb=bv, w=c0, b=r0, act=nv.ACT_GELU, y=hid_g)
aux2 = dict(xs=hid_g, w=gw2, b=gb2, act=nv.ACT_NONE, rows=(out, NW, 32))
aux2c = dict(w=gw2, b=gb2, act=nv.ACT_NONE, rows=(out, NW, 32))
mrg = dict(part_m=pm, part_l=pl, part_ctx16=p16, rows=nh, w_v=wv, o_fix=ofx)
fns = [
    lambda: nv.readout16_gemm(ctx16, w0_16, b0, act=nv.ACT_GELU, out_f16=hid_out),
    lambda: nv.readout16_gemm(ctx16, w0_16, b0, act=nv.ACT_GELU, out_f16=hid_out, aux=aux1),
    lambda: nv.readout16_gemm(ctx16, w0_16, b0, act=nv.ACT_GELU, out_f16=hid_out, merge=mrg),
    lambda: nv.readout16_gemm(hid16, w2_16, b2, y=out),
    lambda: nv.readout16_gemm(hid16, w2_16, b2, y=out, aux=aux2),
    lambda: nv.readout16_gemm(hid16, w2_16, b2, y=out, chain=(aux1, aux2c, state)),
    lambda: nv.readout16_gemm(a_s, w_s, None, out_f16=o_s, chain=(aux1, aux2c, state1)),
    lambda: nv.readout16_gemm(a_s, w_s, None, out_f16=o_s, aux=aux2),
    lambda: nv.readout16_gemm(a_s, w_s, None, out_f16=o_s, aux=aux1),
    lambda: nv.readout16_gemm(a_s, w_s, None, out_f16=o_s, merge=mrg),
]
assert len(fns) == len(VARIANTS)
for f in fns:
    f()
torch.cuda.synchronize()
for f in fns:
    for _ in range(REPS):
        big_b.copy_(big_a)          # 800 MB through the caches: the next launch finds its operands in HBM
        f()
    torch.cuda.synchronize()
print("done")
