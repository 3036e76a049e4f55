"""Dev tool: run under `rocprofv3 --kernel-trace --output-format csv`; 30 plain launches per K, then
`python tools/pgemm_trace.py parse <csv>` prints the median kernel duration per K (true durations, no host bound)."""
import os, sys
KS = (64, 128, 256, 512, 896, 1152, 2304)
if len(sys.argv) > 2 and sys.argv[1] == "parse":
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "planes_gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    for i, K in enumerate(KS):
        for tag, off in (("plain", 0), ("gelu+planes", 30)):
            g = sorted(d[i * 60 + off:i * 60 + off + 30])
            print("K=%4d %-12s median %.2f us  min %.2f" % (K, tag, g[len(g) // 2], g[0]))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hicom_amd import native as nv
dev, M, N = "cuda", 1296, 896
for K in KS:
    ah = torch.randn(M, K, device=dev).bfloat16(); al = (torch.randn(M, K, device=dev) * 1e-3).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); hh = torch.empty_like(out); hl = torch.empty_like(out)
    for _ in range(30): nv.planes_gemm(ah, al, w, b, y=out)
    for _ in range(30): nv.planes_gemm(ah, al, w, b, act=1, out_hi=hh, out_lo=hl)
    torch.cuda.synchronize()
