import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from hicom_amd import native as nv
E, nh, S, hd = 1152, 9, 144, 128
g = torch.Generator().manual_seed(1)
M, N, K = 1296, 896, 1152
a16 = nv.to_f16(torch.randn(M, K, generator=g).cuda()); w16 = nv.to_f16(torch.randn(N, K, generator=g).cuda() * 0.02)
wv = (torch.randn(E, E, generator=g) * 0.02).cuda().bfloat16()
for nparts in (1, 12, 216):
    pm = torch.zeros(nparts, 16).cuda(); pl = torch.ones(nparts, 16).cuda()
    p16 = torch.zeros(nparts, 16, E, dtype=torch.float16, device="cuda")
    vpe = (torch.arange(E).view(E, 1) * 1.0 + torch.arange(S).view(1, S) / 256.0).cuda().to(torch.float16)     # vpe[r][s] = r + s/256 (exact in fp16 for r < 2048: no; coarse is fine)
    vpe = (torch.arange(E).view(E, 1) % 64 + torch.arange(S).view(1, S) / 256.0).cuda().to(torch.float16)
    for s0 in (0, 1, 7, 8, 20, 117):
        mg = torch.zeros(nparts, nh, S, dtype=torch.float16, device="cuda"); mg[:, :, s0] = 1.0
        ofx = torch.zeros(E, dtype=torch.int64, device="cuda"); o16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        nv.readout16_gemm(a16, w16, None, act=nv.ACT_GELU, out_f16=o16, merge=dict(part_m=pm, part_l=pl, part_ctx16=p16, rows=nh, w_v=wv, o_fix=ofx, part_marg=mg, vpe_f16=vpe))
        torch.cuda.synchronize()
        got = ofx.double() / 2.0 ** 36
        want = vpe[:, s0].double()
        bad = (got - want).abs() > 1e-3
        print(f"nparts {nparts:3d} s0 {s0:3d}: wrong {int(bad.sum()):4d} of {E}; got[:6] {[round(float(x), 3) for x in got[:6]]} want[:6] {[round(float(x), 3) for x in want[:6]]}  got[128:131] {[round(float(x), 3) for x in got[128:131]]} want {[round(float(x), 3) for x in want[128:131]]}")
