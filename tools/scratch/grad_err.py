"""Dev: per-parameter gradient error of one fixture case against golden_grad_v1.npz (fraction of the parameter's largest entry)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import cases, make_golden_grad as mg
from gpu_util import build_module, dev_bf16
from hicom_amd import autograd as hag
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_grad_v1.npz"))
for name in sys.argv[1:]:
    case = cases.build_case(name)
    m = build_module(case).train()
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    out = m(ff, fe, g, case.modal, None)
    R = torch.from_numpy(mg.cotangent(name, out.shape)).cuda()
    (out * R).sum().backward()
    fp32 = dict(hag.LAST_FP32_GRADS)
    mx_case = max(float(gold[f][2]) for f in gold.files if f.startswith(name + "/") and f.endswith("/sums"))
    worst = []
    for k, p in m.named_parameters():
        if f"{name}/{k}/samples" not in gold or k not in fp32: continue
        want = gold[f"{name}/{k}/samples"]; mx = float(gold[f"{name}/{k}/sums"][2])
        pos = torch.from_numpy(mg.sample_positions(p.numel())).cuda()
        err = float(np.abs(fp32[k].reshape(-1)[pos].cpu().numpy() - want).max())
        worst.append((err / max(max(mx, 5e-3 * mx_case), 1e-30), k, err, mx))
    worst.sort(reverse=True)
    print(name, [(round(w[0], 5), w[1]) for w in worst[:5]])
