import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_golden_head as mh
from hicom_amd import encoder
from test_gpu_head import _Head
sd = mh.head_state_dict()
m = _Head(sd).to(torch.bfloat16).cuda().train()
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_head_v1.npz"))
x = torch.from_numpy(mh.tokens()).to(torch.bfloat16).cuda()
out = encoder.siglip_head_embed(x, m)
R = torch.from_numpy(mh.cotangent()).cuda()
(out.float() * R).sum().backward()
fp32 = dict(encoder.LAST_FP32_GRADS)
for k in fp32:
    want = gold[f"grad/head.{k}/samples"]; s, sabs, mx = gold[f"grad/head.{k}/sums"]
    pos = torch.from_numpy(mh.sample_positions(fp32[k].numel())).cuda()
    got = fp32[k].reshape(-1)[pos].cpu().numpy()
    print(k, "err", float(np.abs(got - want).max()), "tol", 2e-3 * mx, "max", mx)
