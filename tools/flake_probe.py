"""Dev: hunts a rare nondeterminism -- the training forward of a recipe on two identical modules through weight updates (the loop of
tests/test_gpu_backward.py::test_graph_backward_follows_weight_updates_between_steps); reports which module's forward moved."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from gpu_util import build_module, dev_bf16
name = sys.argv[1] if len(sys.argv) > 1 else "G7_fine"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
graph_a = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
bad = 0
for rep in range(reps):
    case = cases.build_case(name)
    ma, mb = build_module(case).train(), build_module(case).train()
    mb.graph_backward = False
    if not graph_a:
        ma.graph_backward = False
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    gen = torch.Generator(device="cuda").manual_seed(11)
    R = None
    for stepno in range(5):
        outs = []
        for m in (ma, mb):
            m.zero_grad(set_to_none=True)
            out = m(ff, fe, g, case.modal, None)
            if R is None:
                R = torch.randn(out.shape, device="cuda", generator=gen).to(out.dtype)
            snap = out.detach().clone()
            out.backward(R)
            outs.append((snap, out.detach().clone()))
        with torch.no_grad():
            ref = ma.forward_stepwise(ff, fe, g, case.modal, None)
        a0, a1 = outs[0]; b0, b1 = outs[1]
        if not (torch.equal(a0, b0) and torch.equal(a1, b1)):
            bad += 1
            print(f"rep {rep} step {stepno}: a_pre==b_pre {torch.equal(a0, b0)}  a_post==a_pre {torch.equal(a0, a1)}  b_post==b_pre {torch.equal(b0, b1)}  "
                  f"a_pre==ref {torch.equal(a0, ref)}  b_pre==ref {torch.equal(b0, ref)}  max|a-b| {float((a0 - b0).abs().max()):.3e}  rows differing "
                  f"{int(((a0 != b0).any(1)).sum())} of {a0.shape[0]}: {torch.nonzero((a0 != b0).any(1)).flatten()[:8].tolist()}")
        with torch.no_grad():
            for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
                delta = (torch.randn(pa.shape, device="cuda", generator=gen) * 0.01).to(pa.dtype)
                pa.add_(delta); pb.add_(delta)
print(f"{name}: {bad} deviating steps in {reps} repetitions x 5 steps")
