"""Dev: does any kernel of the operator-by-operator forward (the training forward of coarse / fine injection, guide off, query-side adaptors) READ a
temporary it -- or its producer -- never wrote?  Every temporary the Python side allocates (`_f32` in projector.py / injector.py, torch.empty in
the same modules) is filled with NaN (or 1e30) first; a read of unwritten bytes then shows in the output.  usage: python3 tools/poison_temps.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import hicom_amd.projector as P, hicom_amd.injector as I
from test_gpu_backward import cases, build_module, dev_bf16

real_empty = torch.empty
FILL = [None]


def poisoned_empty(*a, **k):
    t = real_empty(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        v = FILL[0]
        if t.dtype == torch.float16 and abs(v) > 6e4:
            v = 6e4 if v > 0 else -6e4
        t.fill_(v)
    return t


for name in ("G6_coarse", "G7_fine", "G2_off_T8", "G5b_adaptqkvg_off", "G1_direct_T8", "G7b_guide_override"):
    case = cases.build_case(name)
    m = build_module(case).eval()
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    with torch.no_grad():
        FILL[0] = None
        want = m.forward_stepwise(ff, fe, g, case.modal, None).clone()
        res = []
        for fill in (float("nan"), 1e30, -1e30, 7.0):
            FILL[0] = fill
            torch.empty = poisoned_empty
            try:
                got = m.forward_stepwise(ff, fe, g, case.modal, None)
            finally:
                torch.empty = real_empty
            torch.cuda.synchronize()
            d = (got.float() - want.float())
            bad = ~torch.isfinite(got.float()) | (d.abs() > 0)
            res.append((fill, int(bad.any(1).sum()), torch.nonzero(bad.any(1)).flatten()[:6].tolist(), float(d[torch.isfinite(d)].abs().max()) if torch.isfinite(d).any() else float("nan")))
    print(name, tuple(want.shape), [f"fill {f}: {n} rows differ {rows} max {mx:.2e}" for f, n, rows, mx in res], flush=True)

# the TRAINING forward (autograd on: the global stage's state and logits go into the per-shape store, the window contexts are kept)
for name in ("G6_coarse", "G7_fine", "G2_off_T8", "G5b_adaptqkvg_off"):
    case = cases.build_case(name)
    m = build_module(case).train()
    m.graph_backward = False
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    FILL[0] = None
    want = m(ff, fe, g, case.modal, None).detach().clone()
    res = []
    for fill in (float("nan"), 1e30, 7.0):
        FILL[0] = fill
        torch.empty = poisoned_empty
        try:
            out = m(ff, fe, g, case.modal, None)
            got = out.detach().clone()
        finally:
            torch.empty = real_empty
        FILL[0] = None
        out.sum().backward()                      # (an eager backward between two forwards, as in the flake's loop)
        m.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        d = got.float() - want.float()
        bad = ~torch.isfinite(got.float()) | (d.abs() > 0)
        res.append(f"fill {fill}: {int(bad.any(1).sum())} rows differ {torch.nonzero(bad.any(1)).flatten()[:6].tolist()}")
    print("training", name, res, flush=True)
