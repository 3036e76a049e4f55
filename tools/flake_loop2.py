"""Dev: the loop of tools/flake_loop.py with host synchronisation points that can be switched on (FLAKE_SYNC = none | after_bwd | after_update |
after_fwd): which asynchronous overlap does the once-in-50 forward mismatch of the eager-backward module need?  Only the cases that failed
(coarse / fine).  usage: FLAKE_SYNC=... python3 tools/flake_loop2.py <reps>"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_gpu_backward as tb
from test_gpu_backward import cases, build_module, dev_bf16

SYNC = os.environ.get("FLAKE_SYNC", "none")
ONLY_B = os.environ.get("FLAKE_ONLY_B", "0") == "1"          # run only the eager-backward module (is the other module's work part of it?)


def one(name, how):
    case = cases.build_case(name)
    ma, mb = build_module(case).train(), build_module(case).train()
    mb.graph_backward = False
    if os.environ.get("FLAKE_ONE_STREAM", "0") == "1":
        ma.overlap_stages = mb.overlap_stages = False               # the operator-by-operator forward on ONE stream
    if os.environ.get("FLAKE_NO_GSTORE", "0") == "1":
        ma.share_global_state = mb.share_global_state = False       # the backward streams the tokens again instead of reading the forward's store
    if os.environ.get("FLAKE_NO_CTX", "0") == "1":
        ma.share_window_contexts = mb.share_window_contexts = False
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    gen = torch.Generator(device="cuda").manual_seed(11)
    R = None
    for stepno in range(5):
        outs = []
        order = (mb,) if ONLY_B else ((mb, ma) if os.environ.get("FLAKE_ORDER", "ab") == "ba" else (ma, mb))
        for m in order:
            m.zero_grad(set_to_none=True)
            out = m(ff, fe, g, case.modal, None)
            if SYNC == "after_fwd":
                torch.cuda.synchronize()
            if R is None:
                R = torch.randn(out.shape, device="cuda", generator=gen).to(out.dtype)
            out.backward(R)
            if SYNC == "after_bwd":
                torch.cuda.synchronize()
            outs.append(out.detach().clone())
        for which, m, o in zip(("ma" if m_ is ma else "mb" for m_ in order), order, outs):
            with torch.no_grad():
                r = m(ff, fe, g, case.modal, None)                  # the same weights, again: a forward that differs from its own repeat is the flake
            if not torch.equal(r, o):
                d = (o.float() - r.float()).abs()
                return f"step {stepno}: {which}'s training forward differs from its repeat by {float(d.max()):.3e}, rows {torch.nonzero(d.amax(1) > 0).flatten()[:6].tolist()}"
        with torch.no_grad():
            for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
                delta = (torch.randn(pa.shape, device="cuda", generator=gen) * 0.01).to(pa.dtype)
                if how == "add_":
                    pa.add_(delta); pb.add_(delta)
                else:
                    new = (pa.detach().float() + delta.float()).to(pa.dtype)
                    pa.data.copy_(new); pb.data.copy_(new)
        if SYNC == "after_update":
            torch.cuda.synchronize()
    return None


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
runs = fails = 0
t0 = time.time()
for rep in range(reps):
    for name in ("G7_fine", "G6_coarse"):
        for how in ("add_", "data_copy_"):
            for _ in range(3):
                runs += 1
                msg = one(name, how)
                if msg:
                    fails += 1
                    print(f"FAIL rep {rep} {name} {how}: {msg}", flush=True)
print(f"FLAKE_LOOP2 order={os.environ.get('FLAKE_ORDER', 'ab')} one_stream={os.environ.get('FLAKE_ONE_STREAM', '0')} no_gstore={os.environ.get('FLAKE_NO_GSTORE', '0')} no_ctx={os.environ.get('FLAKE_NO_CTX', '0')} sync={SYNC} only_b={ONLY_B} nofence={os.environ.get('HICOM_EVENT_NOFENCE', '1')}: failures={fails} of {runs} runs in {time.time() - t0:.0f} s", flush=True)
