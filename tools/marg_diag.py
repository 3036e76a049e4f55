"""Dev: local / global rows of a few cases against the oracle, for A/B of HICOM_RING_MARG (set in the environment of this process)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import cases
from types import SimpleNamespace
from gpu_util import run_native
from oracle_util import run_oracle
from hicom_amd import synth
from oracle import hicom_oracle as orc
print("HICOM_RING_MARG =", os.environ.get("HICOM_RING_MARG"))
def show(name, case):
    want = run_oracle(case)["out"].numpy()
    got = run_native(case)["out"].float().cpu().numpy()
    d = np.abs(got - want)
    print(f"{name:28s} out {got.shape}  local rows max-abs {d[:-32].max():.3e}  global rows {d[-32:].max():.3e}  max|out| {np.abs(want).max():.3f}", flush=True)
for name in ("G1_direct_T8", "G13_outlier_direct", "G13b_outlier_off", "G11_c1_shape", "G13c_outlier_c1", "G10_peaky_direct"):
    show(name, cases.build_case(name))
for (T, h, w) in ((12, 9, 6), (8, 6, 6), (4, 6, 6), (8, 9, 9), (16, 6, 6)):
    tag = f"diag{T}{h}{w}"
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "max_num_frames": 16})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
    x = synth.synth_inputs(T, h, w, cases.D, tag=tag)
    case = SimpleNamespace(cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"], modal="video", newline=None, anyres=None, logit=None)
    show(f"direct T={T} {h}x{w}", case)
