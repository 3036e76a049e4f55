"""Generates tests/golden/golden_v1.npz by running the REFERENCE (imported in place from
/root/reference through oracle/ref_shim.py) in float32 on the case matrix of cases.py.

Run in the build container only:  python tests/golden/make_golden.py
The reference never travels: only its fp32 outputs (a few hundred KB) are committed.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import ref_shim, hicom_oracle as orc   # noqa: E402
import cases                                        # noqa: E402


def run_reference(proj, case, dtype=torch.float32):
    cfg = case.cfg
    torch.manual_seed(0)
    module = proj.build_vision_projector(cfg).float().eval()
    ref_sd = module.state_dict()
    want = orc.param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in ref_sd.items()}
    assert got == want, f"{case.name}: parameter schema mismatch\n ref-only: {set(got)-set(want)}\n ours-only: {set(want)-set(got)}"
    module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    ff, fe, g, nl = t(case.ff), t(case.fe), t(case.g), t(case.newline)
    if dtype != torch.float32:                 # the reference's own inference arithmetic (model_init(torch_dtype=bf16)): context figure only
        module = module.to(dtype)
        with torch.no_grad():
            return {"out": module(ff, fe, g, case.modal, nl).float().numpy()}
    with torch.no_grad():
        if case.logit is not None:
            ls, lb = (torch.tensor(v) for v in case.logit["local"])
            gs, gb = (torch.tensor(v) for v in case.logit["glob"])
            lo = module.local_compressor(ff, fe, g, case.modal, ls, lb)
            go = module.global_compressor(ff, fe, g, case.modal, gs, gb)
            return {"local": lo.numpy(), "global": go.numpy()}
        if case.anyres is not None:
            a = case.anyres
            fdict = {"base": None if a["no_base"] else ff[0], "patch": t(a["patch_ff"])}
            edict = {"base": None if a["no_base"] else fe[0], "patch": t(a["patch_fe"])}
            return {"out": module(fdict, edict, g, case.modal, nl).numpy()}
        return {"out": module(ff, fe, g, case.modal, nl).numpy()}


def main():
    proj, _ = ref_shim.load()
    blobs = {}
    for name in cases.CASES:
        case = cases.build_case(name)
        if case.expect_raises:
            try:
                run_reference(proj, case)
            except Exception as e:  # noqa: BLE001
                blobs[f"{name}/raised"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)
                print(f"{name}: raised {type(e).__name__}: {str(e)[:80]}")
                continue
            raise AssertionError(f"{name}: reference did not raise")
        outs = run_reference(proj, case)
        for k, v in outs.items():
            v = v.astype(np.float32)
            if case.sampled:
                r, c = cases.sample_index(*v.shape)
                blobs[f"{name}/{k}_samples"] = v[r, c]
                blobs[f"{name}/{k}_shape"] = np.array(v.shape, dtype=np.int64)
                blobs[f"{name}/{k}_sum"] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
            else:
                blobs[f"{name}/{k}"] = v
            print(f"{name}/{k}: shape {v.shape} absmax {np.abs(v).max():.4f}")
        if case.ref_bf16:
            # the reference module itself run in bf16 on the same inputs: how far its OWN inference arithmetic is from its fp32 result --
            # stored beside the fp32 output as the context figure of the heavy-tailed cases' tolerance
            lo = run_reference(proj, case, torch.bfloat16)["out"].astype(np.float32)
            d = float(np.abs(lo - outs["out"].astype(np.float32)).max())
            blobs[f"{name}/ref_bf16_max_abs"] = np.array([d, float(np.abs(outs["out"]).max())], dtype=np.float64)
            print(f"{name}: reference bf16 vs fp32 max-abs {d:.3e} (max |out| {np.abs(outs['out']).max():.3f})")
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **blobs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
