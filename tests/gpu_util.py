"""Helpers that run the HIP product path on a golden case (GPU tests / smoke)."""
import numpy as np
import torch

import hicom_amd


def dev_bf16(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).cuda()


def build_module(case, fp32_out=True):
    m = hicom_amd.build_vision_projector(case.cfg)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=False)
    # (a clip-scale projector also owns `*_logit_scale` / `*_logit_bias`, reference projector.py:655-670: the synthetic state dicts do
    # not carry them, the tests give them through set_clip_logits())
    assert not unexpected and all("_logit_" in k for k in missing), (missing, unexpected)
    m = m.to(torch.bfloat16).cuda().eval()
    m.return_fp32 = fp32_out
    for sub in (m.local_compressor, m.global_compressor):
        if sub is not None:
            sub.return_fp32 = fp32_out
    return m


def run_native(case, fp32_out=True):
    m = build_module(case, fp32_out)
    ff, fe, g, nl = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), dev_bf16(case.newline)
    with torch.no_grad():
        if case.logit is not None:
            ls, lb = (torch.tensor(v, device="cuda") for v in case.logit["local"])
            lo = m.local_compressor(ff, fe, g, case.modal, ls, lb)
            return {"local": lo}
        if case.anyres is not None:
            a = case.anyres
            fdict = {"base": None if a["no_base"] else ff[0], "patch": dev_bf16(a["patch_ff"])}
            edict = {"base": None if a["no_base"] else fe[0], "patch": dev_bf16(a["patch_fe"])}
            out = m(fdict, edict, g, case.modal, nl)
        else:
            out = m(ff, fe, g, case.modal, nl)
    torch.cuda.synchronize()
    return {"out": out}
