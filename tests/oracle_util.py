"""Helpers to run the CPU oracle on a golden case (test infrastructure)."""
import numpy as np
import torch

from oracle import hicom_oracle as orc


def to_t(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def run_oracle(case, dtype=torch.float32):
    sd = {k: to_t(v, dtype) for k, v in case.sd.items()}
    ff, fe, g, nl = to_t(case.ff, dtype), to_t(case.fe, dtype), to_t(case.g, dtype), to_t(case.newline, dtype)
    if case.logit is not None:
        spec = orc.parse_projector_type(case.cfg.mm_projector_type)
        ls, lb = (torch.tensor(v, dtype=dtype) for v in case.logit["local"])
        gs, gb = (torch.tensor(v, dtype=dtype) for v in case.logit["glob"])
        lmode = orc.resolve_guide_mode(case.cfg, spec["local"]["force_use_guide"])
        gmode = orc.resolve_guide_mode(case.cfg, spec["global"]["force_use_guide"])
        lo = orc.local_forward(spec["local"], lmode, sd, "local_compressor", ff, fe, g, case.modal, ls, lb)
        go = orc.global_forward(spec["global"], gmode, sd, "global_compressor", ff, g, gs, gb)
        return {"local": lo, "global": go}
    if case.anyres is not None:
        a = case.anyres
        fdict = {"base": None if a["no_base"] else ff[0], "patch": to_t(a["patch_ff"], dtype)}
        edict = {"base": None if a["no_base"] else fe[0], "patch": to_t(a["patch_fe"], dtype)}
        return {"out": orc.projector_forward(case.cfg, sd, fdict, edict, g, case.modal, nl)}
    return {"out": orc.projector_forward(case.cfg, sd, ff, fe, g, case.modal, nl)}
