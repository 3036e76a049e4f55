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


def oracle_grads(case, cot, inputs=()):
    """Gradients of sum(oracle_forward(case) * cot) by torch autograd THROUGH the oracle (fp32): {parameter name | "image_newline" |
    "__guide_embed__" | "__frames_embed__": tensor or None}.  The oracle is a torch restatement of the reference's arithmetic, pinned
    by the forward fixtures; its derivative is pinned to the reference's own autograd by tests/test_oracle_golden.py
    (test_oracle_autograd_reproduces_the_reference_gradients), so it can referee the HIP backward on configurations no fixture holds."""
    sd = {k: to_t(v).requires_grad_(True) for k, v in case.sd.items()}
    ff, fe, g, nl = to_t(case.ff), to_t(case.fe), to_t(case.g), to_t(case.newline)
    if "__guide_embed__" in inputs and g is not None:
        g.requires_grad_(True)
    if "__frames_embed__" in inputs and fe is not None:
        fe.requires_grad_(True)
    if nl is not None:
        nl.requires_grad_(True)
    if case.anyres is not None:
        a = case.anyres
        f_in = {"base": None if a["no_base"] else ff[0], "patch": to_t(a["patch_ff"])}
        e_in = {"base": None if a["no_base"] else fe[0], "patch": to_t(a["patch_fe"])}
    else:
        f_in, e_in = ff, fe
    out = orc.projector_forward(case.cfg, sd, f_in, e_in, g, case.modal, nl)
    (out * (cot if isinstance(cot, torch.Tensor) else torch.from_numpy(cot))).sum().backward()
    grads = {k: v.grad for k, v in sd.items()}
    if nl is not None:
        grads["image_newline"] = nl.grad
    if "__guide_embed__" in inputs:
        grads["__guide_embed__"] = None if g is None else g.grad
    if "__frames_embed__" in inputs:
        grads["__frames_embed__"] = None if fe is None else fe.grad
    return out.detach(), grads
