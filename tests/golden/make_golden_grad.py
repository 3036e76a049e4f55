"""Generates tests/golden/golden_grad_v1.npz (+ golden_grad_v2.npz: d frames_embed of the guide-off recipes): parameter (and, for the direct recipe, guide_embed / frames_embed) gradients of the REFERENCE (imported in place from
/root/reference through oracle/ref_shim.py, float32 autograd) for loss = sum(out * R), R a fixed synthetic cotangent.

Run in the build container only:  python tests/golden/make_golden_grad.py
Per parameter only 512 sampled entries + (sum, abs-sum) are stored (the projection matrices are 1152 x 1152).
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

from oracle import ref_shim            # noqa: E402
from hicom_amd import synth            # noqa: E402
import cases                            # noqa: E402

GRAD_CASES = ["G1_direct_T8", "G4b_image_newline", "G9_grid", "G9_frame", "G10_peaky_direct", "G9_local_only", "G9_global_only",
              "G3_direct_T7", "G2_off_T8", "G2b_off_string", "G10b_peaky_off", "G12_clip768_direct", "G12b_clip768_off", "G9_local22",
              # the second released recipe (k / v adaptors over all tokens) and coarse (FiLM) injection, incl. a mixed override
              "G5_adaptkv", "G6_coarse", "G7b_guide_override",
              # fine (64 text tokens) injection; every adaptor at once (adapt q / k / v / guide, coarse)
              "G7_fine", "G5b_adaptqkvg_off"]


def cotangent(name, shape):
    return synth.normal_like(tuple(shape), synth.seed_of(name + ":cotangent"))


def sample_positions(n, count=512):
    k = np.arange(count, dtype=np.int64)
    return (k * 2654435761 + 12345) % n


ANYRES_CASES = ["G9_anyres", "G9_anyres_nobase"]        # dict inputs of an anyres image (reference projector.py:679-689), into v2


def store(dst, name, items):
    for k, gr in items:
        if gr is None:
            dst[f"{name}/{k}/none"] = np.zeros(1, dtype=np.uint8)
            continue
        v = gr.detach().numpy().astype(np.float32).reshape(-1)
        pos = sample_positions(v.size)
        dst[f"{name}/{k}/samples"] = v[pos]
        dst[f"{name}/{k}/sums"] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum(),
                                            np.abs(v).max()], dtype=np.float64)


def anyres_grads(proj, blobs2):
    for name in ANYRES_CASES:
        case = cases.build_case(name)
        torch.manual_seed(0)
        module = proj.build_vision_projector(case.cfg).float().train()
        module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        a = case.anyres
        fdict = {"base": None if a["no_base"] else t(case.ff)[0], "patch": t(a["patch_ff"])}
        edict = {"base": None if a["no_base"] else t(case.fe)[0], "patch": t(a["patch_fe"])}
        nl = torch.nn.Parameter(t(case.newline)) if case.newline is not None else None
        out = module(fdict, edict, t(case.g), case.modal, nl)
        R = torch.from_numpy(cotangent(name, out.shape))
        (out * R).sum().backward()
        items = [(k, p.grad) for k, p in module.named_parameters()]
        if nl is not None:
            items.append(("image_newline", nl.grad))
        store(blobs2, name, items)
        blobs2[f"{name}/out_shape"] = np.array(out.shape, dtype=np.int64)
        print(name, "(anyres) params with grad:", sum(1 for _, gr in items if gr is not None), "/", len(items))


CLIP_LOCAL_CASES = [k for k in cases.CASES_EXTRA if k not in cases.CASES_CLIP_GLOBAL]     # clip-scale on the local stage under autograd, into golden_grad_v3.npz
CLIP_GLOBAL_CASES = list(cases.CASES_CLIP_GLOBAL)


def clip_global_grads(proj, blobs3):
    """GlobalCompressor.forward(ff, fe, g, modal, logit_scale, logit_bias) of the reference with the two logits as leaf tensors (reference
    projector.py:634-646 -> :184-191; HIComProjector.forward passes its `global_logit_scale` / `global_logit_bias` parameters there): gradients of
    the compressor's parameters, of both logits and of guide_embed for loss = sum(out * R)."""
    for name in CLIP_GLOBAL_CASES:
        case = cases.build_case(name)
        torch.manual_seed(0)
        module = proj.build_vision_projector(case.cfg).float().train()
        module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        ff, fe, g = t(case.ff), t(case.fe), t(case.g)
        uses_guide = getattr(case.cfg, "use_guide", None) in ("direct", "coarse", "fine")
        if uses_guide:
            g = g.clone().requires_grad_(True)
        ls = torch.tensor(float(case.logit["glob"][0]), requires_grad=True)
        lb = torch.tensor(float(case.logit["glob"][1]), requires_grad=True)
        out = module.global_compressor(ff, fe, g, case.modal, ls, lb)
        out = out.reshape(-1, out.shape[-1])
        R = torch.from_numpy(cotangent(name, out.shape))
        (out * R).sum().backward()
        items = [(k, p.grad) for k, p in module.named_parameters()]
        items += [("global_logit_scale", ls.grad.reshape(1)), ("global_logit_bias", (lb.grad if lb.grad is not None else torch.zeros(())).reshape(1))]
        if uses_guide:
            items.append(("__guide_embed__", g.grad))
        store(blobs3, name, items)
        blobs3[f"{name}/out_shape"] = np.array(out.shape, dtype=np.int64)
        blobs3[f"{name}/out"] = out.detach().numpy().astype(np.float32)
        print(name, "(clip-scale global) d logit_scale", float(ls.grad), "params with grad:", sum(1 for _, gr in items if gr is not None), "/", len(items))


def clip_local_grads(proj, blobs3):
    """LocalCompressor.forward(ff, fe, g, modal, logit_scale, logit_bias) of the reference with the two logits as leaf tensors
    (reference projector.py:524-559; HIComProjector.forward passes its `local_logit_scale` / `local_logit_bias` parameters there,
    :691-692): gradients of the compressor's parameters, of both logits, of guide_embed and of frames_embed for loss = sum(out * R)."""
    for name in CLIP_LOCAL_CASES:
        case = cases.build_case(name)
        torch.manual_seed(0)
        module = proj.build_vision_projector(case.cfg).float().train()
        module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        ff, fe, g = t(case.ff), t(case.fe).clone().requires_grad_(True), t(case.g)
        uses_guide = getattr(case.cfg, "use_guide", None) in ("direct", "coarse", "fine")
        if uses_guide:
            g = g.clone().requires_grad_(True)
        ls = torch.tensor(float(case.logit["local"][0]), requires_grad=True)
        lb = torch.tensor(float(case.logit["local"][1]), requires_grad=True)
        out = module.local_compressor(ff, fe, g, case.modal, ls, lb)
        out = out.reshape(-1, out.shape[-1])                               # (= post_process_visual_feature's flatten, mm_utils.py:96-97)
        R = torch.from_numpy(cotangent(name, out.shape))
        (out * R).sum().backward()
        items = [(k, p.grad) for k, p in module.named_parameters()]
        items += [("local_logit_scale", ls.grad.reshape(1)), ("local_logit_bias", lb.grad.reshape(1)), ("__frames_embed__", fe.grad)]
        if uses_guide:
            items.append(("__guide_embed__", g.grad))
        store(blobs3, name, items)
        blobs3[f"{name}/out_shape"] = np.array(out.shape, dtype=np.int64)
        blobs3[f"{name}/out"] = out.detach().numpy().astype(np.float32)
        print(name, "(clip-scale local) d logit_scale", float(ls.grad), "d logit_bias", float(lb.grad), "params with grad:",
              sum(1 for _, gr in items if gr is not None), "/", len(items))


FF_GRAD_CASES = ["G1_direct_T8", "G9_local_only", "G9_global_only", "G4_direct_T1", "G10_peaky_direct", "G12_clip768_direct",
                 # every other injection mode: guide off (pooled per-window queries, 32 learnable global queries), coarse, fine, a mixed override
                 "G2_off_T8", "G2b_off_string", "G6_coarse", "G7_fine", "G7b_guide_override", "G12b_clip768_off",
                 # k / v adaptors (the second released recipe) and every adaptor at once
                 "G5_adaptkv", "G5b_adaptqkvg_off",
                 # window partitions that do not divide the axes (T = 7 under a temporal kernel of 4; 7 x 7 under 2 x 2: the trailing windows overlap)
                 "G3_direct_T7", "G9_local22"]


def ff_grads(proj, blobs3):
    """d frames_feature of the direct recipe (`pure_vision_model` trains the tower body, reference train.py:712-715): the reference's
    autograd with frames_feature as a leaf (frames_embed and the guide leaves too, so that every input gradient of one backward is
    on record), loss = sum(out * R) with the case's usual cotangent."""
    for name in FF_GRAD_CASES:
        case = cases.build_case(name)
        torch.manual_seed(0)
        module = proj.build_vision_projector(case.cfg).float().train()
        module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        ff, fe, g = t(case.ff).clone().requires_grad_(True), t(case.fe).clone().requires_grad_(True), t(case.g).clone().requires_grad_(True)
        out = module(ff, fe, g, case.modal, None)
        (out * torch.from_numpy(cotangent(name, out.shape))).sum().backward()
        store(blobs3, name, [("__frames_feature__", ff.grad)])
        # ... and with frames_embed = None (the keys are the value rows, projector.py:532)
        ff2 = t(case.ff).clone().requires_grad_(True)
        module.zero_grad()
        out2 = module(ff2, None, t(case.g), case.modal, None)
        (out2 * torch.from_numpy(cotangent(name, out2.shape))).sum().backward()
        store(blobs3, name + "@nofe", [("__frames_feature__", ff2.grad)])
        print(name, "d frames_feature max", float(ff.grad.abs().max()), " without frames_embed", float(ff2.grad.abs().max()))


def main():
    proj, _ = ref_shim.load()
    blobs3 = {}
    clip_local_grads(proj, blobs3)
    ff_grads(proj, blobs3)
    clip_global_grads(proj, blobs3)
    blobs = {}
    blobs2 = {}     # golden_grad_v2.npz: d frames_embed of the recipes that do NOT inject the guide (guide off: frames_embed are the window
    #                 keys), and the parameter gradients of anyres dict inputs
    anyres_grads(proj, blobs2)
    for name in GRAD_CASES:
        case = cases.build_case(name)
        torch.manual_seed(0)
        module = proj.build_vision_projector(case.cfg).float().train()
        module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}, strict=True)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        ff, fe, g = t(case.ff), t(case.fe), t(case.g)
        # stage 3 of the reference's script also trains what produces frames_embed and guide_embed (train.py:717-726):
        # record their gradients for the direct, coarse and fine recipes
        direct = getattr(case.cfg, "use_guide", None) in ("direct", "coarse", "fine") and g is not None
        if direct:
            g = g.clone().requires_grad_(True)
            if fe is not None:
                fe = fe.clone().requires_grad_(True)
        elif fe is not None:
            fe = fe.clone().requires_grad_(True)
        nl = None
        if case.newline is not None:
            nl = torch.nn.Parameter(t(case.newline))
        out = module(ff, fe, g, case.modal, nl)
        R = torch.from_numpy(cotangent(name, out.shape))
        (out * R).sum().backward()
        items = [(k, p.grad) for k, p in module.named_parameters()]
        if nl is not None:
            items.append(("image_newline", nl.grad))
        if direct:
            items.append(("__guide_embed__", g.grad))
            if fe is not None:
                items.append(("__frames_embed__", fe.grad))
        elif fe is not None:
            items.append(("__frames_embed__", fe.grad))
        for k, gr in items:
            dst = blobs2 if (k == "__frames_embed__" and not direct) else blobs
            if gr is None:
                dst[f"{name}/{k}/none"] = np.zeros(1, dtype=np.uint8)
                continue
            v = gr.detach().numpy().astype(np.float32).reshape(-1)
            pos = sample_positions(v.size)
            dst[f"{name}/{k}/samples"] = v[pos]
            dst[f"{name}/{k}/sums"] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum(),
                                                np.abs(v).max()], dtype=np.float64)
        print(name, "params with grad:", sum(1 for _, gr in items if gr is not None), "/", len(items))
    for fname, data in (("golden_grad_v1.npz", blobs), ("golden_grad_v2.npz", blobs2), ("golden_grad_v3.npz", blobs3)):
        path = os.path.join(HERE, fname)
        np.savez_compressed(path, **data)
        print("wrote", path, os.path.getsize(path), "bytes", len(data), "arrays")


if __name__ == "__main__":
    main()
