"""Golden-vector case matrix: shared by make_golden.py and the tests.

Tensors stay small by shrinking the grid and the LLM width, never D (=1152 is hard-wired in
the reference, projector.py:408).  Inputs/weights are regenerated from `hicom_amd.synth`
(pure integer hash), so only the reference's fp32 OUTPUTS are stored in golden_v1.npz.
"""
from __future__ import annotations

D = 1152
TOWER = "google/siglip-so400m-patch14-384"

# name -> dict(cfg=config overrides, T,h,w, modal, newline, guide_len, anyres, logit, peaky, expect_raises)
CASES = {
    # release recipe: direct guide, local43 + global32, spatial_unpad / no_token
    "G1_direct_T8": dict(cfg=dict(), T=8, h=6, w=6),
    # guide off: pooled per-window query, 32 distinct learnable queries
    "G2_off_T8": dict(cfg=dict(use_guide=None), T=8, h=6, w=6),
    "G2b_off_string": dict(cfg=dict(use_guide="off", mm_projector_type="local43_global32"), T=4, h=6, w=9),
    # non-divisible T: overlapping windows (T=7) / reference raises (T=5)
    "G3_direct_T7": dict(cfg=dict(), T=7, h=6, w=6),
    "G3b_direct_T5_raises": dict(cfg=dict(), T=5, h=6, w=6, expect_raises="RuntimeError"),
    "G3d_direct_T2": dict(cfg=dict(), T=2, h=6, w=6),
    "G3e_off_T3_h2": dict(cfg=dict(use_guide=None), T=3, h=2, w=6),
    "G3c_off_T10_hw75": dict(cfg=dict(use_guide=None), T=10, h=7, w=5),
    # t-kernel 1: single frame video, image modality with a newline token
    "G4_direct_T1": dict(cfg=dict(), T=1, h=6, w=6),
    "G4b_image_newline": dict(cfg=dict(), T=1, h=6, w=6, modal="image", newline=True),
    "G4c_image_T2_raises": dict(cfg=dict(), T=2, h=6, w=6, modal="image", expect_raises="Exception"),
    # adaptive K/V blends with alpha = 0.5
    "G5_adaptkv": dict(cfg=dict(mm_projector_type="local43_adaptkv_global32"), T=4, h=6, w=6),
    "G5b_adaptqkvg_off": dict(cfg=dict(mm_projector_type="local43_adaptqkvg_global32_adaptg", use_guide="coarse"),
                              T=4, h=6, w=6),
    # coarse (FiLM) and fine (64 text tokens) injection
    "G6_coarse": dict(cfg=dict(use_guide="coarse"), T=8, h=6, w=6),
    "G7_fine": dict(cfg=dict(use_guide="fine"), T=4, h=6, w=6, guide_len=64),
    "G7b_guide_override": dict(cfg=dict(use_guide="direct", mm_projector_type="local43_guidecoarse_global32_guideoff"),
                               T=4, h=6, w=6),
    # clip-scale path (direct compressor calls with logit tensors)
    "G8_clip_scale": dict(cfg=dict(), T=4, h=6, w=6, logit=dict(local=(2.0, -3.0), glob=(1.5, -2.0))),
    # ... with an injector: the guide is normalised BEFORE it is injected (projector.py:527-529 in front of :542)
    "G8b_clip_coarse": dict(cfg=dict(use_guide="coarse"), T=4, h=6, w=6, logit=dict(local=(2.0, -3.0), glob=(1.5, -2.0))),
    "G8c_clip_fine": dict(cfg=dict(use_guide="fine"), T=4, h=6, w=6, guide_len=64, logit=dict(local=(2.0, -3.0), glob=(1.5, -2.0))),
    "G8e_clip_adaptkv": dict(cfg=dict(mm_projector_type="local43_adaptkv_global32"), T=4, h=6, w=6,
                             logit=dict(local=(2.0, -3.0), glob=(1.5, -2.0))),
    "G8d_clip_direct_adaptg": dict(cfg=dict(mm_projector_type="local43_adaptg_global32_adaptg"), T=4, h=6, w=6,
                                   logit=dict(local=(2.0, -3.0), glob=(1.5, -2.0))),
    # packing variants
    "G9_grid": dict(cfg=dict(mm_newline_position="grid"), T=8, h=6, w=6, newline=True),
    "G9_frame": dict(cfg=dict(mm_newline_position="frame"), T=8, h=6, w=6, newline=True),
    "G9_one_token": dict(cfg=dict(mm_newline_position="one_token"), T=8, h=6, w=6, newline=True),
    "G9_flat": dict(cfg=dict(mm_patch_merge_type="flat"), T=8, h=6, w=6),
    "G9_anyres": dict(cfg=dict(), T=1, h=6, w=6, modal="image", newline=True, anyres=dict(ph=6, pw=12)),
    "G9_anyres_nobase": dict(cfg=dict(), T=1, h=6, w=6, modal="image", newline=True,
                             anyres=dict(ph=9, pw=6, no_base=True)),
    # only one compressor
    "G9_local_only": dict(cfg=dict(mm_projector_type="local43"), T=4, h=6, w=6),
    "G9_global_only": dict(cfg=dict(mm_projector_type="global32"), T=4, h=6, w=6),
    "G9_local22": dict(cfg=dict(mm_projector_type="local22_global16", use_guide=None), T=4, h=7, w=7),
    # peaky softmax: exercises online-softmax merging
    "G10_peaky_direct": dict(cfg=dict(), T=8, h=6, w=6, peaky=12.0, in_scale=2.0),
    "G10b_peaky_off": dict(cfg=dict(use_guide=None), T=8, h=6, w=6, peaky=12.0, in_scale=2.0),
    # CLIP tower branch (projector.py:410-412,572-574): qk_dim = 768, 24x24 grid side, 6 heads
    "G12_clip768_direct": dict(cfg=dict(mm_vision_tower="openai/clip-vit-large-patch14-336", mm_hidden_size=768), T=4, h=6, w=6, dim=768),
    "G12b_clip768_off": dict(cfg=dict(mm_vision_tower="openai/clip-vit-large-patch14-336", mm_hidden_size=768, use_guide=None),
                             T=8, h=6, w=6, dim=768),
    # heavy-tailed channels (round-5 verdict): 12 channels x 60 with a non-zero mean in both visual tensors -- the statistics of real SigLIP
    # hidden_states[-2] (reference encoder.py:253-259), where the fp16 activation planes of the HIP path round at 2^-12 of |ctx|
    "G13_outlier_direct": dict(cfg=dict(), T=8, h=6, w=6, outliers=(12, 60.0), ref_bf16=True),
    "G13b_outlier_off": dict(cfg=dict(use_guide=None), T=8, h=6, w=6, outliers=(12, 60.0), ref_bf16=True),
    "G13c_outlier_c1": dict(cfg=dict(hidden_size=896), T=4, h=27, w=27, sampled=True, outliers=(12, 60.0), ref_bf16=True),
    # C1 shape: 27x27 grid, T=4, H=896 -- stored as sampled outputs + checksum only
    "G11_c1_shape": dict(cfg=dict(hidden_size=896), T=4, h=27, w=27, sampled=True),
}

# Cases of the GRADIENT fixtures only (make_golden_grad.py -> golden_grad_v3.npz; not part of golden_v1's forward matrix): clip-scale on
# the local stage under autograd (reference projector.py:527-529, :549; trainable under `attn_scale`, train.py:730-733), taken through
# direct LocalCompressor calls with logit tensors as the forward fixtures G8* do (the reference's projector cannot be CONSTRUCTED with
# use_clip_scale offline: it reads the logits from the SigLIP checkpoint on the hub, :660-670).  Local-only projector types, so that
# HIComProjector.forward of the build is exactly that call + the flatten packing.
CASES_EXTRA = {
    "G8f_clip_local_direct": dict(cfg=dict(mm_projector_type="local43"), T=8, h=6, w=6, logit=dict(local=(2.0, -3.0), glob=None)),
    "G8g_clip_local_off": dict(cfg=dict(mm_projector_type="local43", use_guide=None), T=4, h=6, w=6, logit=dict(local=(1.5, 0.5), glob=None)),
    "G8h_clip_local_coarse": dict(cfg=dict(mm_projector_type="local43", use_guide="coarse"), T=4, h=6, w=9, logit=dict(local=(2.0, -3.0), glob=None)),
    "G8i_clip_local_fine": dict(cfg=dict(mm_projector_type="local43", use_guide="fine"), T=4, h=6, w=6, guide_len=9, logit=dict(local=(2.5, 1.0), glob=None)),
}
# clip-scale on the GLOBAL stage under autograd (golden_grad_v3.npz: clip_global_grads): global-only projectors, every injection mode
CASES_CLIP_GLOBAL = {
    "G8j_clip_global_direct": dict(cfg=dict(mm_projector_type="global32"), T=4, h=6, w=6, logit=dict(local=None, glob=(1.5, -2.0))),
    "G8k_clip_global_off": dict(cfg=dict(mm_projector_type="global32", use_guide=None), T=4, h=6, w=6, logit=dict(local=None, glob=(2.0, 0.5))),
    "G8l_clip_global_coarse": dict(cfg=dict(mm_projector_type="global32", use_guide="coarse"), T=3, h=6, w=5, logit=dict(local=None, glob=(1.0, -1.0))),
    "G8m_clip_global_fine": dict(cfg=dict(mm_projector_type="global32", use_guide="fine"), T=4, h=6, w=6, guide_len=9, logit=dict(local=None, glob=(2.5, 1.0))),
}
CASES_EXTRA.update(CASES_CLIP_GLOBAL)

DEFAULT_CFG = dict(mm_projector_type="local43_global32_coarse", use_guide="direct", use_clip_scale="",
                   mm_patch_merge_type="spatial_unpad", mm_newline_position="no_token",
                   mm_vision_tower=TOWER, mm_hidden_size=D, hidden_size=64, max_num_frames=16)


def case_config(name: str) -> dict:
    cfg = dict(DEFAULT_CFG)
    cfg.update((CASES.get(name) or CASES_EXTRA[name])["cfg"])
    return cfg


def sample_index(n_rows: int, n_cols: int, count: int = 256):
    """Deterministic (row, col) sample positions for the checksum-only case."""
    import numpy as np
    k = np.arange(count, dtype=np.int64)
    return (k * 7919 + 13) % n_rows, (k * 104729 + 7) % n_cols


def build_case(name: str):
    """Regenerates (config, state_dict, inputs) of a case as float32 numpy arrays (bf16-representable)."""
    import os, sys
    from types import SimpleNamespace
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from hicom_amd import synth
    from oracle import hicom_oracle as orc

    c = CASES.get(name) or CASES_EXTRA[name]
    cfg = SimpleNamespace(**case_config(name))
    shapes = orc.param_shapes(cfg)
    sd = synth.synth_state_dict(shapes, tag=name, peaky=c.get("peaky", 1.0))
    T, h, w = c["T"], c["h"], c["w"]
    dim = c.get("dim", D)
    x = synth.synth_inputs(T, h, w, dim, tag=name, guide_len=c.get("guide_len", 0),
                           scale=c.get("in_scale", 1.0), outliers=c.get("outliers"))
    newline = synth.normal_like((cfg.hidden_size,), synth.seed_of(name + ":newline")) if c.get("newline") else None
    anyres = None
    if c.get("anyres"):
        a = c["anyres"]
        p = synth.synth_inputs(1, a["ph"], a["pw"], dim, tag=name + ":patch")
        anyres = dict(patch_ff=p["ff"][0], patch_fe=p["fe"][0], no_base=a.get("no_base", False))
    return SimpleNamespace(name=name, cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"],
                           modal=c.get("modal", "video"), newline=newline, anyres=anyres,
                           logit=c.get("logit"), expect_raises=c.get("expect_raises"),
                           sampled=c.get("sampled", False), ref_bf16=c.get("ref_bf16", False))
