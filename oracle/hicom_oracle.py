"""CPU oracle for the HICom hybrid-level video-token compressor.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`hicom_amd/`) may
import this file; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the timed CPU
baseline.  The product path is HIP-only and fails loudly without its `.so`.

What it is: a functional (state-dict driven) restatement, in plain PyTorch CPU
ops with explicit index arithmetic, of the reference's

    hicom/model/projector.py   HIComProjector / LocalCompressor /
                               GlobalCompressor / GuideInjector /
                               MultiheadAttention / build_vision_projector
    hicom/mm_utils.py:92-140   post_process_visual_feature

Every function cites the reference lines it follows.  It evaluates in the
dtype of its inputs (tests use float32 on bf16-representable values, and
float64 for tighter algebra checks).

Parity pinning: the reference ships no tests or golden vectors (SURVEY.md §4,
§8c), so this oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run
in the build container through `oracle/ref_shim.py`; the vectors and the
script that made them are committed under `tests/golden/`
(`tests/golden/make_golden.py`), and `tests/test_oracle_golden.py` checks the
oracle against them on every CPU run.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

LN_EPS = 1e-6  # projector.py:318,403,565  norm_layer=partial(nn.LayerNorm, eps=1e-6)


# --------------------------------------------------------------------------
# projector-type string parsing            (projector.py:231-304)
# --------------------------------------------------------------------------
def parse_projector_type(projector_type: str) -> dict:
    """Restates build_vision_projector's substring parser (projector.py:246-302).

    Returns {"local": None | {...}, "global": None | {...}}.
    """
    spec = {"local": None, "global": None}
    if "local" in projector_type:
        phase = projector_type.split("local")[-1].split("global")[0]
        digits = ""
        for ch in phase:
            if ch.isdigit():
                digits += ch
            else:
                break
        kt = int(digits[0])                       # :255
        if len(digits) == 2:
            ks = int(digits[1])                   # :257
        elif len(digits) == 3:
            ks = int(digits[1:3])                 # :259
        else:
            raise UnboundLocalError("spatial_kernel_size")  # the reference leaves it unbound
        flags = dict(adapt_q=False, adapt_k=False, adapt_v=False, adapt_guide=False)
        if "adapt" in phase:
            for ch in phase.split("adapt")[-1]:   # :263-273
                if ch == "q":
                    flags["adapt_q"] = True
                elif ch == "k":
                    flags["adapt_k"] = True
                elif ch == "v":
                    flags["adapt_v"] = True
                elif ch == "g":
                    flags["adapt_guide"] = True
                else:
                    break
        force = False
        if "guide" in phase:
            force = phase.split("guide")[-1].split("_")[0]  # :277
        spec["local"] = dict(kt=kt, ks=ks, force_use_guide=force, **flags)
    if "global" in projector_type:
        phase = projector_type.split("global")[-1].split("local")[0]
        digits = ""
        for ch in phase:
            if ch.isdigit():
                digits += ch
            else:
                break
        force = False
        if "guide" in phase:
            force = phase.split("guide")[-1].split("_")[0]  # :297
        spec["global"] = dict(num_queries=int(digits), adapt_guide="adaptg" in phase,
                              force_use_guide=force)
    return spec


def tower_dims(mm_vision_tower: str) -> Tuple[int, int]:
    """(qk_dim, hw) per projector.py:407-414 / 569-576."""
    if "siglip-so400m-patch14-384" in mm_vision_tower:
        return 1152, 27
    if "clip-vit-large-patch14-336" in mm_vision_tower:
        return 768, 24
    raise NotImplementedError


def resolve_guide_mode(config, force_use_guide):
    """projector.py:422 / 585."""
    return getattr(config, "use_guide", None) if force_use_guide is False else force_use_guide


# --------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------
def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    w = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (projector.py:310)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def mlp2(x: Tensor, sd, prefix: str) -> Tensor:
    """build_mlp(depth=2): Linear -> GELU -> Linear (projector.py:307-312)."""
    return linear(gelu_erf(linear(x, sd, prefix + ".0")), sd, prefix + ".2")


def layer_norm(x: Tensor, sd, prefix: str) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * sd[prefix + ".weight"] + sd[prefix + ".bias"]


def mha(query: Tensor, key: Tensor, value: Tensor, sd, prefix: str, num_heads: int,
        logit_scale: Optional[Tensor] = None, logit_bias: Optional[Tensor] = None,
        return_scores: bool = False):
    """MultiheadAttention.forward on un-batched [Lq,E] / [Lk,E] inputs (projector.py:166-228)."""
    E = query.shape[-1]
    hd = E // num_heads
    q = linear(query, sd, prefix + ".q_proj")                     # :180
    k = linear(key, sd, prefix + ".k_proj")                       # :181
    v = linear(value, sd, prefix + ".v_proj")                     # :182
    if logit_scale is not None:                                    # :184-188
        q = q / q.norm(p=2, dim=-1, keepdim=True)
        k = k / k.norm(p=2, dim=-1, keepdim=True)
        scale, bias = logit_scale.exp(), logit_bias
    else:
        scale, bias = hd ** -0.5, 0.0                              # :145,190-191
    Lq, Lk = q.shape[0], k.shape[0]
    qh = q.reshape(Lq, num_heads, hd).permute(1, 0, 2)            # [nh,Lq,hd]
    kh = k.reshape(Lk, num_heads, hd).permute(1, 0, 2)
    vh = v.reshape(Lk, num_heads, hd).permute(1, 0, 2)
    scores = torch.matmul(qh, kh.transpose(1, 2)) * scale + bias   # :197
    p = torch.softmax(scores.float(), dim=-1).to(q.dtype)          # :213 (fp32 softmax)
    o = torch.matmul(p, vh).permute(1, 0, 2).reshape(Lq, E)        # :215,223-224
    out = linear(o, sd, prefix + ".out_proj")                      # :226
    if return_scores:
        return out, scores
    return out


# --------------------------------------------------------------------------
# GuideInjector                                   (projector.py:315-397)
# --------------------------------------------------------------------------
def _adapt_guide(g: Tensor, sd, prefix: str, adapt_guide: bool) -> Tensor:
    """text2qk_proj is Identity (text_dim == qk_dim, :323-326); guide blend :365/:389."""
    if not adapt_guide:
        return g
    a = sd[prefix + ".guide_alpha"]
    return (1 - a) * g + a * layer_norm(mlp2(g, sd, prefix + ".guide_proj"), sd, prefix + ".guide_norm")


def guide_inject(mode, visual: Tensor, guide: Optional[Tensor], sd, prefix: str,
                 adapt_guide: bool) -> Tensor:
    """visual: [..., D] (4-D [t,h,w,D] or 2-D [n,D]); guide: [D] (direct/coarse) or [L,D] (fine)."""
    if mode in (None, "off"):
        return visual                                              # IdentityMap :104-110
    if visual.ndim not in (4, 2):
        raise ValueError("Invalid input shape for guide embedding.")
    if mode in ("direct", "coarse"):
        if guide.ndim != 1:
            # einops 'd -> 1 1 1 d' on a non-1-D guide raises in the reference (:355/:359)
            raise ValueError("guide_embed must be 1-D for direct/coarse injection")
        g = _adapt_guide(guide, sd, prefix, adapt_guide)           # same for every position
        if mode == "direct":
            return g.expand(*visual.shape[:-1], g.shape[-1]).clone()   # :352-368
        cs = mlp2(g, sd, prefix + ".coarse_proj")                  # :370
        scale, shift = torch.chunk(cs, 2, dim=-1)                  # :371
        return layer_norm(visual * (1 + scale) + shift, sd, prefix + ".coarse_norm")  # :372
    if mode == "fine":
        if guide.ndim != 2:
            raise ValueError("guide_embed must be [L, D] for fine injection")
        g = _adapt_guide(guide, sd, prefix, adapt_guide)           # :388-389
        flat = visual.reshape(-1, visual.shape[-1])
        nh = visual.shape[-1] // 128                               # :341
        # 4-D: every position is its own batch entry with q_len 1 (:377-379); 2-D: one batch
        # of n queries (:382-383).  Both reduce to row-wise attention over the L guide tokens.
        att = mha(flat, g, g, sd, prefix + ".fine_proj", nh)       # :391
        out = layer_norm(flat + att, sd, prefix + ".fine_norm")    # :392
        return out.reshape(visual.shape)
    raise NotImplementedError                                      # :350


# --------------------------------------------------------------------------
# window geometry                                 (projector.py:473-522)
# --------------------------------------------------------------------------
def window_starts(n: int, k: int) -> Tuple[List[int], int]:
    """(start index of each group, group length) along an axis of n elements.

    n % k == 0: plain tiling (:476-477).  Otherwise `balance_divide_feature` (:501-522):
    ceil(n/k) groups, the first n % split (or all) are full, each remaining group starts
    one element early (overlap).  n < k gives ONE group holding all n elements (the slice
    x[0:k] is simply short, and a one-element torch.stack cannot fail).  When the groups come out
    with unequal lengths the reference's torch.stack raises RuntimeError (e.g.
    n=5,6,9 with k=4) -- reproduced here.
    """
    if n % k == 0:
        return [i * k for i in range(n // k)], k
    split = math.ceil(n / k)
    no_rep = n % split
    if no_rep == 0:
        no_rep = split
    lens = [k - (0 if i < no_rep else 1) for i in range(split)]
    starts, sizes, s = [], [], 0
    for i in range(split):
        e = s + lens[i]
        if lens[i] < k:
            s -= 1
        starts.append(s)
        sizes.append(len(range(n)[s:e]))          # python slice semantics, like x[s:e]
        s = e
    if any(sz != sizes[0] for sz in sizes):
        raise RuntimeError("stack expects each tensor to be equal size")
    return starts, sizes[0]


def window_token_index(T: int, H: int, W: int, kt: int, ks: int) -> Tensor:
    """Flat token ids [Nw, win]; window order (t1,h1,w1), in-window order (t2,h2,w2)
    row-major (divide_feature's final rearrange, :493)."""
    (ts, kt), (hs, kh), (ws, kw) = window_starts(T, kt), window_starts(H, ks), window_starts(W, ks)
    t = torch.tensor(ts).view(-1, 1, 1, 1, 1, 1) + torch.arange(kt).view(1, 1, 1, -1, 1, 1)
    y = torch.tensor(hs).view(1, -1, 1, 1, 1, 1) + torch.arange(kh).view(1, 1, 1, 1, -1, 1)
    x = torch.tensor(ws).view(1, 1, -1, 1, 1, 1) + torch.arange(kw).view(1, 1, 1, 1, 1, -1)
    idx = (t * H + y) * W + x
    return idx.reshape(len(ts) * len(hs) * len(ws), kt * kh * kw)


def _lerp_taps(n_in: int, n_out: int):
    """1-D taps of F.interpolate(mode='trilinear', align_corners=False) given `size`:
    src = (i + 0.5) * n_in / n_out - 0.5, clamped at 0; i1 = min(i0 + 1, n_in - 1)."""
    scale = n_in / n_out
    i0s, i1s, lams = [], [], []
    for i in range(n_out):
        src = max((i + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        i0s.append(i0), i1s.append(i1), lams.append(src - i0)
    return i0s, i1s, lams


def pooled_query(ff: Tensor, out_size: Tuple[int, int, int]) -> Tensor:
    """F.interpolate(ff as [1,D,T,H,W], size=out_size, 'trilinear') -> [t',h',w',D] (:539-540),
    restated as three separable 2-tap lerps."""
    x = ff
    for axis, n_out in enumerate(out_size):
        n_in = x.shape[axis]
        i0, i1, lam = _lerp_taps(n_in, n_out)
        lam_t = torch.tensor(lam, dtype=x.dtype).view([-1 if a == axis else 1 for a in range(x.ndim)])
        a = x.index_select(axis, torch.tensor(i0))
        b = x.index_select(axis, torch.tensor(i1))
        x = a * (1 - lam_t) + b * lam_t
    return x


# --------------------------------------------------------------------------
# LocalCompressor.forward                          (projector.py:524-559)
# --------------------------------------------------------------------------
def local_context(spec: dict, mode, sd, prefix: str, ff: Tensor, fe: Optional[Tensor],
                  guide: Optional[Tensor], modal: str,
                  logit_scale: Optional[Tensor], logit_bias: Optional[Tensor]):
    """Everything up to (not including) the readout: returns (ctx [t',h',w',D], attn [Nw,win])."""
    T, H, W, D = ff.shape
    if fe is not None and logit_scale is not None:                 # :527-529
        fe = fe / fe.norm(p=2, dim=-1, keepdim=True)
        guide = guide / guide.norm(p=2, dim=-1, keepdim=True)
    fe = ff if fe is None else fe                                  # :532
    if spec["adapt_k"]:                                            # :533
        a = sd[prefix + ".k_alpha"]
        key = (1 - a) * fe + a * layer_norm(mlp2(fe, sd, prefix + ".k_proj"), sd, prefix + ".k_norm")
    else:
        key = fe
    if spec["adapt_v"]:                                            # :534
        a = sd[prefix + ".v_alpha"]
        value = (1 - a) * ff + a * layer_norm(mlp2(ff, sd, prefix + ".v_proj"), sd, prefix + ".v_norm")
    else:
        value = ff
    kt = 1 if (modal == "image" or T == 1) else spec["kt"]         # :536
    ks = spec["ks"]
    out_size = (math.ceil(T / kt), math.ceil(H / ks), math.ceil(W / ks))   # :537
    q = pooled_query(ff, out_size)                                 # :539-540
    adapt_q = spec["adapt_q"] and mode != "direct"                 # :428-429
    if adapt_q:                                                    # :541
        a = sd[prefix + ".q_alpha"]
        qp = q @ sd[prefix + ".q_proj.weight"].t()                 # Linear(bias=False) :433
        q = (1 - a) * q + a * layer_norm(qp, sd, prefix + ".q_norm")
    query = guide_inject(mode, q, guide, sd, prefix + ".guide_injector", spec["adapt_guide"])  # :542
    idx = window_token_index(T, H, W, kt, ks)                      # :544-545
    if idx.shape[0] != out_size[0] * out_size[1] * out_size[2]:
        raise RuntimeError("window count does not match the pooled-query grid")
    kw = key.reshape(-1, D)[idx]                                   # [Nw,win,D]
    vw = value.reshape(-1, D)[idx]
    qw = query.reshape(-1, D)                                      # divide_feature(query,(1,1,1)) :546
    s = torch.einsum("nd,nkd->nk", qw, kw)
    if logit_scale is not None:
        s = s * logit_scale.exp() + logit_bias                     # :549
    else:
        s = s / math.sqrt(D)                                       # :551  (qk_dim, not head dim)
    attn = torch.softmax(s, dim=-1)                                # input-dtype softmax (:549/:551)
    ctx = torch.einsum("nk,nkd->nd", attn, vw)                     # :553
    return ctx.reshape(*out_size, D), attn                         # :554-558


def local_forward(spec, mode, sd, prefix, ff, fe, guide, modal, logit_scale=None, logit_bias=None):
    ctx, _ = local_context(spec, mode, sd, prefix, ff, fe, guide, modal, logit_scale, logit_bias)
    return mlp2(ctx, sd, prefix + ".readout")                      # :559


# --------------------------------------------------------------------------
# 3-D sinusoidal position table                    (projector.py:57-101)
# --------------------------------------------------------------------------
def pos_axis_table(n: int, d_model: int) -> np.ndarray:
    """One axis of get_3d_position_embedding in float64: PE(p)[c] = sin(p / 10000^(2*(c//2)/D))
    for even c, cos for odd c.  `np.float32(d_model)` in the exponent still promotes to
    float64."""
    pos = np.arange(n)[:, None]
    i = np.arange(d_model)[None, :]
    ang = pos / np.power(10000, (2 * (i // 2)) / np.float32(d_model))
    pe = np.zeros_like(ang)
    pe[:, 0::2] = np.sin(ang[:, 0::2])
    pe[:, 1::2] = np.cos(ang[:, 1::2])
    return pe


_POS_CACHE: Dict[Tuple[int, int, int, int], Tensor] = {}


def pos_table(T: int, H: int, W: int, d_model: int) -> Tensor:
    """float32 [T,H,W,D] = (PE_t + PE_h + PE_w in float64).float()  (:95-99, :606).
    Cached like the reference's `pos_embed` buffer, which is built once at construction (:601)."""
    key = (T, H, W, d_model)
    if key not in _POS_CACHE:
        if len(_POS_CACHE) > 4:
            _POS_CACHE.clear()
        _POS_CACHE[key] = _pos_table_uncached(T, H, W, d_model)
    return _POS_CACHE[key]


def _pos_table_uncached(T: int, H: int, W: int, d_model: int) -> Tensor:
    pt, ph, pw = pos_axis_table(T, d_model), pos_axis_table(H, d_model), pos_axis_table(W, d_model)
    full = pt[:, None, None, :] + ph[None, :, None, :] + pw[None, None, :, :]
    return torch.from_numpy(full).float()


# --------------------------------------------------------------------------
# GlobalCompressor.forward                         (projector.py:634-646)
# --------------------------------------------------------------------------
def global_forward(spec: dict, mode, sd, prefix: str, ff: Tensor, guide: Optional[Tensor],
                   logit_scale=None, logit_bias=None, use_pos_emb: bool = True,
                   return_parts: bool = False):
    T, H, W, D = ff.shape
    x = ff
    if use_pos_emb:
        x = x + pos_table(T, H, W, D).to(ff.dtype)                 # :636-640 (fp32 table cast to input dtype)
    query = guide_inject(mode, sd[prefix + ".query"], guide, sd,
                         prefix + ".guide_injector", spec["adapt_guide"])   # :642
    kv = x.reshape(-1, D)                                          # :644
    nh = D // 128                                                  # :579
    att, scores = mha(query, kv, kv, sd, prefix + ".attn_layer", nh, logit_scale, logit_bias,
                      return_scores=True)                          # :645
    pre = query + att                                              # residual with the INJECTED query (:646)
    out = mlp2(pre, sd, prefix + ".readout")
    if return_parts:
        return out, dict(query=query, scores=scores, attn_out=att, pre_readout=pre)
    return out


def global_forward_chunked(spec: dict, mode, sd, prefix: str, ff: Tensor, guide: Optional[Tensor],
                           chunk_frames: int = 32, use_pos_emb: bool = True):
    """global_forward for clips whose K / V states do not fit host memory at once (1024 frames = 746k keys): the same
    un-folded reference formulation -- k = k_proj(x + pos), v = v_proj(x + pos) per token (:180-182, :636-640) -- evaluated
    `chunk_frames` frames at a time, the softmax over ALL keys (:213) carried across chunks as a running (max, sum,
    weighted sum) triple.  Mathematically identical to global_forward (checked against it in tests/test_oracle_golden.py);
    no clip-scale."""
    T, H, W, D = ff.shape
    query = guide_inject(mode, sd[prefix + ".query"], guide, sd, prefix + ".guide_injector", spec["adapt_guide"])   # :642
    nh = D // 128
    hd = D // nh
    att_p = prefix + ".attn_layer"
    q = linear(query, sd, att_p + ".q_proj")                       # :180
    Lq = q.shape[0]
    qh = q.reshape(Lq, nh, hd).permute(1, 0, 2)                    # [nh, Lq, hd]
    m_run = torch.full((nh, Lq), -float("inf"), dtype=torch.float32)
    l_run = torch.zeros((nh, Lq), dtype=torch.float32)
    o_run = torch.zeros((nh, Lq, hd), dtype=torch.float32)
    pt, ph, pw = (pos_axis_table(n, D) for n in (T, H, W))
    for t0 in range(0, T, chunk_frames):
        t1 = min(T, t0 + chunk_frames)
        x = ff[t0:t1]
        if use_pos_emb:
            pos = pt[t0:t1, None, None, :] + ph[None, :, None, :] + pw[None, None, :, :]        # float64 sum (:95-99)
            x = x + torch.from_numpy(pos).float().to(ff.dtype)                                # fp32 buffer cast to the input dtype
        kv = x.reshape(-1, D)
        k = linear(kv, sd, att_p + ".k_proj").reshape(-1, nh, hd).permute(1, 0, 2)             # :181
        v = linear(kv, sd, att_p + ".v_proj").reshape(-1, nh, hd).permute(1, 0, 2)             # :182
        s = torch.matmul(qh, k.transpose(1, 2)).float() * hd ** -0.5                          # :197
        m_new = torch.maximum(m_run, s.max(dim=-1).values)
        alpha = torch.exp(m_run - m_new)
        p = torch.exp(s - m_new[..., None])
        l_run = l_run * alpha + p.sum(dim=-1)
        o_run = o_run * alpha[..., None] + torch.matmul(p, v.float())
        m_run = m_new
    o = (o_run / l_run[..., None]).permute(1, 0, 2).reshape(Lq, D).to(q.dtype)                  # :215, :223-224
    att = linear(o, sd, att_p + ".out_proj")                       # :226
    return mlp2(query + att, sd, prefix + ".readout")              # :646


# --------------------------------------------------------------------------
# post_process_visual_feature                      (mm_utils.py:92-140)
# --------------------------------------------------------------------------
def post_process(config, feat: Tensor, modal: str, image_newline: Optional[Tensor],
                 is_anyres: bool) -> Tensor:
    merge = getattr(config, "mm_patch_merge_type", "flat")          # :93
    nlpos = getattr(config, "mm_newline_position", "one_token")     # :94
    t, h, w, d = feat.shape
    flat = feat.reshape(t * h * w, d)
    if merge == "flat" or not merge.startswith("spatial"):          # :96-97, :137-138
        return flat
    if modal == "video":
        if nlpos == "grid":                                         # :101-107  newline after every grid row
            nl = image_newline.to(feat.dtype).expand(t, h, 1, d)
            return torch.cat([feat, nl], dim=2).reshape(t * h * (w + 1), d)
        if nlpos == "frame":                                        # :108-114  newline after every frame
            nl = image_newline.to(feat.dtype).expand(t, 1, d)
            return torch.cat([feat.reshape(t, h * w, d), nl], dim=1).reshape(t * (h * w + 1), d)
        if nlpos == "one_token":                                    # :115-117
            return torch.cat([flat, image_newline[None].to(feat.dtype)], dim=0)
        if nlpos == "no_token":                                     # :118-119
            return flat
        raise ValueError(f"Unexpected mm_newline_position: {nlpos}")
    if modal == "image":
        if t != 1:
            raise ValueError("image modality expects a single [1,h,w,d] grid")   # einops '1 h w d' (:125,132,135)
        if is_anyres:                                               # :124-130
            nl = image_newline.to(feat.dtype).expand(h, 1, d)
            return torch.cat([feat[0], nl], dim=1).reshape(h * (w + 1), d)
        if image_newline is not None:                               # :131-133
            return torch.cat([flat, image_newline[None].to(feat.dtype)], dim=0)
        return flat                                                 # :134-135
    return feat  # spatial + unknown modal: the reference returns the tensor untouched


# --------------------------------------------------------------------------
# HIComProjector.forward                           (projector.py:676-708)
# --------------------------------------------------------------------------
def projector_forward(config, sd: Dict[str, Tensor], ff, fe, guide, modal: str,
                      image_newline: Optional[Tensor] = None,
                      logit: Optional[dict] = None) -> Tensor:
    """`logit` optionally carries {"local": (scale, bias), "global": (scale, bias)} for the
    clip-scale variant (:655-670); None entries mean the plain 1/sqrt(d) scaling."""
    spec = parse_projector_type(getattr(config, "mm_projector_type", "linear"))
    assert spec["local"] is not None or spec["global"] is not None, \
        "At least one compressor should be provided."              # :674
    logit = logit or {}
    l_scale, l_bias = logit.get("local", (None, None))
    g_scale, g_bias = logit.get("global", (None, None))
    local_x = global_x = None
    if spec["local"] is not None:
        lmode = resolve_guide_mode(config, spec["local"]["force_use_guide"])
        run = lambda f, e: local_forward(spec["local"], lmode, sd, "local_compressor", f, e, guide,
                                         modal, l_scale, l_bias)
        if isinstance(ff, dict):                                    # :679-689
            parts = []
            if ff["base"] is not None:
                b = run(ff["base"].unsqueeze(0), fe["base"].unsqueeze(0) if fe is not None else None)
                parts.append(post_process(config, b, modal, image_newline, False))
            p = run(ff["patch"].unsqueeze(0), fe["patch"].unsqueeze(0) if fe is not None else None)
            parts.append(post_process(config, p, modal, image_newline, True))
            local_x = torch.cat(parts, dim=-2)
        else:                                                       # :691-692
            local_x = post_process(config, run(ff, fe), modal, image_newline, False)
    if spec["global"] is not None:
        gmode = resolve_guide_mode(config, spec["global"]["force_use_guide"])
        gff = ff["patch"].unsqueeze(0) if isinstance(ff, dict) else ff   # :695-700
        global_x = global_forward(spec["global"], gmode, sd, "global_compressor", gff, guide,
                                  g_scale, g_bias)
    if local_x is None:
        return global_x
    if global_x is None:
        return local_x
    return torch.cat([local_x, global_x], dim=-2)                   # :707


# --------------------------------------------------------------------------
# parameter schema (names/shapes of the reference state dict, SURVEY §8a "Parameters")
# --------------------------------------------------------------------------
def param_shapes(config) -> Dict[str, Tuple[int, ...]]:
    spec = parse_projector_type(config.mm_projector_type)
    D, _ = tower_dims(config.mm_vision_tower)
    E, Hd = config.mm_hidden_size, config.hidden_size
    shapes: Dict[str, Tuple[int, ...]] = {}

    def lin(name, n_in, n_out, bias=True):
        shapes[name + ".weight"] = (n_out, n_in)
        if bias:
            shapes[name + ".bias"] = (n_out,)

    def mlp(name, n_in, n_out):
        lin(name + ".0", n_in, n_out)
        lin(name + ".2", n_out, n_out)

    def ln(name, n):
        shapes[name + ".weight"] = (n,)
        shapes[name + ".bias"] = (n,)

    def attn(name, n):
        for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(f"{name}.{p}", n, n)

    def injector(name, mode, adapt_guide, qk):
        if mode in (None, "off"):
            return
        if adapt_guide:
            mlp(name + ".guide_proj", qk, qk)
            ln(name + ".guide_norm", qk)
            shapes[name + ".guide_alpha"] = (1,)
        if mode == "coarse":
            mlp(name + ".coarse_proj", qk, 2 * qk)
            ln(name + ".coarse_norm", qk)
        elif mode == "fine":
            attn(name + ".fine_proj", qk)
            ln(name + ".fine_norm", qk)

    if spec["local"] is not None:
        s = spec["local"]
        mode = resolve_guide_mode(config, s["force_use_guide"])
        p = "local_compressor"
        injector(p + ".guide_injector", mode, s["adapt_guide"], D)
        if s["adapt_q"] and mode != "direct":
            lin(p + ".q_proj", D, D, bias=False)
            ln(p + ".q_norm", D)
            shapes[p + ".q_alpha"] = (1,)
        if s["adapt_k"]:
            mlp(p + ".k_proj", D, D)
            ln(p + ".k_norm", D)
            shapes[p + ".k_alpha"] = (1,)
        if s["adapt_v"]:
            mlp(p + ".v_proj", E, E)
            ln(p + ".v_norm", E)
            shapes[p + ".v_alpha"] = (1,)
        mlp(p + ".readout", E, Hd)
    if spec["global"] is not None:
        s = spec["global"]
        mode = resolve_guide_mode(config, s["force_use_guide"])
        p = "global_compressor"
        shapes[p + ".query"] = (s["num_queries"], E)
        injector(p + ".guide_injector", mode, s["adapt_guide"], E)
        attn(p + ".attn_layer", E)
        mlp(p + ".readout", E, Hd)
    return shapes


def make_config(**kw) -> SimpleNamespace:
    base = dict(mm_projector_type="local43_global32_coarse", use_guide="direct", use_clip_scale="",
                mm_patch_merge_type="spatial_unpad", mm_newline_position="no_token",
                mm_vision_tower="google/siglip-so400m-patch14-384", mm_hidden_size=1152,
                hidden_size=896, max_num_frames=32)
    base.update(kw)
    return SimpleNamespace(**base)


# ---------------------------------------------------------------------------------------------
# Upstream neighbour of the path: the per-patch SigLIP pooling-head projection that
# produces frames_embed,
#     image_embeds = head.layernorm(last_hidden_state); image_embeds = last_hidden_state + head.mlp(image_embeds)
# (reference hicom/model/encoder.py:284-286).  `head` is HF transformers' SiglipMultiheadAttentionPoolingHead (a
# third-party dependency of the reference, README.md:21 pins transformers 4.45-class; here 5.15): layernorm = nn.LayerNorm
# (eps = config.layer_norm_eps = 1e-6), mlp = SiglipMLP: fc1 -> ACT2FN[config.hidden_act] -> fc2 with
# hidden_act = "gelu_pytorch_tanh" and intermediate_size 4304 for siglip-so400m-patch14-384 (HF config values, not in
# the reference).  Pinned by tests/golden/make_golden_head.py, which evaluates those two reference lines on the HF modules.
def gelu_tanh(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def siglip_head_embed(x: Tensor, sd: Dict[str, Tensor], prefix: str = "head", eps: float = 1e-6, act: str = "gelu_pytorch_tanh") -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    h = (x - mu) / torch.sqrt(var + eps) * sd[prefix + ".layernorm.weight"] + sd[prefix + ".layernorm.bias"]
    h = linear(h, sd, prefix + ".mlp.fc1")
    h = gelu_tanh(h) if act == "gelu_pytorch_tanh" else gelu_erf(h)
    return x + linear(h, sd, prefix + ".mlp.fc2")
