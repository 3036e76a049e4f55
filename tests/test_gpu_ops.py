"""Operator-level parity (GPU): every C-ABI entry point against the oracle's math on small seeded
inputs.  fp tolerances are written next to each check; integer/index behaviour is exact."""
import math

import numpy as np
import pytest
import torch

import cases
from hicom_amd import geometry as geo
from hicom_amd import native as nv
from hicom_amd import synth
from oracle import hicom_oracle as orc

pytestmark = pytest.mark.gpu

D = 1152


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).cuda()


def f32(shape):
    return torch.empty(shape, dtype=torch.float32, device="cuda")


def maxabs(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


@pytest.mark.parametrize("T,H,W,kt,ks", [(8, 6, 6, 4, 3), (7, 7, 5, 4, 3), (1, 6, 6, 1, 3), (4, 27, 27, 4, 3), (4, 7, 7, 4, 2), (3, 2, 6, 4, 3)])
@pytest.mark.parametrize("shared_query", [True, False])
def test_local_attn_matches_oracle(T, H, W, kt, ks, shared_query):
    x = synth.synth_inputs(T, H, W, D, tag=f"loc{T}{H}{W}")
    ff, fe, g = x["ff"], x["fe"], x["g"]
    spec = dict(kt=kt, ks=ks, adapt_q=False, adapt_k=False, adapt_v=False, adapt_guide=False)
    mode = "direct" if shared_query else None
    tff, tfe, tg = (torch.from_numpy(a) for a in (ff, fe, g))
    want, _ = orc.local_context(spec, mode, {}, "lc", tff, tfe, tg, "video", None, None)
    at, ay, ax = geo.axis_tiling(T, 1 if T == 1 else kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
    nw = at.nwin * ay.nwin * ax.nwin
    ctx = f32((nw, D))
    dff, dfe = bf(ff), bf(fe)
    if shared_query:
        nv.local_attn(dfe, dff, axes, bf(g), 0, 1 / math.sqrt(D), 0.0, 0, ctx)
    else:
        q = f32((at.nwin, ay.nwin, ax.nwin, D))
        nv.trilinear_pool(dff, q)
        want_q = orc.pooled_query(tff, (at.nwin, ay.nwin, ax.nwin))
        assert maxabs(q, want_q) <= 5e-6          # fp32 lerp weights vs the oracle's double taps
        nv.local_attn(dfe, dff, axes, q, D, 1 / math.sqrt(D), 0.0, 0, ctx)
    torch.cuda.synchronize()
    assert maxabs(ctx, want.reshape(nw, D)) <= 2e-5          # fp32 accumulation-order noise only


def test_local_attn_clip_scale_and_uniform():
    T, H, W = 4, 6, 6
    x = synth.synth_inputs(T, H, W, D, tag="clip")
    spec = dict(kt=4, ks=3, adapt_q=False, adapt_k=False, adapt_v=False, adapt_guide=False)
    ls, lb = torch.tensor(2.0), torch.tensor(-3.0)
    tff, tfe, tg = (torch.from_numpy(x[k]) for k in ("ff", "fe", "g"))
    want, _ = orc.local_context(spec, "direct", {}, "lc", tff, tfe, tg, "video", ls, lb)
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (geo.axis_tiling(T, 4), geo.axis_tiling(H, 3), geo.axis_tiling(W, 3)))
    ctx = f32((4, D))
    nv.local_attn(bf(x["fe"]), bf(x["ff"]), axes, bf(x["g"]), 0, math.exp(2.0), -3.0, 3, ctx)
    assert maxabs(ctx, want.reshape(4, D)) <= 2e-5
    # known answer: zero query -> uniform attention -> plain window mean of the value stream
    nv.local_attn(bf(x["fe"]), bf(x["ff"]), axes, bf(np.zeros(D, np.float32)), 0, 1.0, 0.0, 0, ctx)
    idx = orc.window_token_index(T, H, W, 4, 3)
    mean = tff.reshape(-1, D)[idx].mean(dim=1)
    assert maxabs(ctx, mean) <= 2e-6


@pytest.mark.parametrize("M,N,K,xdt,wdt", [(1, 1152, 1152, "bf16", "bf16"), (32, 1152, 1152, "f32", "bf16"),
                                           (9, 118, 1152, "f32", "f32"), (3, 64, 1152, "f32", "bf16"), (32, 896, 896, "f32", "bf16"),
                                           (64, 1152, 1152, "bf16", "bf16"), (17, 1152, 1152, "f32", "bf16"), (50, 48, 96, "f32", "bf16"),
                                           (2, 2304, 1152, "bf16", "bf16"), (65, 1152, 1152, "f32", "bf16")])
def test_linear_matches_torch(M, N, K, xdt, wdt):
    x = synth.normal_like((M, K), 11)
    w = synth.normal_like((N, K), 12, 0.05)
    b = synth.normal_like((N,), 13, 0.1)
    res = synth.normal_like((M, N), 14)
    want = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    want_g = (0.5 * want * (1 + torch.erf(want / math.sqrt(2)))) + torch.from_numpy(res).double()
    dx = bf(x) if xdt == "bf16" else torch.from_numpy(x).cuda()
    dw = bf(w) if wdt == "bf16" else torch.from_numpy(w).cuda()
    y = f32((M, N))
    nv.linear(dx, dw, bf(b), y)
    assert maxabs(y, want) <= 2e-5 * max(1.0, float(want.abs().max()))
    nv.linear(dx, dw, bf(b), y, res=torch.from_numpy(res).cuda(), act=nv.ACT_GELU)
    assert maxabs(y, want_g) <= 2e-5 * max(1.0, float(want_g.abs().max()))


def test_fold_split_headproj():
    nq, nh, E = 5, 9, 1152
    hd = E // nh
    qp = synth.normal_like((nq, E), 21)
    wk = synth.normal_like((E, E), 22, 0.02)
    qt = f32((nq * nh, E))
    nv.fold_query(torch.from_numpy(qp).cuda(), bf(wk), nh, hd ** -0.5, qt)
    tq, tw = torch.from_numpy(qp).double(), torch.from_numpy(wk).double()
    want = torch.stack([tq[q, h * hd:(h + 1) * hd] @ tw[h * hd:(h + 1) * hd, :] for q in range(nq) for h in range(nh)]) * hd ** -0.5
    assert maxabs(qt, want) <= 1e-6
    hi = torch.empty((48, E), dtype=torch.bfloat16, device="cuda")
    lo = torch.empty_like(hi)
    nv.split_bf16(qt, 48, hi, lo)
    rec = hi.float() + lo.float()
    assert float((rec[:45] - qt).abs().max()) <= 2 ** -16 * float(qt.abs().max())
    assert float(rec[45:].abs().max()) == 0.0
    assert torch.equal(hi[:45], qt.to(torch.bfloat16))              # hi is the RNE bf16 of the value
    # per-head v_proj: o[q, f] = w_v[f] . ctx[q*nh + f // hd] + b_v[f]
    ctx = synth.normal_like((nq * nh, E), 23)
    bv = synth.normal_like((E,), 24, 0.1)
    o = f32((nq, E))
    nv.linear(torch.from_numpy(ctx).cuda(), bf(wk), bf(bv), o, head_rows=nh, head_dim=hd)
    tc = torch.from_numpy(ctx).double().reshape(nq, nh, E)
    want_o = torch.stack([torch.cat([tw[h * hd:(h + 1) * hd] @ tc[q, h] for h in range(nh)]) for q in range(nq)]) + torch.from_numpy(bv).double()
    assert maxabs(o, want_o) <= 1e-5


@pytest.mark.parametrize("M,N,K,act", [(8, 64, 1152, 1), (81, 896, 1152, 1), (130, 64, 64, 0), (1296, 896, 896, 0)])
def test_readout_gemm_matches_torch(M, N, K, act):
    x = synth.normal_like((M, K), 31)
    x = (x + synth.normal_like((M, K), 35) * 2.0 ** -10).astype(np.float32)   # NOT bf16-representable
    w = synth.normal_like((N, K), 32, 0.03)
    b = synth.normal_like((N,), 33, 0.1)
    want = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    if act:
        want = 0.5 * want * (1 + torch.erf(want / math.sqrt(2)))
    y = f32((M, N))
    nv.readout_gemm(torch.from_numpy(x).cuda(), bf(w), bf(b), y, act=act)
    assert maxabs(y, want) <= 5e-5 * max(1.0, float(want.abs().max()))        # hi+lo split: ~2^-16 relative
    # packed store: newline gap after every 9 rows, offset 3, bf16 output
    rows = 3 + M + M // 9 + 1
    yp = torch.zeros((rows, N), dtype=torch.bfloat16, device="cuda")
    nv.readout_gemm(torch.from_numpy(x).cuda(), bf(w), bf(b), yp, act=act, row0=3, nl_group=9)
    got = yp.float().cpu()
    for m in sorted({0, min(8, M - 1), min(9, M - 1), M - 1}):
        r = 3 + m + m // 9
        assert torch.equal(got[r], want[m].float().to(torch.bfloat16).float()) or maxabs(got[r], want[m]) <= 2 ** -8 * float(want.abs().max())
    assert float(got[:3].abs().max()) == 0.0 and (M < 9 or float(got[3 + 9].abs().max()) == 0.0)   # gap rows untouched


def test_scatter_rows():
    src = synth.normal_like((6, 64), 41)
    dst = torch.zeros((20, 64), dtype=torch.float32, device="cuda")
    nv.scatter_rows(bf(src), dst, 1, 6, nl_group=2)             # rows 1,2,_,4,5,_,7,8
    got = dst.cpu().numpy()
    for i, r in enumerate((1, 2, 4, 5, 7, 8)):
        assert np.array_equal(got[r], src[i])
    assert not got[[0, 3, 6, 9]].any()
    nv.scatter_rows(bf(src[:1]), dst, 10, 5, row_step=2)        # broadcast one row to 10,12,..,18
    got = dst.cpu().numpy()
    for r in (10, 12, 14, 16, 18):
        assert np.array_equal(got[r], src[0])
    assert not got[[11, 13]].any()


def _global_reference(T, H, W, nq, seed_tag, peaky=1.0, in_scale=1.0):
    """fp64 oracle pieces of the global attention for nq distinct queries."""
    E, nh = D, 9
    x = synth.synth_inputs(T, H, W, D, tag=seed_tag, scale=in_scale)
    q = synth.normal_like((nq, E), synth.seed_of(seed_tag + ":q"))
    shapes = {f"a.{p}.{k}": s for p in ("q_proj", "k_proj", "v_proj", "out_proj") for k, s in (("weight", (E, E)), ("bias", (E,)))}
    sd = synth.synth_state_dict(shapes, tag=seed_tag, peaky=peaky)
    tsd = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    xp = torch.from_numpy(x["ff"]).double().reshape(-1, E) + orc.pos_table(T, H, W, E).double().reshape(-1, E)
    # note: the oracle proper adds the float32 pos table; float64 here isolates kernel error
    _, scores = orc.mha(torch.from_numpy(q).double(), xp, xp, tsd, "a", nh, return_scores=True)   # [nh,nq,N]
    p = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("hqn,ne->qhe", p, xp).reshape(nq * nh, E)
    return x, q, sd, scores.permute(1, 0, 2).reshape(nq * nh, -1), ctx


@pytest.mark.parametrize("T,H,W,nq,peaky,in_scale", [(8, 6, 6, 1, 1.0, 1.0), (4, 7, 5, 3, 1.0, 1.0), (8, 6, 6, 2, 12.0, 2.0), (2, 27, 27, 1, 1.0, 1.0)])
def test_global_stream_merge_combine(T, H, W, nq, peaky, in_scale):
    E, nh = D, 9
    x, q, sd, want_scores, want_ctx = _global_reference(T, H, W, nq, f"glob{T}{H}{W}{nq}", peaky, in_scale)
    N, R = T * H * W, nq * nh
    rows_pad = (R + 15) // 16 * 16
    dq = bf(q)
    qp, qt = f32((nq, E)), f32((R, E))
    nv.linear(dq, bf(sd["a.q_proj.weight"]), bf(sd["a.q_proj.bias"]), qp)
    nv.fold_query(qp, bf(sd["a.k_proj.weight"]), nh, (E // nh) ** -0.5, qt)
    qhi = torch.empty((rows_pad, E), dtype=torch.bfloat16, device="cuda")
    qlo = torch.empty_like(qhi)
    nv.split_bf16(qt, rows_pad, qhi, qlo)
    cap = T + 3
    pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
    pos_a = torch.zeros((rows_pad, pe.shape[0]), dtype=torch.float32, device="cuda")
    nv.linear(qt, pe, None, pos_a, M=R)
    ff = bf(x["ff"])
    for nparts in (1, 3, nv.global_stream_nparts(N, rows_pad)):
        stride = (N + 15) // 16 * 16
        scores = f32((rows_pad, stride))
        pm, pl, pacc = f32((nparts, rows_pad)), f32((nparts, rows_pad)), f32((nparts, rows_pad, E))
        nv.global_stream(ff, N, qhi, qlo, pos_a, H, W, 0, cap, cap + H, scores, pm, pl, pacc, rows=R)
        ml, acc = f32((R, 2)), f32((R, E))
        scratch = f32((R * T * (H + W + 2),))
        nv.global_merge(pm, pl, pacc, R, scores, N, H, W, pe, 0, cap, cap + H, scratch, ml, acc)
        ctx = f32((R, E))
        nv.global_combine(ml.unsqueeze(0), acc.unsqueeze(0), ctx)
        torch.cuda.synchronize()
        # logits agree up to the per-row constant q_h . b_k that softmax cancels
        diff = scores[:R, :N].double().cpu() - want_scores
        diff = diff - diff.mean(dim=1, keepdim=True)
        assert float(diff.abs().max()) <= 2e-4 * max(1.0, float(want_scores.abs().max())), nparts
        # hi+lo bf16 queries carry ~2^-17 relative logit error; with |logit| ~ 1e2 (peaky case) that is
        # ~1e-3 absolute in the exponent, i.e. ~3e-4 relative in the context
        rel = 3e-4 if peaky > 1 else 1e-4
        assert maxabs(ctx, want_ctx) <= rel * max(1.0, float(want_ctx.abs().max())), nparts


def test_global_frame_shard_merge_equals_unsharded():
    """Two frame shards with absolute frame offsets, combined, equal the single pass."""
    E, nh, T, H, W = D, 9, 8, 6, 6
    x, q, sd, _, want_ctx = _global_reference(T, H, W, 1, "shard")
    R, rows_pad = nh, 16
    qp, qt = f32((1, E)), f32((R, E))
    nv.linear(bf(q), bf(sd["a.q_proj.weight"]), bf(sd["a.q_proj.bias"]), qp)
    nv.fold_query(qp, bf(sd["a.k_proj.weight"]), nh, (E // nh) ** -0.5, qt)
    qhi = torch.empty((rows_pad, E), dtype=torch.bfloat16, device="cuda")
    qlo = torch.empty_like(qhi)
    nv.split_bf16(qt, rows_pad, qhi, qlo)
    cap = 16
    pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
    pos_a = torch.zeros((rows_pad, pe.shape[0]), dtype=torch.float32, device="cuda")
    nv.linear(qt, pe, None, pos_a, M=R)
    ff = bf(x["ff"])
    mls, accs = [], []
    for t0, t1 in ((0, 4), (4, 8)):
        shard = ff[t0:t1].contiguous()
        n = (t1 - t0) * H * W
        scores = f32((rows_pad, (n + 15) // 16 * 16))
        pm, pl, pacc = f32((2, rows_pad)), f32((2, rows_pad)), f32((2, rows_pad, E))
        nv.global_stream(shard, n, qhi, qlo, pos_a, H, W, t0, cap, cap + H, scores, pm, pl, pacc, rows=R)
        ml, acc = f32((R, 2)), f32((R, E))
        nv.global_merge(pm, pl, pacc, R, scores, n, H, W, pe, t0, cap, cap + H, f32((R * 4 * (H + W + 2),)), ml, acc)
        mls.append(ml), accs.append(acc)
    ctx = f32((R, E))
    nv.global_combine(torch.stack(mls), torch.stack(accs), ctx)
    assert maxabs(ctx, want_ctx) <= 1e-4 * max(1.0, float(want_ctx.abs().max()))


def test_fold_query_split_matches_unfused_chain():
    """hicom_fold_query_split_fwd (hot path) == fold -> split -> qt . PE^T (reference chain of ops)."""
    nq, nh, E, H, W, cap = 3, 9, 1152, 6, 5, 12
    hd = E // nh
    qp = torch.from_numpy(synth.normal_like((nq, E), 51)).cuda()
    wk = bf(synth.normal_like((E, E), 52, 0.02))
    pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
    P, R, rows_pad = pe.shape[0], nq * nh, 32
    qt = f32((R, E))
    nv.fold_query(qp, wk, nh, hd ** -0.5, qt)
    hi0 = torch.empty((rows_pad, E), dtype=torch.bfloat16, device="cuda")
    lo0 = torch.empty_like(hi0)
    nv.split_bf16(qt, rows_pad, hi0, lo0)
    pa0 = torch.zeros((rows_pad, P), dtype=torch.float32, device="cuda")
    nv.linear(qt, pe, None, pa0, M=R)
    kpe = f32((E, P))
    nv.linear(wk, pe, None, kpe)
    hi1 = torch.zeros_like(hi0)
    lo1 = torch.zeros_like(lo0)
    pa1 = torch.zeros_like(pa0)
    nv.fold_query_split(qp, wk, kpe, nh, hd ** -0.5, hi1, lo1, pa1)
    rec0, rec1 = hi0.float() + lo0.float(), hi1.float() + lo1.float()
    assert float((rec0 - rec1).abs().max()) <= 2e-6 * max(1.0, float(rec0.abs().max()))
    assert float((pa0 - pa1).abs().max()) <= 2e-5 * max(1.0, float(pa0.abs().max()))
    assert float(rec1[R:].abs().max()) == 0.0 and float(pa1[R:].abs().max()) == 0.0


def test_combine_strided_equals_dense():
    nsets, R, E = 3, 9, 1152
    stride = 2 * R + R * E + 40
    buf = torch.from_numpy(synth.normal_like((nsets, stride), 61)).cuda()
    buf[:, 1:2 * R:2] = buf[:, 1:2 * R:2].abs() + 0.5                 # L > 0
    ml = buf[:, :2 * R].contiguous().view(nsets, R, 2)
    acc = buf[:, 2 * R:2 * R + R * E].contiguous().view(nsets, R, E)
    a, b = f32((R, E)), f32((R, E))
    nv.global_combine(ml, acc, a)
    nv.global_combine_strided(buf, buf.view(-1)[2 * R:], stride, nsets, R, E, b)
    assert torch.equal(a, b)


def test_linear_to_rows_replicates_and_casts():
    M, N, K = 2, 64, 128
    x = synth.normal_like((M, K), 71)
    w = synth.normal_like((N, K), 72, 0.1)
    b = synth.normal_like((N,), 73, 0.1)
    y = f32((M, N))
    nv.linear(torch.from_numpy(x).cuda(), bf(w), bf(b), y)
    dst = torch.zeros((12, N), dtype=torch.bfloat16, device="cuda")
    nv.linear_to_rows(torch.from_numpy(x).cuda(), bf(w), bf(b), dst, 3, 6)
    for k in range(3):
        assert torch.equal(dst[3 + 2 * k:5 + 2 * k], y.to(torch.bfloat16))
    assert float(dst[:3].float().abs().max()) == 0.0 and float(dst[9:].float().abs().max()) == 0.0


@pytest.mark.parametrize("T,H,W,kt,ks", [(8, 6, 6, 4, 3), (4, 27, 27, 4, 3), (16, 27, 27, 4, 3), (8, 6, 6, 2, 3), (8, 6, 6, 4, 2), (12, 9, 6, 4, 3)])
def test_fused_stream_matches_separate_kernels(T, H, W, kt, ks):
    """The fused local+global kernel reproduces hicom_local_attn_fwd and hicom_global_stream_fwd."""
    E, nh, R = D, 9, 9
    x = synth.synth_inputs(T, H, W, D, tag=f"fused{T}{H}{W}{kt}{ks}")
    ff, fe, g = bf(x["ff"]), bf(x["fe"]), bf(x["g"])
    N = T * H * W
    # global operand: random folded queries (hi/lo), positional table
    qt = torch.from_numpy(synth.normal_like((R, E), 81, 0.05)).cuda()
    qhi = torch.zeros((16, E), dtype=torch.bfloat16, device="cuda")
    qlo = torch.zeros_like(qhi)
    nv.split_bf16(qt, 16, qhi, qlo)
    cap = T + 2
    pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
    pos_a = torch.zeros((16, pe.shape[0]), dtype=torch.float32, device="cuda")
    nv.linear(qt, pe, None, pos_a, M=R)
    pe_hi = torch.empty(pe.shape, dtype=torch.bfloat16, device="cuda")
    pe_lo = torch.empty_like(pe_hi)
    nv.split_bf16(pe, pe.shape[0], pe_hi, pe_lo)
    # reference: the two separate kernels
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (geo.axis_tiling(T, kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)))
    nw = (T // kt) * (H // ks) * (W // ks)
    ctx_ref = f32((nw, E))
    nv.local_attn(fe, ff, axes, g, 0, 1 / math.sqrt(E), 0.0, 0, ctx_ref)
    stride = (N + 15) // 16 * 16
    def merged(pm, pl, pacc, scores):
        ml, acc = f32((R, 2)), f32((R, E))
        nv.global_merge(pm, pl, pacc, R, scores, N, H, W, pe, 0, cap, cap + H, f32((R * T * (H + W + 2),)), ml, acc, normalize=True)
        return acc
    np0 = nv.global_stream_nparts(N, 16)
    s0 = f32((16, stride)); pm0, pl0, pa0 = f32((np0, 16)), f32((np0, 16)), f32((np0, 16, E))
    nv.global_stream(ff, N, qhi, qlo, pos_a, H, W, 0, cap, cap + H, s0, pm0, pl0, pa0, rows=R)
    g_ref = merged(pm0, pl0, pa0, s0)
    # fused: local query rows R..15 = guide
    qhi_f = qhi.clone()
    qhi_f[R:] = g
    ran = False
    for nparts in sorted({nv.fused_stream_nparts(nw), max(1, (nw + 15) // 16), min(nw, 3), nw}):
        wpw = (nw + nparts - 1) // nparts
        per_t = (H // ks) * (W // ks)
        span_frames = ((wpw + per_t - 2) // per_t + 1) * kt          # frames one workgroup may touch
        lds = 4 * 9 * 4096 + 9 * 1024 + 8 * 80 * 4 + 8 * 64 + (64 + 32 + 65 + 32 + 16) * 4 + R * (8 + H + W) * 4
        if wpw > 32 or span_frames > 8 or (nparts - 1) * wpw >= nw or lds > 163840 or R * (8 + H + W) > 1024 \
                or 8 + min(H, ((wpw + W // ks - 2) // (W // ks) + 1) * ks) + min(W, wpw * ks) > 64:
            continue                                                  # outside the kernel's LDS-table limits
        ran = True
        pm1, pl1, pa1 = f32((nparts, 16)), f32((nparts, 16)), f32((nparts, 16, E))
        ctx = torch.full((nw, E), float("nan"), dtype=torch.float32, device="cuda")
        nv.fused_stream(ff, fe, kt, ks, qhi_f, qlo, R, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, cap, cap + H, pm1, pl1, pa1, ctx)
        # the kernel folds the value-side pos-emb into its partial contexts: a plain merge finishes the job
        ml, acc = f32((R, 2)), f32((R, E))
        nv.global_merge(pm1, pl1, pa1, R, None, N, H, W, None, 0, 0, 0, None, ml, acc, normalize=True)
        torch.cuda.synchronize()
        assert maxabs(ctx, ctx_ref) <= 5e-5 * max(1.0, float(ctx_ref.abs().max())), (nparts, "local")
        assert maxabs(acc, g_ref) <= 2e-5 * max(1.0, float(g_ref.abs().max())), (nparts, "global")
        # run-to-run determinism (no atomics anywhere)
        pa2, acc2 = torch.empty_like(pa1), f32((R, E))
        nv.fused_stream(ff, fe, kt, ks, qhi_f, qlo, R, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, cap, cap + H, pm1, pl1, pa2, ctx)
        nv.global_merge(pm1, pl1, pa2, R, None, N, H, W, None, 0, 0, 0, None, ml, acc2, normalize=True)
        assert torch.equal(pa1[:, :R], pa2[:, :R]) and torch.equal(acc, acc2), (nparts, "determinism")
        # precomputed local logits (one f32 per token: fe_n . guide, what the head projection's row-dot epilogue hands over)
        # instead of the frames_embed stream (fused_ring_logits_kernel): same windows, same global state
        if lds + (32 + 64) * 4 <= 163840:
            llog = (fe.view(N, E).float() @ g.float()).contiguous()
            ctx3 = torch.full((nw, E), float("nan"), dtype=torch.float32, device="cuda")
            pa3 = torch.empty_like(pa1)
            nv.fused_stream(ff, None, kt, ks, qhi_f, qlo, R, 1 / math.sqrt(E), 0.0, pos_a, pe_hi, pe_lo, 0, cap, cap + H, pm1, pl1, pa3, ctx3,
                            local_logits=llog)
            torch.cuda.synchronize()
            assert maxabs(ctx3, ctx_ref) <= 5e-5 * max(1.0, float(ctx_ref.abs().max())), (nparts, "local from logits")
            assert maxabs(pa3[:, :R], pa1[:, :R]) <= 1e-5 * max(1.0, float(pa1[:, :R].abs().max())), (nparts, "global state with logits")
    assert ran


@pytest.mark.parametrize("T,H,W,nparts_pick", [(8, 6, 6, "max"), (16, 27, 27, "cu"), (64, 27, 27, "cu"), (4, 27, 27, "cu"), (12, 9, 6, "few")])
def test_fused_stream_marginals_out_equal_the_in_kernel_pos_emb(T, H, W, nparts_pick):
    """Round 6: hicom_fused_stream_fwd(part_marg_f16) -- the value-side pos-emb (reference projector.py:636-640 on the VALUE side of
    :215) leaves the ring kernel as NORMALISED t / y / x marginals in absolute slot order [T | H | W | 0...] instead of being multiplied by
    the pe rows behind the token stream.  Per partial and row:  ctx_with_pe = ctx_without_pe + sum_s marg[s] pe_sel[s]  (both fp16
    planes: 2^-11 of the context's magnitude per side, pe entries are O(1)), every row's three axis marginals sum to 1, the padding is
    zero, and the softmax state (m, l) and the local contexts are bit-identical to the pe-tile form."""
    E, R, kt, ks = D, 9, 4, 3
    x = synth.synth_inputs(T, H, W, D, tag=f"marg{T}{H}{W}")
    ff, fe, g = bf(x["ff"]), bf(x["fe"]), bf(x["g"])
    qt = torch.from_numpy(synth.normal_like((R, E), 83, 0.05)).cuda()
    qhi = torch.zeros((16, E), dtype=torch.bfloat16, device="cuda")
    qlo = torch.zeros_like(qhi)
    nv.split_bf16(qt, 16, qhi, qlo)
    qhi[R:] = g
    cap = T + 3
    pe = torch.from_numpy(geo.stacked_pos_tables(cap, H, W, E)).cuda()
    pos_a = torch.zeros((16, pe.shape[0]), dtype=torch.float32, device="cuda")
    nv.linear(qt, pe, None, pos_a, M=R)
    pe_hi = torch.empty(pe.shape, dtype=torch.bfloat16, device="cuda")
    pe_lo = torch.empty_like(pe_hi)
    nv.split_bf16(pe, pe.shape[0], pe_hi, pe_lo)
    nw = (T // kt) * (H // ks) * (W // ks)
    nparts = {"max": nw, "cu": nv.fused_stream_nparts(nw), "few": 3}[nparts_pick]
    S = 144
    f16 = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float16, device="cuda")
    pm1, pl1, pm2, pl2 = f32((nparts, 16)), f32((nparts, 16)), f32((nparts, 16)), f32((nparts, 16))
    c1, c2, p1, p2 = f16(nw, E), f16(nw, E), f16(nparts, 16, E), f16(nparts, 16, E)
    mg = f16(nparts, R, S)
    sc = 1 / math.sqrt(E)
    nv.fused_stream(ff, fe, kt, ks, qhi, qlo, R, sc, 0.0, pos_a, pe_hi, pe_lo, 0, cap, cap + H, pm1, pl1, None, None, ctx_f16=c1, part_ctx_f16=p1)
    nv.fused_stream(ff, fe, kt, ks, qhi, qlo, R, sc, 0.0, pos_a, None, None, 0, cap, cap + H, pm2, pl2, None, None, ctx_f16=c2, part_ctx_f16=p2,
                    part_marg=mg)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2) and torch.equal(pm1[:, :R], pm2[:, :R]) and torch.equal(pl1[:, :R], pl2[:, :R])
    m = mg.float()
    assert bool(torch.isfinite(m).all()) and float(m[..., T + H + W:].abs().max()) == 0.0
    for lo, hi in ((0, T), (T, T + H), (T + H, T + H + W)):
        assert float((m[..., lo:hi].sum(-1) - 1.0).abs().max()) <= 3e-3          # (each axis: a partition of the row's weights; fp16 entries)
    pe_sel = torch.cat([pe[:T], pe[cap:cap + H + W]])                              # [T + H + W, E]
    want = p2[:, :R].float() + torch.einsum("prs,se->pre", m[..., :T + H + W], pe_sel)
    got = p1[:, :R].float()
    assert maxabs(got, want) <= 2.0 ** -9 * max(1.0, float(got.abs().max())), float((got - want).abs().max())
    # run-to-run: bit-identical marginals
    mg2 = f16(nparts, R, S)
    nv.fused_stream(ff, fe, kt, ks, qhi, qlo, R, sc, 0.0, pos_a, None, None, 0, cap, cap + H, pm2, pl2, None, None, ctx_f16=c2, part_ctx_f16=p2,
                    part_marg=mg2)
    assert torch.equal(mg, mg2)


@pytest.mark.parametrize("M,N,K,act", [(8, 64, 1152, 1), (81, 896, 1152, 1), (130, 64, 64, 0), (1296, 896, 896, 0), (1296, 896, 1152, 1),
                                       (50, 200, 128, 1), (97, 66, 192, 0), (49, 130, 256, 1), (300, 257, 320, 0), (48, 128, 384, 1),
                                       (1, 4, 448, 0)])
def test_planes_gemm_matches_torch(M, N, K, act):
    x = synth.normal_like((M, K), 91)
    x = (x + synth.normal_like((M, K), 95) * 2.0 ** -10).astype(np.float32)   # NOT bf16-representable
    w = synth.normal_like((N, K), 92, 0.03)
    b = synth.normal_like((N,), 93, 0.1)
    want = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    if act:
        want = 0.5 * want * (1 + torch.erf(want / math.sqrt(2)))
    a_hi = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    a_lo = torch.empty_like(a_hi)
    nv.split_bf16(torch.from_numpy(x).cuda(), M, a_hi, a_lo)
    o_hi = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    o_lo = torch.empty_like(o_hi)
    y = f32((M + M // 9 + 4, N))
    y.zero_()
    nv.planes_gemm(a_hi, a_lo, bf(w), bf(b), act=act, out_hi=o_hi, out_lo=o_lo, y=y, row0=2, nl_group=9)
    tol = 5e-5 * max(1.0, float(want.abs().max()))
    assert maxabs(o_hi.float() + o_lo.float(), want) <= tol
    rows = torch.tensor([2 + m + m // 9 for m in range(M)])
    assert maxabs(y.cpu()[rows], want) <= tol
    assert float(y[:2].abs().max()) == 0.0
    # bf16 output rows, and the hi-plane-only form (exactly-bf16 activations)
    yb = torch.zeros((M + M // 9 + 4, N), dtype=torch.bfloat16, device="cuda")
    nv.planes_gemm(a_hi, a_lo, bf(w), bf(b), act=act, y=yb, row0=2, nl_group=9)
    assert torch.equal(yb[rows.cuda()], y[rows.cuda()].bfloat16())
    want_hi = a_hi.double().cpu() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    if act:
        want_hi = 0.5 * want_hi * (1 + torch.erf(want_hi / math.sqrt(2)))
    nv.planes_gemm(a_hi, None, bf(w), bf(b), act=act, out_hi=o_hi, out_lo=o_lo)
    assert maxabs(o_hi.float() + o_lo.float(), want_hi) <= tol


@pytest.mark.parametrize("E", [1152, 768, 100, 1528])
def test_row_ln_variants_match_torch(E):
    """E % 8 == 0: the 8-wide kernel (E = 1528: a partly filled third chunk); E = 100: the scalar form"""
    M = 37
    x = torch.from_numpy(synth.normal_like((M, E), 101)).cuda()
    mul = torch.from_numpy(synth.normal_like((1, E), 102, 0.3)).cuda()
    add = torch.from_numpy(synth.normal_like((M, E), 103)).cuda()
    src = bf(synth.normal_like((M, E), 104))
    norm = torch.nn.LayerNorm(E, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(torch.from_numpy(1 + synth.normal_like((E,), 105, 0.1)))
        norm.bias.copy_(torch.from_numpy(synth.normal_like((E,), 106, 0.1)))
    norm_d = torch.nn.LayerNorm(E, eps=1e-6).to(torch.bfloat16).cuda()
    norm_d.load_state_dict(norm.state_dict())
    alpha = torch.tensor([0.5], dtype=torch.bfloat16, device="cuda")
    ln = lambda t: torch.nn.functional.layer_norm(t.double(), (E,), norm_d.weight.double().cpu(), norm_d.bias.double().cpu(), 1e-6)
    out = f32((M, E))
    nv.row_ln(x, norm_d, out, mul=mul, add=add[:1].contiguous())             # coarse: broadcast FiLM rows
    want = ln(x.cpu() * (1 + mul.cpu()) + add[:1].cpu())
    assert maxabs(out, want) <= 2e-5
    nv.row_ln(x, norm_d, out, add=add)                                         # fine: per-row residual
    assert maxabs(out, ln(x.cpu() + add.cpu())) <= 2e-5
    nv.row_ln(x, norm_d, out, src=src, alpha=alpha)                            # adaptor blend
    assert maxabs(out, 0.5 * src.double().cpu() + 0.5 * ln(x.cpu())) <= 2e-5


@pytest.mark.parametrize("M,L,nh,hd", [(21, 64, 9, 128), (1296, 64, 9, 128), (100, 37, 12, 64), (70, 5, 6, 96), (65, 64, 3, 256)])
def test_small_mha_matches_torch(M, L, nh, hd):
    """both forms: the LDS-staged one (head dim <= 128) and one wave per (row, head) for wider heads"""
    E = nh * hd
    q, k, v = (torch.from_numpy(synth.normal_like(s, sd, 0.3)).cuda() for s, sd in (((M, E), 111), ((L, E), 112), ((L, E), 113)))
    out = f32((M, E))
    nv.small_mha(q, k, v, nh, out)
    qh, kh, vh = (t.double().cpu().reshape(-1, nh, hd).permute(1, 0, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(1, 2) * hd ** -0.5, dim=-1)
    want = (p @ vh).permute(1, 0, 2).reshape(M, E)
    assert maxabs(out, want) <= 1e-5


def test_local_attn_fp32_streams():
    """adapt_k / adapt_v hand the local kernel fp32 key / value streams."""
    T, H, W = 4, 6, 6
    x = synth.synth_inputs(T, H, W, D, tag="f32s")
    key = (x["fe"] + synth.normal_like(x["fe"].shape, 121) * 2.0 ** -10).astype(np.float32)
    val = (x["ff"] + synth.normal_like(x["ff"].shape, 122) * 2.0 ** -10).astype(np.float32)
    spec = dict(kt=4, ks=3, adapt_q=False, adapt_k=False, adapt_v=False, adapt_guide=False)
    want, _ = orc.local_context(spec, "direct", {}, "lc", torch.from_numpy(val), torch.from_numpy(key), torch.from_numpy(x["g"]),
                                "video", None, None)
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (geo.axis_tiling(T, 4), geo.axis_tiling(H, 3), geo.axis_tiling(W, 3)))
    ctx = f32((4, D))
    nv.local_attn(torch.from_numpy(key).cuda(), torch.from_numpy(val).cuda(), axes, bf(x["g"]), 0, 1 / math.sqrt(D), 0.0, 0, ctx)
    assert maxabs(ctx, want.reshape(4, D)) <= 2e-5


# ---- round 2: fp16 readout GEMM (+ co-scheduled GEMV), fused merge + v_proj -------------------------------------------
@pytest.mark.parametrize("M,N,K,act", [(8, 64, 1152, 1), (81, 896, 1152, 1), (130, 64, 64, 0), (1296, 896, 896, 0), (1296, 896, 1152, 1),
                                       (648, 3584, 1152, 1), (97, 200, 128, 0),
                                       # round 5: 128-column tiles (chosen when they take fewer rounds of one-per-CU workgroups)
                                       (648, 3584, 3584, 0), (1296, 3584, 3584, 1), (1300, 2176, 192, 0),
                                       # ... and 192 x 128 tiles where even those take two rounds (1296 rows at 3584; a ragged last row block)
                                       (1000, 3584, 1152, 1), (2592, 2048, 896, 0)])
def test_readout16_gemm_matches_torch(M, N, K, act):
    """fp16-plane GEMM against fp64 torch on the SAME fp16-rounded operands (the kernel's arithmetic: exact products,
    fp32 accumulation), both output forms (fp16 plane, packed rows with a newline gap)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=g) * 0.7).cuda()
    w = bf(torch.randn(N, K, generator=g) * 0.02)                               # bf16, on the GPU
    b = bf(torch.randn(N, generator=g) * 0.02)
    a16 = nv.to_f16(x)
    w16 = nv.to_f16(w)
    # bf16 -> fp16 is exact down to |w| = 2^-17 (fp16 subnormals carry multiples of 2^-24); below: <= 2^-25 absolute
    big = w.float().abs() >= 2.0 ** -17
    assert torch.equal(w16.float()[big], w.float()[big]) and float((w16.float() - w.float()).abs().max()) <= 2.0 ** -25
    ref = a16.double() @ w16.double().t() + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    o16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    y = torch.full((M + M // 9 + 3, N), float("nan"), device="cuda")
    nv.readout16_gemm(a16, w16, b, act=act, out_f16=o16, y=y, row0=2, nl_group=9)
    torch.cuda.synchronize()
    rows = (2 + torch.arange(M) + torch.arange(M) // 9).cuda()
    assert maxabs(y[rows], ref) <= 3e-6 * max(1.0, float(ref.abs().max()))
    assert maxabs(o16, ref) <= 2 ** -11 * float(ref.abs().max()) + 1e-6
    untouched = torch.ones(y.shape[0], dtype=torch.bool, device="cuda")
    untouched[rows] = False
    assert bool(torch.isnan(y[untouched]).all())


@pytest.mark.parametrize("M,N,K", [(648, 3584, 1152), (1296, 3584, 3584), (1296, 896, 1152), (1000, 3584, 1152)])
def test_readout16_row_line_epilogue_equals_the_general_one(M, N, K):
    """One 16-bit output (the fp16 plane alone, or packed bf16 rows alone) leaves through the row-line epilogue (LDS image, 16-byte
    stores); both outputs at once through the general one.  Same values: the plane bit for bit, the bf16 rows = the rounded f32 rows.
    (648 x 3584: 96 x 128 tiles; 1296 / 1000 x 3584: 192 x 128; N = 896: 96 x 64.)"""
    g = torch.Generator().manual_seed(M + N + K)
    a16 = nv.to_f16((torch.randn(M, K, generator=g) * 0.7).cuda())
    w16 = nv.to_f16(bf(torch.randn(N, K, generator=g) * 0.02))
    b = bf(torch.randn(N, generator=g) * 0.02)
    o_ref = torch.empty(M, N, dtype=torch.float16, device="cuda")
    y_ref = torch.zeros(M + M // 9 + 3, N, device="cuda")
    nv.readout16_gemm(a16, w16, b, act=nv.ACT_GELU, out_f16=o_ref, y=y_ref, row0=2, nl_group=9)
    o16 = torch.empty_like(o_ref)
    nv.readout16_gemm(a16, w16, b, act=nv.ACT_GELU, out_f16=o16)
    y16 = torch.zeros(M + M // 9 + 3, N, device="cuda", dtype=torch.bfloat16)
    nv.readout16_gemm(a16, w16, b, act=nv.ACT_GELU, y=y16, row0=2, nl_group=9)
    torch.cuda.synchronize()
    assert torch.equal(o16, o_ref)
    assert torch.equal(y16, y_ref.to(torch.bfloat16))


def test_readout16_aux_gemv_and_merge_vproj():
    """The GEMV that rides in the GEMM launch (x summed from partial vectors + bias, bf16 weights, residual, GELU) and the
    fused merge + v_proj kernel against the separate merge / per-head linear chain they replace."""
    g = torch.Generator().manual_seed(5)
    E, nh, nparts = 1152, 9, 216
    pm = torch.randn(nparts, 16, generator=g).cuda()
    pl = (torch.rand(nparts, 16, generator=g) + 0.5).cuda()
    pacc = torch.randn(nparts, 16, E, generator=g).cuda()
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    bv = bf(torch.randn(E, generator=g) * 0.02)
    po = torch.empty(E // 64, E, device="cuda")
    ml, ctx = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
    nv.merge_vproj(pm, pl, pacc, nh, wv, po, ml, ctx)
    # reference chain: merge (normalised) then the per-head linear
    ml2, ctx2 = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
    nv.global_merge(pm, pl, pacc, nh, None, 1, 1, 1, None, 0, 0, 0, None, ml2, ctx2, normalize=True)
    o2 = torch.empty(1, E, device="cuda")
    nv.linear(ctx2, wv, None, o2, head_rows=nh, head_dim=E // nh)
    torch.cuda.synchronize()
    assert maxabs(ctx, ctx2) <= 1e-5 and maxabs(ml, ml2) <= 1e-5
    assert maxabs(po.sum(0), o2[0]) <= 2e-5
    # aux GEMV under a GEMM launch: y = W_o (sum po + b_v) + b_o + res
    wo = bf(torch.randn(E, E, generator=g) * 0.02)
    bo = bf(torch.randn(E, generator=g) * 0.02)
    res = bf(torch.randn(E, generator=g))
    yv = torch.full((E,), float("nan"), device="cuda")
    a16 = nv.to_f16(torch.randn(200, 128, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(64, 128, generator=g).cuda())
    o16 = torch.empty(200, 64, dtype=torch.float16, device="cuda")
    nv.readout16_gemm(a16, w16, None, out_f16=o16, aux=dict(xs=po, xb=bv, w=wo, b=bo, res=res, act=nv.ACT_NONE, y=yv))
    want = (po.sum(0).double() + bv.double()) @ wo.double().t() + bo.double() + res.double()
    torch.cuda.synchronize()
    assert maxabs(yv, want) <= 2e-5
    assert maxabs(o16, a16.double() @ w16.double().t()) <= 2 ** -10 * float((a16.double() @ w16.double().t()).abs().max())
    # GELU + a non-multiple-of-4 column count
    w3 = bf(torch.randn(897 - 1, E, generator=g) * 0.02)
    y3 = torch.empty(896, device="cuda")
    nv.readout16_gemm(a16, w16, None, out_f16=o16, aux=dict(xs=yv.view(1, -1), w=w3, act=nv.ACT_GELU, y=y3))
    torch.cuda.synchronize()
    assert maxabs(y3, torch.nn.functional.gelu(want.float().double() @ w3.double().t())) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("nparts", [216, 7, 256])
def test_merge_vproj_fixed_point_accumulators(nparts):
    """Round 4: merge + v_proj with the slab sums taken inside the launch.  Integer (fixed-point, 2^36) atomic adds: the result
    equals the slab-ordered float sum of the partial-vector form to ~1e-6, is BIT-IDENTICAL from launch to launch (integer addition
    is associative: arrival order cannot matter), and the aux GEMV reads it through x_fixed."""
    g = torch.Generator().manual_seed(50 + nparts)
    E, nh = 1152, 9
    pm = torch.randn(nparts, 16, generator=g).cuda() * 3
    pl = (torch.rand(nparts, 16, generator=g) + 0.5).cuda()
    pacc = torch.randn(nparts, 16, E, generator=g).cuda()
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    bv = bf(torch.randn(E, generator=g) * 0.02)
    po = torch.empty(E // 64, E, device="cuda")
    ml, ctx = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
    nv.merge_vproj(pm, pl, pacc, nh, wv, po, ml, ctx)
    runs = []
    for _ in range(3):
        ofx = torch.zeros(E, dtype=torch.int64, device="cuda")
        ml2, ctx2 = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
        nv.merge_vproj_fixed(pm, pl, pacc, nh, wv, ofx, ml2, ctx2)
        runs.append(ofx)
    torch.cuda.synchronize()
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    o_fixed = runs[0].double() / 2.0 ** 36
    assert maxabs(ctx2, ctx) <= 1e-5 and maxabs(ml2, ml) <= 1e-5
    assert maxabs(o_fixed, po.double().sum(0)) <= 5e-6
    # the partial states as NORMALISED fp16 contexts (what the stream kernel writes on the hot path): acc / l per partial
    p16 = (pacc / pl[:, :, None]).to(torch.float16)
    ofx16 = torch.zeros(E, dtype=torch.int64, device="cuda")
    ctx3 = torch.empty(nh, E, device="cuda")
    nv.merge_vproj_fixed(pm, pl, p16, nh, wv, ofx16, None, ctx3)
    torch.cuda.synchronize()
    assert maxabs(ctx3, ctx) <= 2.0 ** -11 * float(p16.float().abs().max())       # one fp16 rounding per partial, weights sum to 1
    assert maxabs(ofx16.double() / 2.0 ** 36, po.double().sum(0)) <= 1e-3
    # the consumer: y = GELU(W (o + b_v) + b) from the fixed-point vector
    w3 = bf(torch.randn(896, E, generator=g) * 0.02)
    b3 = torch.randn(896, generator=g).cuda()
    y3 = torch.empty(896, device="cuda")
    a16 = nv.to_f16(torch.randn(200, 128, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(64, 128, generator=g).cuda())
    o16 = torch.empty(200, 64, dtype=torch.float16, device="cuda")
    nv.readout16_gemm(a16, w16, None, out_f16=o16, aux=dict(x_fixed=runs[0], xb=bv, w=w3, b=b3, act=nv.ACT_GELU, y=y3))
    torch.cuda.synchronize()
    want = torch.nn.functional.gelu((o_fixed + bv.double()) @ w3.double().t() + b3.double())
    assert maxabs(y3, want) <= 2e-5


@pytest.mark.parametrize("nparts", [216, 7])
def test_merge_role_inside_the_readout_gemm_launch_equals_the_merge_launch(nparts):
    """Round 5: HICOM_ROLE_MERGE_VPROJ -- the (head, slab) items of the merge + v_proj dealt over the workgroups behind readout
    GEMM 1's tile grid.  Same device function as the standalone launch: bit-identical fixed-point sums, (M, L) and contexts; the
    GEMM's own result is untouched."""
    g = torch.Generator().manual_seed(150 + nparts)
    E, nh = 1152, 9
    pm = torch.randn(nparts, 16, generator=g).cuda() * 3
    pl = (torch.rand(nparts, 16, generator=g) + 0.5).cuda()
    p16 = torch.randn(nparts, 16, E, generator=g).cuda().to(torch.float16)
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    ofx_ref = torch.zeros(E, dtype=torch.int64, device="cuda")
    ml_ref, ctx_ref = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
    nv.merge_vproj_fixed(pm, pl, p16, nh, wv, ofx_ref, ml_ref, ctx_ref)
    M, N, K = 1296, 896, 1152
    a16 = nv.to_f16(torch.randn(M, K, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(N, K, generator=g).cuda() * 0.02)
    b = bf(torch.randn(N, generator=g) * 0.02)
    o_ref = torch.empty(M, N, dtype=torch.float16, device="cuda")
    nv.readout16_gemm(a16, w16, b, act=nv.ACT_GELU, out_f16=o_ref)
    for _ in range(2):
        ofx = torch.zeros(E, dtype=torch.int64, device="cuda")
        ml, ctx = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
        o16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        nv.readout16_gemm(a16, w16, b, act=nv.ACT_GELU, out_f16=o16,
                          merge=dict(part_m=pm, part_l=pl, part_ctx16=p16, rows=nh, w_v=wv, o_fix=ofx, out_ml=ml, out_ctx=ctx))
        torch.cuda.synchronize()
        assert torch.equal(ofx, ofx_ref) and torch.equal(ml, ml_ref) and torch.equal(ctx, ctx_ref)
        assert torch.equal(o16, o_ref)


@pytest.mark.parametrize("nparts", [216, 81, 12, 1])
def test_merge_role_applies_the_value_side_pos_emb_from_marginals(nparts):
    """Round 6: HICOM_ROLE_MERGE_VPROJ with part_marg / vpe_f16 -- o = W_v (merged ctx) + sum_s (merged marginal[s]) VPE[:, s], the
    marginals merged with the same weights l_i e^(m_i - M) / L as the contexts.  Against float64 torch; bit-identical from launch to launch."""
    g = torch.Generator().manual_seed(170 + nparts)
    E, nh, S = 1152, 9, 144
    pm = torch.randn(nparts, 16, generator=g).cuda() * 3
    pl = (torch.rand(nparts, 16, generator=g) + 0.5).cuda()
    p16 = torch.randn(nparts, 16, E, generator=g).cuda().to(torch.float16)
    mg = torch.rand(nparts, nh, S, generator=g)
    mg[..., 118:] = 0
    mg = (mg / mg.sum(-1, keepdim=True) * 3).cuda().to(torch.float16)
    vpe = (torch.randn(E, S, generator=g) * 0.5).cuda().to(torch.float16)
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    M, N, K = 1296, 896, 1152
    a16 = nv.to_f16(torch.randn(M, K, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(N, K, generator=g).cuda() * 0.02)
    # reference in float64
    m64, l64 = pm[:, :nh].double(), pl[:, :nh].double()
    wgt = l64 * torch.exp(m64 - m64.max(0, keepdim=True).values)              # [nparts, nh]
    wgt = wgt / wgt.sum(0, keepdim=True)
    ctx = torch.einsum("ph,phe->he", wgt, p16[:, :nh].double())
    mgm = torch.einsum("ph,phs->hs", wgt, mg.double())
    hd = E // nh
    want = torch.einsum("hje,he->hj", wv.double().view(nh, hd, E), ctx) + torch.einsum("hjs,hs->hj", vpe.double().view(nh, hd, S), mgm)
    outs = []
    for _ in range(2):
        ofx = torch.zeros(E, dtype=torch.int64, device="cuda")
        o16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        nv.readout16_gemm(a16, w16, None, act=nv.ACT_GELU, out_f16=o16,
                          merge=dict(part_m=pm, part_l=pl, part_ctx16=p16, rows=nh, w_v=wv, o_fix=ofx, part_marg=mg, vpe_f16=vpe))
        torch.cuda.synchronize()
        outs.append(ofx.clone())
    got = outs[0].double() / 2.0 ** 36
    assert torch.equal(outs[0], outs[1])
    assert maxabs(got, want.reshape(-1)) <= 2e-5 * max(1.0, float(want.abs().max())), float((got - want.reshape(-1)).abs().max())


@pytest.mark.parametrize("n_mid,n_out,tile_rows", [(896, 896, 1296), (896, 896, 200), (3584 // 4, 512, 648), (1536, 1000, 96),
                                                   (3584, 3584, 648), (2048, 520, 200), (4096, 1000, 96)])      # round 6: layers up to 4096 wide
def test_gemv_chain_role_hands_the_hidden_layer_over_inside_the_launch(n_mid, n_out, tile_rows):
    """Round 5: HICOM_ROLE_GEMV_CHAIN -- h = GELU(C (o + b_v) + r0) and y = W2 h + b2 -> replicated output rows in ONE launch, h handed
    over between the role's workgroups as {epoch, value} granules.  Against float64 torch; repeated launches on one state block (the
    epoch advances by itself), results bit-identical from launch to launch; the hand-off failure counter stays 0."""
    g = torch.Generator().manual_seed(7 + n_mid + n_out)
    E = 1152
    o = torch.randn(E, generator=g).double() * 2
    ofx = torch.round(o * 2.0 ** 36).to(torch.int64).cuda()
    bv = bf(torch.randn(E, generator=g) * 0.02)
    c = (torch.randn(n_mid, E, generator=g) * 0.03).cuda()                     # f32 weights (a cached product of weight matrices)
    r0 = torch.randn(n_mid, generator=g).cuda() * 0.1
    w2 = bf(torch.randn(n_out, n_mid, generator=g) * 0.03)
    b2 = bf(torch.randn(n_out, generator=g) * 0.02)
    M, N, K = tile_rows, 128, 128
    a16 = nv.to_f16(torch.randn(M, K, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(N, K, generator=g).cuda())
    state = nv.r16_chain_state(n_mid, "cuda")
    h_want = torch.nn.functional.gelu((ofx.cpu().double() / 2.0 ** 36 + bv.cpu().double()) @ c.cpu().double().t() + r0.cpu().double())
    y_want = h_want @ w2.cpu().double().t() + b2.cpu().double()
    outs = []
    for rep in range(4):
        dst = torch.full((40, n_out + 8), 7.0, device="cuda")
        hbuf = torch.empty(n_mid, device="cuda")
        o16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        nv.readout16_gemm(a16, w16, None, out_f16=o16,
                          chain=(dict(x_fixed=ofx, xb=bv, w=c, b=r0, act=nv.ACT_GELU, y=hbuf),
                                 dict(w=w2, b=b2, act=nv.ACT_NONE, rows=(dst, 3, 32)), state))
        torch.cuda.synchronize()
        assert maxabs(hbuf, h_want) <= 2e-5
        assert maxabs(dst[3:35, :n_out], y_want.expand(32, n_out)) <= 5e-5 * max(1.0, float(y_want.abs().max()))
        assert bool((dst[:3] == 7).all()) and bool((dst[35:] == 7).all()) and bool((dst[:, n_out:] == 7).all())
        assert maxabs(o16.float(), a16.float() @ w16.float().t()) <= 0.05 * K ** 0.5
        outs.append(dst.clone())
    assert all(torch.equal(outs[0], x) for x in outs[1:])
    words = state[:16].view(torch.int32)
    assert int(words[2]) == 0                                                  # no failed hand-off
    assert int(state[:8].view(torch.int64)[0]) == 4 * int(words[3])            # four launches' worth of role arrivals


def test_fused_stream_clears_the_scratch_it_is_given():
    """hicom_fused_stream_fwd's zero_ptr: workgroup 0 clears the accumulators of the merge + v_proj launch behind it."""
    T, H, W, kt, ks, R = 8, 6, 6, 4, 3, 9
    x = synth.synth_inputs(T, H, W, D, tag="fz")
    ff, fe = bf(x["ff"]), bf(x["fe"])
    qhi = bf(torch.randn(16, D) * 0.05)
    qlo = torch.zeros_like(qhi)
    nw = (T // kt) * (H // ks) * (W // ks)
    nparts = nv.fused_stream_nparts(nw)
    pm, pl, pa = torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, device="cuda"), torch.empty(nparts, 16, D, device="cuda")
    ctx = torch.empty(nw, D, device="cuda")
    z = torch.full((D,), 0x7FFFFFFF, dtype=torch.int64, device="cuda")
    guard = torch.full((16,), 77, dtype=torch.int64, device="cuda")
    buf = torch.cat([guard, z, guard])
    nv.fused_stream(ff, fe, kt, ks, qhi, qlo, R, 1 / math.sqrt(D), 0.0, None, None, None, 0, T, T + H, pm, pl, pa, ctx, zero=buf[16:16 + D])
    torch.cuda.synchronize()
    assert int(buf[16:16 + D].abs().sum()) == 0 and bool((buf[:16] == 77).all()) and bool((buf[-16:] == 77).all())


@pytest.mark.parametrize("T,H,W,kt,ks,shared_query", [(8, 6, 6, 4, 3, True), (7, 7, 5, 4, 3, False), (4, 6, 6, 4, 3, False)])
def test_local_attn_bwd_matches_torch_autograd(T, H, W, kt, ks, shared_query):
    """dq per window and d key of the windowed attention -- any geometry, overlapping windows included (round 6: d key accumulated over the
    shared planes by one launch per parity class) -- against torch autograd on the oracle's window gather (double precision)."""
    x = synth.synth_inputs(T, H, W, D, tag=f"lb{T}{H}{W}")
    at, ay, ax = geo.axis_tiling(T, kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
    nw = at.nwin * ay.nwin * ax.nwin
    exact = all(a.nwin * a.k == a.n for a in (at, ay, ax))
    kb, vb = bf(x["fe"]), bf(x["ff"])
    idx = orc.window_token_index(T, H, W, kt, ks).cuda()                              # [nw, win] token ids
    key = kb.double().reshape(-1, D).requires_grad_(True)
    val = vb.double().reshape(-1, D)
    if shared_query:
        qd = bf(x["g"])
        q = qd.double().reshape(1, D).requires_grad_(True)
        qw = q.expand(nw, D)
    else:
        qd = torch.from_numpy(synth.normal_like((nw, D), 77)).cuda()
        q = qd.double().requires_grad_(True)
        qw = q
    scale = 1 / math.sqrt(D)
    s = torch.einsum("wd,wnd->wn", qw, key[idx]) * scale
    ctx = torch.einsum("wn,wnd->wd", torch.softmax(s, dim=1), val[idx])
    dctx = torch.from_numpy(synth.normal_like((nw, D), 78)).cuda()
    (ctx * dctx.double()).sum().backward()
    dq = f32((nw, D))
    dkey = torch.full_like(kb, 3.0)                                                  # (stale content: the overlap path clears it itself)
    nv.local_attn_bwd(kb, vb, axes, qd.reshape(-1) if shared_query else qd, 0 if shared_query else D, scale, 0.0, dctx, dq, dkey)
    torch.cuda.synchronize()
    want_dq = q.grad if not shared_query else None
    if shared_query:
        assert maxabs(dq.sum(0), q.grad.reshape(-1)) <= 2e-5 * max(1.0, float(q.grad.abs().max()))
    else:
        assert maxabs(dq, want_dq) <= 2e-5 * max(1.0, float(want_dq.abs().max()))
    mx = float(key.grad.abs().max())
    # bf16 store of the fp32 value; a token shared by up to 8 windows is rounded once per contribution
    assert maxabs(dkey.float().reshape(-1, D), key.grad) <= (2 ** -8 if exact else 2 ** -6) * mx + 1e-7
    # d value = p_i dctx_w, with the key gradient added where the keys ARE the value rows
    dv = torch.full_like(kb, -2.0)
    nv.local_attn_bwd(vb, vb, axes, qd.reshape(-1) if shared_query else qd, 0 if shared_query else D, scale, 0.0, dctx, dq, None, dvalue=dv, value_is_key=True)
    v2 = vb.double().reshape(-1, D).requires_grad_(True)
    s2 = torch.einsum("wd,wnd->wn", qw.detach(), v2[idx]) * scale
    (torch.einsum("wn,wnd->wd", torch.softmax(s2, dim=1), v2[idx]) * dctx.double()).sum().backward()
    assert maxabs(dv.float().reshape(-1, D), v2.grad) <= (2 ** -8 if exact else 2 ** -6) * float(v2.grad.abs().max()) + 1e-7


def test_small_op_dispatch_boundaries_random_sweep():
    """Seeded sweep across the dispatch boundaries of the small operators (GEMV / MFMA / 8-row linears, 8-wide / scalar LayerNorm,
    LDS / per-wave MHA): every shape against a float64 torch reference."""
    rng = np.random.default_rng(20250614)
    for it in range(24):
        M = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 70]))
        N = int(rng.choice([16, 48, 100, 118, 128, 896, 1152]))
        K = int(rng.choice([8, 24, 32, 64, 96, 1152]))
        xdt = str(rng.choice(["f32", "bf16"]))
        x = synth.normal_like((M, K), 1000 + it)
        w = synth.normal_like((N, K), 2000 + it, 0.05)
        b = synth.normal_like((N,), 3000 + it, 0.1)
        dx = bf(x) if xdt == "bf16" else torch.from_numpy(x).cuda()
        dw, db = bf(w), bf(b)
        y = f32((M, N))
        nv.linear(dx, dw, db, y, act=nv.ACT_GELU)
        ref = dx.double().cpu() @ dw.double().cpu().t() + db.double().cpu()
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
        assert maxabs(y, ref) <= 3e-5 * max(1.0, float(ref.abs().max())), (M, N, K, xdt)
    for E in (8, 64, 100, 768, 1152, 1160, 1536, 2048):
        M = int(rng.integers(1, 40))
        x = torch.from_numpy(synth.normal_like((M, E), 4000 + E)).cuda()
        add = torch.from_numpy(synth.normal_like((M, E), 5000 + E)).cuda()
        norm = torch.nn.LayerNorm(E, eps=1e-6).to(torch.bfloat16).cuda()
        with torch.no_grad():
            norm.weight.copy_(torch.from_numpy(1 + synth.normal_like((E,), 6000 + E, 0.1)))
            norm.bias.copy_(torch.from_numpy(synth.normal_like((E,), 7000 + E, 0.1)))
        out = f32((M, E))
        nv.row_ln(x, norm, out, add=add)
        ref = torch.nn.functional.layer_norm((x + add).double().cpu(), (E,), norm.weight.detach().double().cpu(), norm.bias.detach().double().cpu(), 1e-6)
        assert maxabs(out, ref) <= 3e-5, E
    for (M, L, nh, hd) in ((1, 1, 9, 128), (7, 64, 9, 128), (8, 63, 9, 128), (9, 2, 12, 64), (33, 17, 8, 96), (40, 64, 4, 32)):
        E = nh * hd
        q, k, v = (torch.from_numpy(synth.normal_like(s, sd, 0.3)).cuda() for s, sd in (((M, E), 8100 + M), ((L, E), 8200 + M), ((L, E), 8300 + M)))
        out = f32((M, E))
        nv.small_mha(q, k, v, nh, out)
        qh, kh, vh = (t.double().cpu().reshape(-1, nh, hd).permute(1, 0, 2) for t in (q, k, v))
        pr = torch.softmax(qh @ kh.transpose(1, 2) * hd ** -0.5, dim=-1)
        assert maxabs(out, (pr @ vh).permute(1, 0, 2).reshape(M, E)) <= 1e-5, (M, L, nh, hd)


# ---- round 3: one-launch query prep (in-grid granule hand-off), f32-weight aux GEMV with row output ---------------------
@pytest.mark.parametrize("P,hidden", [(310, 896), (0, 3584), (70, 64)])
def test_query_prep_matches_torch(P, hidden):
    """hicom_query_prep_fwd: q_proj -> (granule hand-off inside the grid) -> fold, positional table, local query rows, r0
    against fp64 torch.  Run as a train of back-to-back launches with alternating guides and NO host sync between them:
    every launch must wait for ITS q_proj outputs (epoch tags), never read those of the launch before."""
    g = torch.Generator().manual_seed(17 + P)
    E, nh = D, 9
    hd = E // nh
    wq, wk = bf(torch.randn(E, E, generator=g) * 0.02), bf(torch.randn(E, E, generator=g) * 0.02)
    bq = bf(torch.randn(E, generator=g) * 0.1)
    kpe = (torch.randn(E, P, generator=g) * 0.3).cuda() if P else None
    gw0, gb0, bo = bf(torch.randn(hidden, E, generator=g) * 0.02), bf(torch.randn(hidden, generator=g) * 0.1), bf(torch.randn(E, generator=g) * 0.1)
    guides = [bf(torch.randn(E, generator=g)) for _ in range(3)]
    scale = hd ** -0.5
    state = nv.query_prep_state(E, "cuda")

    def reference(gd):
        qp = wq.double() @ gd.double() + bq.double()
        qt = scale * torch.einsum("hje,hj->he", wk.double().view(nh, hd, E), qp.view(nh, hd))
        pa = scale * torch.einsum("hjp,hj->hp", kpe.double().view(nh, hd, P), qp.view(nh, hd)) if P else None
        r0 = gw0.double() @ (bo.double() + gd.double()) + gb0.double()
        return qt, pa, r0

    outs = []
    for it in range(12):
        gd = guides[it % 3]
        qhi = torch.zeros(16, E, dtype=torch.bfloat16, device="cuda")
        qlo = torch.zeros_like(qhi)
        pos_a = torch.zeros(16, P, device="cuda") if P else None
        r0 = torch.empty(hidden, device="cuda")
        nv.query_prep(gd, gd, wq, bq, wk, kpe, nh, scale, qhi, qlo, pos_a, state, gw0, gb0, bo, r0)
        outs.append((it % 3, qhi, qlo, pos_a, r0))
    torch.cuda.synchronize()
    assert int(state[:12].view(torch.int32)[2]) == 0                   # no spin gave up
    arrivals = int(state[:8].view(torch.int64)[0])                     # one arrival per workgroup and launch
    assert arrivals > 0 and arrivals % 12 == 0
    refs = [reference(gd) for gd in guides]
    for k, qhi, qlo, pos_a, r0 in outs:
        qt, pa, rr = refs[k]
        got = qhi[:nh].double() + qlo[:nh].double()
        assert float((got.cpu() - qt.cpu()).abs().max()) <= 2 ** -15 * float(qt.abs().max()) + 1e-6
        assert torch.equal(qhi[nh:], guides[k].view(1, E).expand(16 - nh, E)) and float(qlo[nh:].float().abs().max()) == 0.0
        if P:
            assert maxabs(pos_a[:nh], pa) <= 1e-5 * max(1.0, float(pa.abs().max())) and float(pos_a[nh:].abs().max()) == 0.0
        assert maxabs(r0, rr) <= 1e-5 * max(1.0, float(rr.abs().max()))


def test_readout16_aux_gemv_f32_weights_and_row_output():
    """The aux role with an f32 weight matrix / f32 bias (the cached product readout[0] . out_proj of the five-launch step)
    and with its result replicated into packed output rows (the 32 global rows), bf16 and f32 destinations."""
    g = torch.Generator().manual_seed(23)
    E, N = 1152, 896
    po = torch.randn(E // 64, E, generator=g).cuda()
    bv = bf(torch.randn(E, generator=g) * 0.02)
    cw = (torch.randn(N, E, generator=g) * 0.02).cuda()
    r0 = torch.randn(N, generator=g).cuda()
    a16 = nv.to_f16(torch.randn(200, 128, generator=g).cuda())
    w16 = nv.to_f16(torch.randn(64, 128, generator=g).cuda())
    o16 = torch.empty(200, 64, dtype=torch.float16, device="cuda")
    hid = torch.full((N,), float("nan"), device="cuda")
    nv.readout16_gemm(a16, w16, None, out_f16=o16, aux=dict(xs=po, xb=bv, w=cw, b=r0, act=nv.ACT_GELU, y=hid))
    want = torch.nn.functional.gelu(cw.double() @ (po.sum(0).double() + bv.double()) + r0.double())
    torch.cuda.synchronize()
    assert maxabs(hid, want) <= 2e-5
    w2, b2 = bf(torch.randn(N, N, generator=g) * 0.02), bf(torch.randn(N, generator=g) * 0.1)
    want2 = (w2.double() @ hid.double() + b2.double()).cpu()
    for dt in (torch.float32, torch.bfloat16):
        out = torch.full((50, N), 7.0, dtype=dt, device="cuda")
        nv.readout16_gemm(a16, w16, None, out_f16=o16, aux=dict(xs=hid.view(1, -1), w=w2, b=b2, rows=(out, 11, 32)))
        torch.cuda.synchronize()
        assert bool((out[:11] == 7.0).all()) and bool((out[43:] == 7.0).all())
        assert torch.equal(out[11:43], out[11:12].expand(32, N))
        tol = 2e-5 if dt == torch.float32 else 2 ** -8 * float(want2.abs().max())
        assert float((out[11].double().cpu() - want2).abs().max()) <= tol
    assert maxabs(o16, a16.double() @ w16.double().t()) <= 2 ** -10 * float((a16.double() @ w16.double().t()).abs().max())


def test_streaming_backward_kernels_leave_their_column_sums():
    """hicom_gelu_bwd_fwd / hicom_adapt_dy_fwd with col_parts: the bias gradients (column sums of what the launch writes, as stored)
    come out of the same launch -- equal to summing the written matrix, and the matrix itself equals the launch without them;
    hicom_partials_sum_fwd in its many-partials form against torch."""
    g = torch.Generator(device="cuda").manual_seed(2)
    N, Dm = 16 * 6 * 6, 1152
    h = (torch.randn(N, Dm, device="cuda", generator=g) * 1.5).half()
    da0 = torch.randn(N, Dm, device="cuda", generator=g).bfloat16()
    a, b = da0.clone(), da0.clone()
    assert nv.gelu_bwd_(a, h) is None
    cs = nv.gelu_bwd_(b, h, colsum=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    hf = h.float()
    want = da0.float() * (0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-hf * hf / 2) / math.sqrt(2 * math.pi))
    assert maxabs(a.float(), want) <= 2 ** -7 * float(want.abs().max())
    assert maxabs(cs, b.float().sum(0)) <= 1e-4 * max(1.0, float(b.float().sum(0).abs().max()))
    # adapt_dy on a 16 x 6 x 6 grid with 4 x 3 x 3 windows
    axes = tuple(nv.Axis(ax.n, ax.k, ax.nwin, ax.nfull) for ax in (geo.axis_tiling(16, 4), geo.axis_tiling(6, 3), geo.axis_tiling(6, 3)))
    y = torch.randn(N, Dm, device="cuda", generator=g).half()
    gamma = (1 + 0.1 * torch.randn(Dm, device="cuda", generator=g)).bfloat16()
    vec = torch.randn(16, Dm, device="cuda", generator=g)
    coef = torch.randn(N, device="cuda", generator=g)
    alpha = torch.full((1,), 0.5, device="cuda").bfloat16()
    d0, d1 = torch.empty(N, Dm, device="cuda", dtype=torch.bfloat16), torch.empty(N, Dm, device="cuda", dtype=torch.bfloat16)
    nv.adapt_dy(y, gamma, vec, Dm, coef, alpha, axes, d0)
    cs = nv.adapt_dy(y, gamma, vec, Dm, coef, alpha, axes, d1, colsum=True)
    torch.cuda.synchronize()
    assert torch.equal(d0, d1) and float(d0.float().abs().max()) > 0
    assert maxabs(cs, d1.float().sum(0)) <= 1e-4 * max(1.0, float(d1.float().sum(0).abs().max()))
    for nparts, M in ((1024, 1152), (100, 70), (64, 16)):
        parts = torch.randn(nparts, M, device="cuda", generator=g)
        out = torch.empty(M, device="cuda")
        nv.partials_sum(parts, out)
        assert maxabs(out, parts.double().sum(0).float()) <= 1e-4


@pytest.mark.parametrize("which", ["both", "key", "value"])
@pytest.mark.parametrize("T,H,W,kt,ks,shared_query", [(4, 6, 6, 4, 3, True), (7, 6, 9, 4, 3, False), (1, 4, 4, 1, 2, False)])
def test_local_attn_adapt_matches_torch(T, H, W, kt, ks, shared_query, which):
    """hicom_local_attn_adapt_fwd: the LayerNorm + alpha blend of the k / v adaptors (reference projector.py:533-534) fused into the
    window attention's row loads, against the materialised blend in torch fp64 followed by the window attention of the oracle
    geometry (exact and overlapping partitions, shared and per-window queries, either stream alone)."""
    D = 1152
    at, ay, ax = geo.axis_tiling(T, kt), geo.axis_tiling(H, ks), geo.axis_tiling(W, ks)
    axes = tuple(nv.Axis(a.n, a.k, a.nwin, a.nfull) for a in (at, ay, ax))
    nw = at.nwin * ay.nwin * ax.nwin
    g = torch.Generator().manual_seed(T * 100 + H + (3 if which == "key" else 0))
    kx = torch.randn(T, H, W, D, generator=g).to(torch.bfloat16)
    vx = torch.randn(T, H, W, D, generator=g).to(torch.bfloat16)
    ky = (torch.randn(T, H, W, D, generator=g) * 1.5 + 0.3).to(torch.float16)
    vy = (torch.randn(T, H, W, D, generator=g) * 0.7 - 0.2).to(torch.float16)
    norms = []
    for _ in range(2):
        n = torch.nn.LayerNorm(D, eps=1e-6)
        with torch.no_grad():
            n.weight.copy_(1 + 0.1 * torch.randn(D, generator=g)); n.bias.copy_(0.1 * torch.randn(D, generator=g))
        norms.append(n.to(torch.bfloat16).cuda())
    ak, av = torch.tensor([0.4]).to(torch.bfloat16).cuda(), torch.tensor([0.7]).to(torch.bfloat16).cuda()
    q = (torch.randn(1 if shared_query else nw, D, generator=g)).to(torch.bfloat16 if shared_query else torch.float32)
    use_k, use_v = which in ("both", "key"), which in ("both", "value")
    ctx = torch.empty(nw, D, device="cuda")
    scale = 1.0 / math.sqrt(D)
    nv.local_attn_adapt(kx.cuda(), ky.cuda() if use_k else None, norms[0] if use_k else None, ak if use_k else None,
                        vx.cuda(), vy.cuda() if use_v else None, norms[1] if use_v else None, av if use_v else None,
                        axes, q.cuda().reshape(-1) if shared_query else q.cuda(), 0 if shared_query else D, scale, 0.0, ctx)
    torch.cuda.synchronize()

    def blend(x, y, n, a):
        ln = torch.nn.functional.layer_norm(y.double(), (D,), n.weight.detach().double().cpu(), n.bias.detach().double().cpu(), 1e-6)
        return (1 - float(a.float())) * x.double() + float(a.float()) * ln
    K = blend(kx, ky, norms[0], ak) if use_k else kx.double()
    V = blend(vx, vy, norms[1], av) if use_v else vx.double()
    want = torch.empty(nw, D, dtype=torch.float64)
    for w in range(nw):
        t1, r = divmod(w, ay.nwin * ax.nwin)
        h1, w1 = divmod(r, ax.nwin)
        t0, y0, x0 = at.start(t1), ay.start(h1), ax.start(w1)
        kw = K[t0:t0 + at.k, y0:y0 + ay.k, x0:x0 + ax.k].reshape(-1, D)
        vw = V[t0:t0 + at.k, y0:y0 + ay.k, x0:x0 + ax.k].reshape(-1, D)
        qq = q[0 if shared_query else w].double()
        p = torch.softmax(kw @ qq * scale, 0)
        want[w] = p @ vw
    assert float((ctx.double().cpu() - want).abs().max()) <= 2e-4


@pytest.mark.parametrize("M,dt,act", [(729 * 4, torch.bfloat16, nv.ACT_GELU), (1000, torch.float16, nv.ACT_NONE)])
def test_dense16_pair_launch_equals_two_launches(M, dt, act):
    """Round 5: hicom_dense16_gemm_pair_fwd -- the same layer of the k and of the v adaptor MLP (reference projector.py:533-534) as ONE
    launch of two problems; every output bit-identical to the single-problem launch (same tiles, same arithmetic), ragged M included."""
    g = torch.Generator().manual_seed(M)
    E = 1152
    a_k, a_v = (torch.randn(M, E, generator=g).cuda().to(dt) for _ in range(2))
    w_k, w_v = ((torch.randn(E, E, generator=g) * 0.03).cuda().to(dt) for _ in range(2))
    b_k, b_v = (bf(torch.randn(E, generator=g) * 0.05) for _ in range(2))
    ref_k, ref_v = (torch.empty(M, E, dtype=torch.float16, device="cuda") for _ in range(2))
    nv.dense16_gemm(a_k, w_k, b_k, act=act, out_f16=ref_k)
    nv.dense16_gemm(a_v, w_v, b_v, act=act, out_f16=ref_v)
    out_k, out_v = (torch.full((M, E), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))
    nv.dense16_gemm_pair(a_k, w_k, b_k, out_k, a_v, w_v, b_v, out_v, act=act)
    torch.cuda.synchronize()
    assert torch.equal(out_k, ref_k) and torch.equal(out_v, ref_v)
    want = a_k.float() @ w_k.float().t() + b_k.float()
    if act == nv.ACT_GELU:
        want = torch.nn.functional.gelu(want)
    assert maxabs(out_k, want) <= 2e-2 * max(1.0, float(want.abs().max()))


def test_merge_over_shard_states_and_chain_launch_alone():
    """Round 5, FINISH phase of the frame-sharded step: hicom_merge_vproj_sets_fwd (the merge item over `world` gathered shard states
    [(M, L) pairs | un-normalised ACC], strided) against the combine + per-head v_proj of float64 torch, and hicom_gemv_chain_fwd (the
    two-layer chain as a launch of its own) fed from its fixed-point result."""
    g = torch.Generator().manual_seed(77)
    E, nh, world, hid = 1152, 9, 8, 896
    hd = E // nh
    stride = 2 * nh + nh * E + 6                      # (a packed send buffer is longer than the state: tokens follow)
    stride += stride % 2
    sets = torch.zeros(world, stride)
    M = torch.randn(world, nh, generator=g) * 3
    L = torch.rand(world, nh, generator=g) + 0.5
    ACC = torch.randn(world, nh, E, generator=g) * L[:, :, None]
    sets[:, 0:2 * nh:2], sets[:, 1:2 * nh:2] = M, L
    sets[:, 2 * nh:2 * nh + nh * E] = ACC.reshape(world, -1)
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    bv = bf(torch.randn(E, generator=g) * 0.02)
    Mx = M.max(0).values
    w = torch.exp(M - Mx)                                                         # [world, nh]
    ctx = (w[:, :, None] * ACC).sum(0).double() / (w * L).sum(0).double()[:, None]    # [nh, E]
    o_want = torch.cat([ctx[h] @ wv.cpu().double()[h * hd:(h + 1) * hd].t() for h in range(nh)])
    ofx = torch.zeros(E, dtype=torch.int64, device="cuda")
    ml, cx = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
    nv.merge_vproj_sets(sets.cuda(), nh, E, wv, ofx, ml, cx)
    torch.cuda.synchronize()
    assert maxabs(cx, ctx) <= 1e-5 and maxabs(ml[:, 0], Mx) == 0
    assert maxabs(ofx.double() / 2.0 ** 36, o_want) <= 1e-5
    c = (torch.randn(hid, E, generator=g) * 0.03).cuda()
    r0 = torch.randn(hid, generator=g).cuda() * 0.1
    w2 = bf(torch.randn(hid, hid, generator=g) * 0.03)
    b2 = bf(torch.randn(hid, generator=g) * 0.02)
    state = nv.r16_chain_state(hid, "cuda")
    h_want = torch.nn.functional.gelu((ofx.cpu().double() / 2.0 ** 36 + bv.cpu().double()) @ c.cpu().double().t() + r0.cpu().double())
    y_want = h_want @ w2.cpu().double().t() + b2.cpu().double()
    for _ in range(3):
        dst = torch.full((40, hid), 7.0, device="cuda")
        nv.gemv_chain(dict(x_fixed=ofx, xb=bv, w=c, b=r0, act=nv.ACT_GELU), dict(w=w2, b=b2, act=nv.ACT_NONE, rows=(dst, 5, 32)), state)
        torch.cuda.synchronize()
        assert maxabs(dst[5:37], y_want.expand(32, hid)) <= 5e-5 * max(1.0, float(y_want.abs().max()))
        assert bool((dst[:5] == 7).all()) and bool((dst[37:] == 7).all())
    assert int(state[:16].view(torch.int32)[2]) == 0


@pytest.mark.parametrize("M", [1296, 96, 700])
def test_fused_tail_launch_equals_the_two_role_launches(M):
    """Round 5: hicom_readout_tail_fwd -- tile workgroups that compute a tile of GEMM 1 (publishing the fp16 hidden plane) and then the
    same tile of GEMM 2 (consuming the plane behind its row block's counter), beside the merge -> chain role, in ONE grid.  Same tile and role device functions as the two launches: every output
    bit-identical, on every one of a run of launches whose inputs CHANGE from launch to launch (a consumer that read a stale or an
    unpublished line of the plane, or a chain that read x before the last merge item, would differ); counters: no failed wait."""
    g = torch.Generator().manual_seed(31 + M)
    E, nh, H, nparts = 1152, 9, 896, 216
    wv = bf(torch.randn(E, E, generator=g) * 0.02)
    bv = bf(torch.randn(E, generator=g) * 0.02)
    w1 = nv.to_f16(torch.randn(H, E, generator=g).cuda() * 0.03)
    b1 = bf(torch.randn(H, generator=g) * 0.02)
    w2 = nv.to_f16(torch.randn(H, H, generator=g).cuda() * 0.03)
    b2 = bf(torch.randn(H, generator=g) * 0.02)
    c = (torch.randn(H, E, generator=g) * 0.03).cuda()
    r0 = torch.randn(H, generator=g).cuda() * 0.1
    gw2 = bf(torch.randn(H, H, generator=g) * 0.03)
    gb2 = bf(torch.randn(H, generator=g) * 0.02)
    st_ref, st_tail, sync = nv.r16_chain_state(H, "cuda"), nv.r16_chain_state(H, "cuda"), nv.readout_tail_state("cuda")
    hid_ref = torch.empty(M, H, dtype=torch.float16, device="cuda")
    hid = torch.empty(M, H, dtype=torch.float16, device="cuda")                # ONE plane, re-used by every launch (as the workspace is)
    nrep = 12
    for rep in range(nrep):
        pm = torch.randn(nparts, 16, generator=g).cuda() * 3
        pl = (torch.rand(nparts, 16, generator=g) + 0.5).cuda()
        p16 = torch.randn(nparts, 16, E, generator=g).cuda().to(torch.float16)
        a16 = nv.to_f16(torch.randn(M, E, generator=g).cuda())

        def run(fused, hidbuf, state):
            ofx = torch.zeros(E, dtype=torch.int64, device="cuda")
            ml, ctx = torch.empty(nh, 2, device="cuda"), torch.empty(nh, E, device="cuda")
            hg = torch.empty(H, device="cuda")
            y = torch.full((M + 40, H), 7.0, device="cuda", dtype=torch.bfloat16)
            merge = dict(part_m=pm, part_l=pl, part_ctx16=p16, rows=nh, w_v=wv, o_fix=ofx, out_ml=ml, out_ctx=ctx)
            chain = (dict(x_fixed=ofx, xb=bv, w=c, b=r0, act=nv.ACT_GELU, y=hg), dict(w=gw2, b=gb2, act=nv.ACT_NONE, rows=(y, M + 2, 32)), state)
            if fused:
                nv.readout_tail(a16, w1, b1, hidbuf, w2, b2, y, merge, chain, sync)
            else:
                nv.readout16_gemm(a16, w1, b1, act=nv.ACT_GELU, out_f16=hidbuf, merge=merge)
                nv.readout16_gemm(hidbuf, w2, b2, y=y, chain=chain)
            torch.cuda.synchronize()
            return ofx, ml, ctx, hg, y, hidbuf.clone()

        want = run(False, hid_ref, st_ref)
        got = run(True, hid, st_tail)
        for k, (a, b) in enumerate(zip(want, got)):
            assert torch.equal(a, b), (rep, k)
        assert bool((got[4][M:M + 2] == 7).all()) and bool((got[4][M + 34:] == 7).all())
    words = sync.view(torch.int64)
    assert int(words[1]) == 0 and int(st_tail[:16].view(torch.int32)[2]) == 0                  # no failed wait, no failed hand-off
    nby, grid = (M + 95) // 96, 8 * ((14 * ((M + 95) // 96) + 7) // 8) + 54
    assert int(words[0]) == nrep * grid and int(words[16]) == nrep * 54
    assert all(int(words[32 + 16 * r]) == nrep * 14 for r in range(nby))
    flags = words[16 * 66:16 * 66 + 14 * nby]
    assert bool((flags == 2 * nrep).all())                                                      # every GEMM-2 tile owned (none abandoned) in the last launch


def test_fused_tail_launch_refuses_shapes_outside_its_form():
    """fp32 output rows (no row-line epilogue) are outside the fused form: HICOM_EUNSUP, nothing launched -- the executor then issues
    the two role launches."""
    E, nh, H, M = 1152, 9, 896, 96
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")
    merge = dict(part_m=z(8, 16), part_l=z(8, 16) + 1, part_ctx16=z(8, 16, E, dt=torch.float16), rows=nh, w_v=z(E, E, dt=torch.bfloat16), o_fix=z(E, dt=torch.int64))
    chain = (dict(x_fixed=merge["o_fix"], w=z(H, E), act=nv.ACT_GELU), dict(w=z(H, H, dt=torch.bfloat16), act=nv.ACT_NONE, rows=(z(40, H), 0, 32)),
             nv.r16_chain_state(H, "cuda"))
    with pytest.raises(nv.HicomNativeError):
        nv.readout_tail(z(M, E, dt=torch.float16), z(H, E, dt=torch.float16), None, z(M, H, dt=torch.float16), z(H, H, dt=torch.float16), None,
                        z(M, H), merge, chain, nv.readout_tail_state("cuda"))
