"""the SigLIP pooling-head projection that produces frames_embed (reference encoder.py:284-286) on the
dense MFMA GEMM, against (a) the fixture made from those two reference lines on HF's head module
(tests/golden/golden_head_v1.npz) and (b) the oracle restatement; plus the dense GEMM operator itself.

Tolerance 1e-3 max-abs on the fp32 result (fp16 operands: 2^-12 relative rounding of the normalised activations and of the
hidden layer; the residual is exact); the bf16 result is its round-to-nearest cast."""
import os

import numpy as np
import pytest
import torch

import make_golden_head as mh
from hicom_amd import native as nv
from oracle import hicom_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Head(torch.nn.Module):
    """Stand-in with the attribute names of HF's SiglipMultiheadAttentionPoolingHead that encoder.py:284-285 touches."""
    def __init__(self, sd):
        super().__init__()
        D, I = mh.D, mh.INTER
        self.layernorm = torch.nn.LayerNorm(D, eps=1e-6)
        self.mlp = torch.nn.Module()
        self.mlp.fc1 = torch.nn.Linear(D, I)
        self.mlp.fc2 = torch.nn.Linear(I, D)
        self.load_state_dict({k[len("head."):]: torch.from_numpy(v.copy()) for k, v in sd.items()})


@pytest.fixture(scope="module")
def head():
    sd = mh.head_state_dict()
    return _Head(sd).to(torch.bfloat16).cuda().eval(), {k: torch.from_numpy(v) for k, v in sd.items()}


def test_head_projection_matches_reference_fixture(head):
    from hicom_amd.encoder import siglip_head_embed
    m, sd = head
    x = torch.from_numpy(mh.tokens()).to(torch.bfloat16).cuda()
    want = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "golden_head_v1.npz"))["out"])
    with torch.no_grad():
        got32 = siglip_head_embed(x, m, out_dtype=torch.float32)
        got16 = siglip_head_embed(x.view(8, 8, mh.D), m)
    assert float((got32.cpu() - want).abs().max()) <= 1e-3
    assert got16.dtype == torch.bfloat16 and got16.shape == (8, 8, mh.D)
    assert torch.equal(got16.view(-1, mh.D), got32.to(torch.bfloat16))
    assert siglip_head_embed(x, m).requires_grad      # grad mode + trainable head: a graph (stage 3), never a silently detached tensor


def test_head_projection_at_benchmark_size(head):
    """All 64 x 729 tokens (925 GFLOP): 2048 sampled rows against the oracle (the operator is token-wise), and timing."""
    import time
    from hicom_amd.encoder import siglip_head_embed
    m, sd = head
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(64, 729, mh.D, device="cuda", generator=g).to(torch.bfloat16)
    with torch.no_grad():
        out = siglip_head_embed(x, m, out_dtype=torch.float32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            siglip_head_embed(x, m)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    rows = torch.randint(0, 64 * 729, (2048,), generator=torch.Generator().manual_seed(3))
    want = orc.siglip_head_embed(x.view(-1, mh.D)[rows.cuda()].float().cpu(), sd)
    assert float((out.view(-1, mh.D)[rows.cuda()].cpu() - want).abs().max()) <= 1e-3
    flops = 2 * 2 * 64 * 729 * mh.D * mh.INTER
    print(f"\n[head] 64x729 tokens: {dt * 1e3:.3f} ms, {flops / dt / 1e12:.0f} TFLOP/s")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (200, 132, 128), (129, 256, 64), (1000, 1152, 1152), (2300, 640, 192), (4096, 1152, 128)])
def test_dense16_gemm_matches_torch(M, N, K, dt):
    g = torch.Generator().manual_seed(M + N)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    a = (torch.randn(M, K + 64, generator=g) * 0.5).to(tdt).cuda()          # lda > K
    w = (torch.randn(N, K, generator=g) * 0.05).to(tdt).cuda()
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    ref = a[:, :K].double() @ w.double().t() + b.double()
    npad = (N + 127) // 128 * 128
    o16 = torch.full((M, npad), 7.0, dtype=torch.float16, device="cuda")
    y = torch.empty(M, N, device="cuda")
    ssq = torch.zeros((N + 63) // 64, M, device="cuda")
    nv.dense16_gemm(a, w, b, K=K, act=nv.ACT_GELU_TANH, out_f16=o16, n_store=npad, y=y, res=res, ssq=ssq)
    torch.cuda.synchronize()
    act = torch.nn.functional.gelu(ref, approximate="tanh")
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.double() - (act + res.double())).abs().max()) <= 2e-5 * scale
    assert float((o16[:, :N].double() - act).abs().max()) <= 2 ** -11 * scale + 1e-6
    assert float(o16[:, N:].abs().max()) == 0.0 if npad > N else True
    assert float((ssq.sum(0).double() - (ref ** 2).sum(1)).abs().max()) <= 1e-4 * float((ref ** 2).sum(1).max())


@pytest.mark.parametrize("act,approx", [(nv.ACT_GELU_TANH, "tanh"), (nv.ACT_GELU, "none")])
def test_pitched_activation_forward_and_backward(act, approx):
    """hicom_act_rows_fwd / hicom_act_bwd_rows_fwd (the head backward's elementwise steps on the [M, 4304] hidden layer inside rows of
    4544 fp16 elements) against torch's gelu / gelu_backward in fp32."""
    g = torch.Generator(device="cuda").manual_seed(4)
    rows, cols, ld = 777, 4304, 4544
    h = (torch.randn(rows, ld, device="cuda", generator=g) * 2).half()
    da = torch.randn(rows, cols, device="cuda", generator=g).bfloat16()
    a = nv.act_rows(h, cols, act)
    hf = h[:, :cols].float()
    want_a = torch.nn.functional.gelu(hf, approximate=approx)
    assert a.shape == (rows, cols) and float((a.float() - want_a).abs().max()) <= 2 ** -8 * float(want_a.abs().max()) + 2e-3
    d = da.clone()
    nv.act_bwd_rows_(d, h, act)
    want_d = torch.ops.aten.gelu_backward(da.float(), hf, approximate=approx)
    assert float((d.float() - want_d).abs().max()) <= 2 ** -7 * float(want_d.abs().max()) + 2e-3


def test_dense16_pre_activation_output():
    """pre_f16: acc + b BEFORE the activation as a second fp16 output of the same launch (the training forward of the adaptor MLPs
    keeps it for GELU'): equals the activation-free launch bit for bit, and the activated output is unchanged by asking for it."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 1300, 1152, 192
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = (torch.randn(N, generator=g) * 0.1).bfloat16().cuda()
    act0, act1, pre, plain = (torch.empty(M, N, dtype=torch.float16, device="cuda") for _ in range(4))
    nv.dense16_gemm(a, w, b, act=nv.ACT_GELU, out_f16=act0)
    nv.dense16_gemm(a, w, b, act=nv.ACT_GELU, out_f16=act1, pre_f16=pre)
    nv.dense16_gemm(a, w, b, act=nv.ACT_NONE, out_f16=plain)
    torch.cuda.synchronize()
    assert torch.equal(act0, act1) and torch.equal(pre, plain)
    ref = a.double() @ w.double().t() + b.double()
    assert float((pre.double() - ref).abs().max()) <= 2 ** -10 * max(1.0, float(ref.abs().max()))
    with pytest.raises(nv.HicomNativeError):                      # N % 8 != 0: no row epilogue
        nv.dense16_gemm(a, w[:132], b[:132], act=nv.ACT_GELU, out_f16=act1[:, :136], n_store=136, pre_f16=pre[:, :136])


@pytest.mark.parametrize("vec_dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,with_y", [(200, 136, 128, True), (1000, 1152, 192, False), (4099, 1152, 128, True)])
def test_dense16_row_dot_epilogue(M, N, K, with_y, vec_dt):
    """row_dot partials: sum over the 64-column slices = vec . (A W^T + b + res) per row, in fp32 from the accumulators -- with and
    without the packed output beside it (y = NULL: nothing but the [N/64, M] partials is written)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.float16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.float16).cuda()
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    vec = torch.randn(N, generator=g).to(vec_dt).cuda()
    parts = torch.full(((N + 63) // 64, M), float("nan"), device="cuda")
    y = torch.empty(M, N, device="cuda") if with_y else None
    nv.dense16_gemm(a, w, b, y=y, res=res, row_dot=(vec, parts))
    out = torch.empty(M, device="cuda")
    nv.partials_sum(parts, out)
    torch.cuda.synchronize()
    ref = a.double() @ w.double().t() + b.double() + res.double()
    want = ref @ vec.double()
    scale = float((ref.abs() @ vec.double().abs()).max())
    assert float((out.double() - want).abs().max()) <= 2e-6 * scale
    assert float((parts.sum(0).double() - want).abs().max()) <= 2e-6 * scale
    if with_y:
        assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    with pytest.raises(nv.HicomNativeError):
        nv.dense16_gemm(a, w, b, N=N - 4, y=None, res=None, row_dot=(vec, parts))     # N % 8 != 0: no row-contiguous epilogue


def test_head_scores_feed_the_compressor_c2(head):
    """SURVEY §8 row f2 completed (reference encoder.py:277-286 -> projector.py:542-551, use_guide='direct'): the head projection
    emits only the per-token logits guide . frames_embed_n, the compressor streams frames_feature alone.  C2 shape (T=64, 27x27,
    H=896) against the ORACLE CHAIN head (fp32) -> projector_forward (fp32), 1e-3 max-abs; and against the two-tensor path."""
    from types import SimpleNamespace
    import cases
    from gpu_util import build_module, dev_bf16
    from hicom_amd import siglip_head_embed, siglip_head_scores, synth
    m_head, head_sd = head
    T = 64
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": 896, "max_num_frames": T})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="c2h")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    x = synth.synth_inputs(T, 27, 27, 1152, tag="c2h")
    hs, guide = dev_bf16(x["ff"]), dev_bf16(x["g"])
    with torch.no_grad():
        logits = siglip_head_scores(hs, m_head, guide)
        out = m(hs, None, guide, "video", None, local_logits=logits)
        logits2, fe = siglip_head_scores(hs, m_head, guide, return_embed=True)
        out_two = m(hs, fe, guide, "video", None)
        fe_plain = siglip_head_embed(hs, m_head)
    torch.cuda.synchronize()
    assert logits.shape == (T, 27, 27) and logits.dtype == torch.float32 and torch.equal(logits, logits2)
    assert torch.equal(fe, fe_plain)
    fe_ref = orc.siglip_head_embed(hs.float().cpu().view(-1, 1152), head_sd).view(T, 27, 27, 1152)
    s_ref = (fe_ref.double() @ guide.double().cpu()) / 1152 ** 0.5
    assert float((logits.double().cpu() / 1152 ** 0.5 - s_ref).abs().max()) <= 1e-3       # the scaled logits the softmax sees
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    want = orc.projector_forward(cfg, sdt, hs.float().cpu(), fe_ref, guide.float().cpu(), "video", None)
    assert out.shape == want.shape == (T // 4 * 81 + 32, 896)
    assert float((out.float().cpu() - want).abs().max()) <= 1e-3
    assert float((out.float() - out_two.float()).abs().max()) <= 1e-3                      # (fe rounded to bf16 on that path)
    assert torch.equal(out[-32:], out_two[-32:])                                          # the global rows never saw frames_embed


def test_local_logits_argument_checks(head):
    from types import SimpleNamespace
    import cases
    from gpu_util import build_module, dev_bf16
    from hicom_amd import siglip_head_scores
    m_head, _ = head
    case = cases.build_case("G1_direct_T8")
    m = build_module(case)
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    with torch.no_grad():
        ll = siglip_head_scores(ff, m_head, g)
        assert bool(torch.isfinite(m(ff, None, g, "video", None, local_logits=ll)).all())
        with pytest.raises(ValueError):
            m(ff, fe, g, "video", None, local_logits=ll)                     # both given
        with pytest.raises(ValueError):
            m(ff, None, g, "video", None, local_logits=ll[:4])               # shape
        with pytest.raises(ValueError):
            m(ff, None, g, "video", None, local_logits=ll.double())          # dtype
        with pytest.raises(ValueError):
            siglip_head_scores(ff, m_head, g.view(1, -1))
        mc = build_module(cases.build_case("G6_coarse"))
        with pytest.raises(NotImplementedError):
            mc(ff, None, g, "video", None, local_logits=ll)                  # not the release recipe


def test_head_backward_matches_reference_autograd(head):
    """Stage 3 of the reference's script trains "vision_model_head" (train.py:717-720; release scripts :175): gradients of the six
    parameters encoder.py:284-285 touches, for loss = sum(out * R), against the fixture made by the HF module's own fp32 autograd
    (512 samples + sums per parameter).  Tolerance 4e-3 of each parameter's largest gradient entry: the [M, 4304] intermediate
    gradients travel as bf16 between the GEMMs (8 significand bits, as under the reference's own bf16 training; GEMM accumulation
    and every reduction in fp32), and the 64-token fixture averages little of that rounding away (measured: <= 2.3e-3 of the largest
    entry).  The .grad tensors are the bf16 casts."""
    from hicom_amd import encoder
    m, _ = head
    m = m.train()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_head_v1.npz"))
    x = torch.from_numpy(mh.tokens()).to(torch.bfloat16).cuda()
    for p_ in m.parameters():
        p_.grad = None
    with torch.no_grad():
        want_out = encoder.siglip_head_embed(x, m)
    out = encoder.siglip_head_embed(x, m)
    assert out.requires_grad and torch.equal(out.detach(), want_out)            # same kernels, same bits as inference
    R = torch.from_numpy(mh.cotangent()).cuda()
    (out.float() * R).sum().backward()
    fp32 = dict(encoder.LAST_FP32_GRADS)
    names = {"layernorm.weight": m.layernorm.weight, "layernorm.bias": m.layernorm.bias, "mlp.fc1.weight": m.mlp.fc1.weight,
             "mlp.fc1.bias": m.mlp.fc1.bias, "mlp.fc2.weight": m.mlp.fc2.weight, "mlp.fc2.bias": m.mlp.fc2.bias}
    for k, p_ in names.items():
        want = gold[f"grad/head.{k}/samples"]
        s, sabs, mx = gold[f"grad/head.{k}/sums"]
        pos = torch.from_numpy(mh.sample_positions(p_.numel())).cuda()
        got = fp32[k].reshape(-1)[pos].cpu().numpy()
        tol = 4e-3 * mx + 1e-6
        assert np.abs(got - want).max() <= tol, (k, float(np.abs(got - want).max()), tol)
        assert abs(float(fp32[k].double().sum()) - s) <= 4e-3 * sabs + tol * p_.numel() ** 0.5, k
        assert p_.grad is not None and p_.grad.dtype == p_.dtype and p_.grad.shape == p_.shape
        assert np.abs(p_.grad.float().reshape(-1)[pos].cpu().numpy() - want).max() <= 2 ** -7 * mx + tol, k
    xg = x.clone().requires_grad_(True)
    with pytest.raises(NotImplementedError):
        encoder.siglip_head_embed(xg, m).sum().backward()                       # frozen tower body: no d tokens, and never a silent None
    for p_ in m.parameters():
        p_.grad = None
    m.eval()


def test_head_training_forward_reads_the_live_weights_after_a_bypassing_update(head):
    """stage-3 step n-1's head backward re-stamps the fp16 copies of fc1 / fc2 from PRE-step weights; an
    optimizer that writes through `p.data.copy_` bumps no version counter, so step n's head forward must not hit them.  Every
    training forward of the head bumps the weights epoch; inference after training consumes the dirty mark."""
    import copy
    from hicom_amd import encoder
    from hicom_amd import native as nv
    m, _ = head
    m = copy.deepcopy(m).train()
    x = torch.from_numpy(mh.tokens()).to(torch.bfloat16).cuda()
    for step in range(2):
        for p_ in m.parameters():
            p_.grad = None
        out = encoder.siglip_head_embed(x, m)
        nv.note_training_forward()                                       # (the projector's training forward of the same step)
        out.float().square().sum().backward()                            # the head backward rebuilds its copies: pre-step weights
        for lin in (m.mlp.fc1, m.mlp.fc2):
            v = lin.weight._version
            lin.weight.data.copy_((lin.weight.data.float() * 1.5).to(torch.bfloat16))
            assert lin.weight._version == v
        if step == 0:
            got = encoder.siglip_head_embed(x, m).detach()               # next TRAINING forward
        else:
            m.eval()
            with torch.no_grad():
                got = encoder.siglip_head_embed(x, m)                    # inference right after training
        fresh = copy.deepcopy(m).eval()
        fresh.__dict__.pop("_hicom_f16", None)                           # no cached copies: built from the live weights
        with torch.no_grad():
            want = encoder.siglip_head_embed(x, fresh)
        assert torch.equal(got, want), (step, float((got.float() - want.float()).abs().max()))


def test_stage3_chain_head_into_compressor_backward(head):
    """head (trainable) -> frames_embed -> HIComProjector.forward -> loss: the compressor's d frames_embed flows into the head's
    backward; every head parameter and every projector parameter ends up with a finite gradient of its own dtype."""
    import cases
    from gpu_util import build_module, dev_bf16
    from hicom_amd import siglip_head_embed
    m_head, _ = head
    m_head.train()
    case = cases.build_case("G1_direct_T8")
    m = build_module(case).train()
    ff, g = dev_bf16(case.ff), dev_bf16(case.g)
    fe = siglip_head_embed(ff, m_head)                                          # the tower's hidden states double as frames_feature
    assert fe.requires_grad
    out = m(ff, fe, g, case.modal, None)
    out.float().square().sum().backward()
    for p_ in list(m_head.layernorm.parameters()) + list(m_head.mlp.parameters()):
        assert p_.grad is not None and bool(torch.isfinite(p_.grad.float()).all()) and float(p_.grad.float().abs().max()) > 0
        p_.grad = None
    for n, p_ in m.named_parameters():
        if n != "global_compressor.query":
            assert p_.grad is not None and bool(torch.isfinite(p_.grad.float()).all()), n
    m_head.eval()


@pytest.mark.parametrize("Kt,M,N,dt", [(5832, 1152, 1152, torch.bfloat16), (729, 1152, 256, torch.float16), (200, 136, 72, torch.bfloat16),
                                        (46656, 1152, 1152, torch.bfloat16)])
def test_dense16_tn_matches_torch(Kt, M, N, dt):
    """TN form of the dense MFMA GEMM (round 4): C = A^T B over the token axis, the weight gradients dW = dY^T X of the token-stream
    layers.  Ragged token counts (not a multiple of the 64-row stage), widths that are not multiples of the 128 x 128 tile, split
    contraction with partial tiles summed in slice order; bit-stable from launch to launch."""
    g = torch.Generator(device="cuda").manual_seed(Kt + M)
    a = (torch.randn(Kt, M, device="cuda", generator=g) * 0.5).to(dt)
    b = (torch.randn(Kt, N, device="cuda", generator=g) * 0.5).to(dt)
    c = nv.dense16_tn(a, b)
    c2 = nv.dense16_tn(a, b)
    torch.cuda.synchronize()
    assert c.shape == (M, N) and c.dtype == torch.float32 and torch.equal(c, c2)
    if Kt > 10000:                                                   # (fp64 reference of sampled rows: the full product is 124 GFLOP)
        rows = torch.randint(0, M, (64,), device="cuda", generator=g)
        want = a[:, rows].double().t() @ b.double()
        got = c[rows]
    else:
        want, got = a.double().t() @ b.double(), c
    tol = 2e-6 * Kt ** 0.5 * 4 + 1e-5 * float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= tol, (float((got.double() - want).abs().max()), tol)
    # one split and an explicit split count give the same sums up to association
    c1 = nv.dense16_tn(a, b, splits=1)
    torch.cuda.synchronize()
    assert float((c1 - c).abs().max()) <= tol
    if Kt > 10000:
        import time
        for _ in range(3):
            nv.dense16_tn(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            nv.dense16_tn(a, b)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 10
        print(f"\\ndense16_tn {Kt}x{M}x{N}: {dt_s * 1e6:.0f} us = {2.0 * Kt * M * N / dt_s / 1e12:.0f} TFLOP/s (incl. the partial sum)")
