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
    with pytest.raises(RuntimeError):
        siglip_head_embed(x, m)                       # grad mode + trainable head: refuse, never detach silently


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
