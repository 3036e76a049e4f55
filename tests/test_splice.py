"""the splice of compressed visual tokens into the LLM input embeddings (reference
hicom_arch.py:271-373).  Integer / byte work: every comparison is bit-exact.

CPU part: the oracle restatement against the fixture made by the reference's own method (golden_splice_v1.npz), and the
product's host-side layout plan against the oracle.  GPU part: the HIP row placement + label / mask kernels."""
import os

import numpy as np
import pytest
import torch

import make_golden_splice as mg
from oracle import splice_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden_splice():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_splice_v1.npz"))


@pytest.mark.parametrize("name", list(mg.SPLICE_CASES))
def test_oracle_matches_reference_method(name, golden_splice):
    ids, weight, feats, labels, mask = mg.build(name)
    if mask is not None and labels is None and name == "never":
        pytest.skip("")
    m, e, l = so.splice(weight, ids, mask, labels, feats)
    assert np.array_equal(e.numpy(), golden_splice[name + "/embeds"])
    if labels is not None:
        assert np.array_equal(l.numpy(), golden_splice[name + "/labels"])
    if mask is not None:
        assert np.array_equal(m.numpy().astype(np.int64), golden_splice[name + "/mask"])


@pytest.mark.parametrize("name", list(mg.SPLICE_CASES))
def test_host_plan_matches_oracle(name):
    """plan_layout (pure integer host logic of the product) reproduces the oracle's row sources and lengths."""
    from hicom_amd.splice import plan_layout
    ids, weight, feats, labels, mask = mg.build(name)
    kind, feat, new_len, Lmax = plan_layout(ids.numpy(), [f.shape[0] for f in feats])
    _, e, _ = so.splice(weight, ids, None, None, feats)
    assert e.shape[1] == Lmax
    for b in range(ids.shape[0]):
        for p in range(Lmax):
            if kind[b, p] >= 0:
                want = weight[ids[b, kind[b, p]]]
            elif kind[b, p] == -1:
                want = feats[feat[b, p, 0]][feat[b, p, 1]]
            else:
                want = torch.zeros(weight.shape[1])
            assert torch.equal(e[b, p], want), (b, p)


def test_text_only_and_decode_step_pass_through():
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    ids = torch.tensor([[1, 2, 3]])
    assert prepare_inputs_labels_for_multimodal(None, ids, None, "pkv", None, None) == (ids, None, "pkv", None, None)
    one = torch.tensor([[5]])
    out = prepare_inputs_labels_for_multimodal(None, one, None, None, None, [torch.zeros(1, 4)])
    assert out[0] is one and out[3] is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.SPLICE_CASES))
def test_hip_splice_matches_reference_fixture(name, golden_splice):
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    ids, weight, feats, labels, mask = mg.build(name)
    emb = torch.nn.Embedding(mg.VOCAB, mg.HIDDEN).to(torch.bfloat16).cuda()
    emb.weight.data.copy_(weight)
    d = lambda t: None if t is None else t.cuda()
    with torch.no_grad():
        r_ids, r_mask, r_pkv, r_emb, r_lab = prepare_inputs_labels_for_multimodal(emb, d(ids), d(mask), "pkv", d(labels),
                                                                                  [f.to(torch.bfloat16).cuda() for f in feats])
    torch.cuda.synchronize()
    assert r_ids is None and r_pkv == "pkv"
    assert np.array_equal(r_emb.float().cpu().numpy(), golden_splice[name + "/embeds"])     # values are bf16-representable
    if labels is not None:
        assert r_lab.dtype == torch.int64 and np.array_equal(r_lab.cpu().numpy(), golden_splice[name + "/labels"])
    else:
        assert r_lab is None
    if mask is not None:
        assert r_mask.dtype == mask.dtype and np.array_equal(r_mask.cpu().numpy().astype(np.int64), golden_splice[name + "/mask"])
    else:
        assert r_mask is None


@pytest.mark.gpu
def test_hip_splice_c4_shape_and_errors():
    """BASELINE configs[3] shape: 680 compressed tokens of width 3584 (32 frames, Qwen2.5-7B) into two 2048-token prompts, one
    of them text-only -> ragged batch; bit-exact against the oracle; the reference's failure modes."""
    import time
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    g = torch.Generator().manual_seed(1)
    V, H, S = 4096, 3584, 2048
    weight = torch.randn(V, H, generator=g).to(torch.bfloat16)
    feats = [torch.randn(680, H, generator=g).to(torch.bfloat16), torch.randn(680, H, generator=g).to(torch.bfloat16)]
    ids = torch.randint(0, V, (2, S), generator=g)
    ids[0, 17] = -201
    labels = torch.where(ids >= 0, ids, torch.full_like(ids, -100))
    mask = torch.ones(2, S, dtype=torch.long)
    mask[1, -5:] = 0
    wm, we, wl = so.splice(weight.float(), ids, mask, labels, [f.float() for f in feats])
    emb = torch.nn.Embedding(V, H).to(torch.bfloat16).cuda()
    emb.weight.data.copy_(weight)
    args = (emb, ids.cuda(), mask.cuda(), None, labels.cuda(), [f.cuda() for f in feats])
    _, m, _, e, l = prepare_inputs_labels_for_multimodal(*args)
    torch.cuda.synchronize()
    assert e.requires_grad                       # trainable embedding table, autograd on: the splice is part of the graph
    e = e.detach()
    assert e.shape == (2, S - 1 + 680, H)
    assert torch.equal(e.float().cpu(), we) and torch.equal(l.cpu(), wl) and torch.equal(m.cpu(), wm)
    with torch.no_grad():
        for _ in range(3):
            prepare_inputs_labels_for_multimodal(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            prepare_inputs_labels_for_multimodal(*args)
        torch.cuda.synchronize()
    print(f"\n[splice] C4 shape: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call ({e.numel() * 2 / 1e6:.1f} MB of rows)")
    with pytest.raises(UnboundLocalError):                      # ragged batch with a mask but no labels (reference :352)
        prepare_inputs_labels_for_multimodal(emb, ids.cuda(), mask.cuda(), None, None, [f.cuda() for f in feats])
    with pytest.raises(IndexError):                             # more placeholders than feature tensors
        prepare_inputs_labels_for_multimodal(emb, ids.cuda(), mask.cuda(), None, labels.cuda(), [feats[0].cuda()])
    from hicom_amd import native as nv
    with pytest.raises(nv.HicomNativeError):                    # no CPU path
        prepare_inputs_labels_for_multimodal(torch.nn.Embedding(V, 8), ids, None, None, None, [torch.zeros(3, 8)])


def test_host_plan_zero_row_feature_and_positions():
    """A feature with zero rows shortens its sample (the reference cats an empty tensor, :314); feat_at records where every
    placed feature starts (the backward's gather ranges)."""
    from hicom_amd.splice import plan_layout
    ids = np.array([[5, -201, 7, -200], [1, 2, 3, 4], [-202, 9, 9, 9]], dtype=np.int64)
    plan = plan_layout(ids, [0, 3, 2, 4])          # sample 1 is text-only: consumes feature 2 without placing it
    kind, feat, new_len, Lmax = plan
    assert new_len.tolist() == [2 + 0 + 3, 4, 3 + 4] and Lmax == 7
    assert kind[0].tolist() == [0, 2, -1, -1, -1, -2, -2]
    assert feat[0, 2:5].tolist() == [[1, 0], [1, 1], [1, 2]]
    assert kind[2].tolist() == [-1, -1, -1, -1, 1, 2, 3] and feat[2, :4, 0].tolist() == [3] * 4
    assert plan.feat_at.tolist() == [[0, 1, 0], [0, 2, 3], [0, 0, 0], [2, 0, 4]]
    with pytest.raises(IndexError):
        plan_layout(ids, [0, 3, 2])


@pytest.mark.gpu
def test_hip_splice_gradients_match_reference_autograd():
    """the splice sits inside every training forward (hicom_arch.py:313-330: embed_tokens + torch.cat), so
    gradients must reach the compressed tokens and the embedding table.  Expected values: the oracle restatement (plain
    indexing + cat) under torch autograd on the CPU, fp32."""
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    g = torch.Generator().manual_seed(3)
    V, H, S = 64, 32, 12
    weight = torch.randn(V, H, generator=g)
    feats = [torch.randn(5, H, generator=g), torch.randn(0, H, generator=g), torch.randn(3, H, generator=g), torch.randn(4, H, generator=g)]
    ids = torch.randint(0, V, (3, S), generator=g)
    ids[0, 2], ids[0, 9] = -201, -200           # two placeholders (the second takes the zero-row feature)
    ids[2, 0] = -202                            # sample 1 stays text-only and consumes feature 2
    ids[0, 4] = ids[0, 5]                       # a repeated token: two rows add into one embedding row
    w_ref = weight.clone().requires_grad_(True)
    f_ref = [f.clone().requires_grad_(True) for f in feats]
    _, e_ref, _ = so.splice(w_ref, ids, None, None, f_ref)
    d_out = torch.randn(e_ref.shape, generator=g)
    e_ref.backward(d_out)
    w = weight.clone().cuda().requires_grad_(True)
    f = [t.clone().cuda().requires_grad_(True) for t in feats]
    _, _, _, e, _ = prepare_inputs_labels_for_multimodal(w, ids.cuda(), None, None, None, f)
    assert e.requires_grad and torch.equal(e.detach().cpu(), e_ref.detach())
    e.backward(d_out.cuda())
    assert torch.allclose(w.grad.cpu(), w_ref.grad, atol=1e-6)
    for k in (0, 2, 3):
        want = f_ref[k].grad if f_ref[k].grad is not None else torch.zeros_like(feats[k])
        assert torch.allclose(f[k].grad.cpu(), want, atol=0), k
    # no graph when nothing requires grad / under no_grad
    with torch.no_grad():
        assert not prepare_inputs_labels_for_multimodal(w, ids.cuda(), None, None, None, f)[3].requires_grad


@pytest.mark.gpu
def test_hip_splice_input_validation():
    """3-D features, views at odd storage offsets, zero-row features against a mask."""
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    V, H, S = 32, 24, 6                           # 24 bf16 = 48-byte rows: every row 16-byte aligned
    emb = torch.nn.Embedding(V, H).to(torch.bfloat16).cuda()
    ids = torch.tensor([[1, -201, 2, 3, 4, 5]]).cuda()
    with pytest.raises(ValueError):
        prepare_inputs_labels_for_multimodal(emb, ids, None, None, None, [torch.zeros(2, 3, H, dtype=torch.bfloat16).cuda()])
    base = torch.randn(4 * H + 4).to(torch.bfloat16).cuda()
    view = base[4:4 + 3 * H].view(3, H)           # storage offset 8 bytes: not 16-byte aligned
    assert view.data_ptr() % 16 != 0
    out = prepare_inputs_labels_for_multimodal(emb, ids, None, None, None, [view])[3]
    torch.cuda.synchronize()
    assert torch.equal(out[0, 1:4], view)
    mask = torch.ones(1, S, dtype=torch.long).cuda()
    with pytest.raises(RuntimeError, match="negative dimension"):
        prepare_inputs_labels_for_multimodal(emb, ids, mask, None, ids.clamp(min=0), [torch.zeros(0, H, dtype=torch.bfloat16).cuda()])


@pytest.mark.gpu
def test_hip_splice_random_sweep_against_the_oracle():
    """Seeded sweep of `prepare_inputs_labels_for_multimodal` (hicom_arch.py:283-372): batch sizes, prompt lengths, 0..3 placeholders per
    sample of all three modal tokens (also adjacent, first and last position), feature row counts incl. zero rows, text-only samples
    between multimodal ones, with / without labels and mask -- bit-exact embeddings, labels and mask against the splice oracle."""
    import random
    from hicom_amd.splice import prepare_inputs_labels_for_multimodal
    rng = random.Random(31337)
    V, H = 300, 64
    gw = torch.Generator().manual_seed(2)
    weight = torch.randn(V, H, generator=gw).to(torch.bfloat16)
    emb = torch.nn.Embedding(V, H).to(torch.bfloat16).cuda()
    emb.weight.data.copy_(weight)
    emb.weight.requires_grad_(False)
    ran = 0
    for k in range(40):
        B, S = rng.choice([1, 2, 3, 5]), rng.choice([4, 9, 33, 128])
        ids = torch.randint(0, V, (B, S), generator=gw)
        feats = []
        for b in range(B):
            nph = rng.choice([0, 1, 1, 2, 3])
            if nph == 0:
                feats.append(torch.randn(rng.choice([0, 3]), H, generator=gw).to(torch.bfloat16))       # a text-only sample still consumes one entry (:290-299)
                continue
            pos = sorted(rng.sample(range(S), min(nph, S)))
            if rng.random() < 0.3:
                pos[0] = 0
            if rng.random() < 0.3:
                pos[-1] = S - 1
            for p_ in sorted(set(pos)):
                ids[b, p_] = rng.choice([-200, -201, -202])
                feats.append(torch.randn(rng.choice([0, 1, 7, 40]), H, generator=gw).to(torch.bfloat16))
        with_labels = rng.random() < 0.7
        with_mask = with_labels and rng.random() < 0.7            # (a mask without labels on a ragged batch is the reference's UnboundLocalError)
        labels = torch.where(ids >= 0, ids, torch.full_like(ids, -100)) if with_labels else None
        mask = None
        if with_mask:
            mask = torch.ones(B, S, dtype=torch.long)
            for b in range(B):
                mask[b, S - rng.randrange(0, min(S, 4)):] = 0
        try:
            wm, we, wl = so.splice(weight.float(), ids, mask, labels, [f.float() for f in feats])
        except RuntimeError:
            continue          # (the reference cannot build the mask of a sample that SHRANK -- a placeholder with zero feature rows, :354-363)
        ran += 1
        d = lambda t: None if t is None else t.cuda()
        with torch.no_grad():
            _, m, _, e, l = prepare_inputs_labels_for_multimodal(emb, d(ids), d(mask), None, d(labels), [f.cuda() for f in feats])
        torch.cuda.synchronize()
        what = (k, B, S, with_labels, with_mask)
        assert tuple(e.shape) == tuple(we.shape) and torch.equal(e.float().cpu(), we), what
        assert (l is None) == (wl is None) and (l is None or torch.equal(l.cpu(), wl)), what
        assert (m is None) == (wm is None) and (m is None or torch.equal(m.cpu().to(wm.dtype), wm)), what
    assert ran >= 30, ran
