"""Reference-signature surfaces and robustness contracts that round 2 left without a test (VERDICT r2 "Missing" #6,
"Next round" #1):

  * MultiheadAttention.forward (reference projector.py:166-228): the <= 64-key branch and the streamed long-key branch;
  * GuideInjector.forward (reference :344-397): direct / coarse / fine on 2-D and 4-D visual_embed;
  * the chained C4 segment  siglip_head_embed -> HIComProjector.forward (T=32, H=3584) -> prepare_inputs_labels_for_multimodal
    against the oracle chain (reference encoder.py:284-286 -> projector.py:676-708 -> hicom_arch.py:271-373);
  * plan key / weight-cache / sharded-path guards.

Tolerance: 1e-3 max-abs on fp32 results against the fp32 oracle (BASELINE.json north_star); labels / masks / row placement bit-exact.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cases
from gpu_util import build_module, dev_bf16
from hicom_amd import native as nv
from hicom_amd import synth
from oracle import hicom_oracle as orc
from oracle import splice_oracle as so

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module_and_sd(name):
    case = cases.build_case(name)
    return build_module(case), {k: torch.from_numpy(v) for k, v in case.sd.items()}, case


def _bf16_cuda(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


# ---------------------------------------------------------------------------------------------------------------------
# MultiheadAttention.forward
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kv_len", [64, 7])
def test_mha_forward_small_key_branch(kv_len):
    """[B, q, E] x [B, kv <= 64, E] (the text tokens of "fine" injection): projected states + hicom_small_mha_fwd; key and
    value are different tensors here (the small-key path projects them separately)."""
    m, sd, _ = _module_and_sd("G7_fine")
    att = m.local_compressor.guide_injector.fine_proj
    att.return_fp32 = True
    B, q_len, E = 2, 5, 1152
    q, k, v = _bf16_cuda((B, q_len, E), 1), _bf16_cuda((B, kv_len, E), 2), _bf16_cuda((B, kv_len, E), 3)
    with torch.no_grad():
        out, weights = att(q, k, v)
    torch.cuda.synchronize()
    assert weights is None and out.shape == (B, q_len, E) and out.dtype == torch.float32
    for b in range(B):
        want = orc.mha(q[b].float().cpu(), k[b].float().cpu(), v[b].float().cpu(), sd, "local_compressor.guide_injector.fine_proj", 9)
        assert float((out[b].cpu() - want).abs().max()) <= TOL, b
    # clip-scale on the short-key branch (ref :184-191): L2-normalised projected states, * exp(logit_scale) + bias
    ls, lb = torch.tensor(1.2), torch.tensor(-3.0)
    with torch.no_grad():
        out_c, _ = att(q, k, v, logit_scale=ls.cuda(), logit_bias=lb.cuda())
    for b in range(B):
        want = orc.mha(q[b].float().cpu(), k[b].float().cpu(), v[b].float().cpu(), sd, "local_compressor.guide_injector.fine_proj", 9, ls, lb)
        assert float((out_c[b].cpu() - want).abs().max()) <= TOL, b
    att.return_fp32 = False
    with torch.no_grad():
        assert att(q, k, v)[0].dtype == torch.bfloat16                      # module dtype, as the reference returns it
    with pytest.raises(ValueError):
        att(q[0], k, v)                                                      # "Batch x Time x Channel" (ref :170-172)
    with pytest.raises(NotImplementedError):
        att(q, k, v, attention_mask=torch.zeros(B, 1, q_len, kv_len).cuda())


@pytest.mark.parametrize("q_len,kv_len", [(1, 300), (4, 1000), (32, 333)])
def test_mha_forward_long_key_stream_branch(q_len, kv_len):
    """key is value (bf16), more than 64 keys: k_proj / v_proj folded around the raw tokens, one streaming pass without
    positional terms (H, W = 1, N), combine, v_proj per head, out_proj -- against the reference's un-folded form."""
    m, sd, _ = _module_and_sd("G2_off_T8")
    att = m.global_compressor.attn_layer
    att.return_fp32 = True
    E = 1152
    q, x = _bf16_cuda((1, q_len, E), 5, 0.5), _bf16_cuda((1, kv_len, E), 6)
    with torch.no_grad():
        out, _ = att(q, x, x)
    torch.cuda.synchronize()
    want = orc.mha(q[0].float().cpu(), x[0].float().cpu(), x[0].float().cpu(), sd, "global_compressor.attn_layer", 9)
    assert out.shape == (1, q_len, E)
    assert float((out[0].cpu() - want).abs().max()) <= TOL
    # clip-scale through the reference signature (ref :184-191): queries and PROJECTED keys L2-normalised, * exp(logit_scale) + bias
    ls, lb = torch.tensor(1.5), torch.tensor(-2.0)
    with torch.no_grad():
        out_c, _ = att(q, x, x, logit_scale=ls.cuda(), logit_bias=lb.cuda())
    torch.cuda.synchronize()
    want_c = orc.mha(q[0].float().cpu(), x[0].float().cpu(), x[0].float().cpu(), sd, "global_compressor.attn_layer", 9, ls, lb)
    assert float((out_c[0].cpu() - want_c).abs().max()) <= TOL
    att.return_fp32 = False
    with pytest.raises(NotImplementedError):
        att(q, x, x.clone())                                                 # long keys: key must BE value (folded projections)


# ---------------------------------------------------------------------------------------------------------------------
# GuideInjector.forward
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,mode", [("G1_direct_T8", "direct"), ("G6_coarse", "coarse"), ("G7_fine", "fine")])
@pytest.mark.parametrize("shape", [(10, 1152), (2, 3, 3, 1152)])
@pytest.mark.parametrize("vis_dtype", [torch.float32, torch.bfloat16])
def test_guide_injector_forward(name, mode, shape, vis_dtype):
    m, sd, case = _module_and_sd(name)
    inj = m.local_compressor.guide_injector
    assert inj.use_guide == mode
    g = torch.Generator().manual_seed(7)
    vis = torch.randn(*shape, generator=g).to(torch.bfloat16)              # bf16-representable values in either dtype
    guide = torch.from_numpy(case.g).to(torch.bfloat16)
    with torch.no_grad():
        got = inj(vis.to(vis_dtype).cuda(), guide.cuda())
    torch.cuda.synchronize()
    want = orc.guide_inject(mode, vis.float(), guide.float(), sd, "local_compressor.guide_injector", False)
    assert got.shape == want.shape
    if mode == "direct":
        # the guide broadcast to the visual shape (ref :352-368) as a stride-0 view: nothing is materialised
        assert torch.equal(got.float().cpu(), want) and got.stride()[0] == 0
    else:
        assert got.dtype == torch.float32 and float((got.cpu() - want).abs().max()) <= TOL
    with pytest.raises(ValueError):
        inj(vis.view(1, *shape).cuda(), guide.cuda())                        # "Invalid input shape for guide embedding." (:350)


# ---------------------------------------------------------------------------------------------------------------------
# chained C4 segment (BASELINE configs[3] without the tower body and the LLM)
# ---------------------------------------------------------------------------------------------------------------------
def test_c4_chain_head_compressor_splice_matches_oracle_chain():
    import make_golden_head as mh
    from hicom_amd import prepare_inputs_labels_for_multimodal, siglip_head_embed, synth
    from test_gpu_head import _Head
    T, HID, S, V = 32, 3584, 2048, 4096
    head_sd = mh.head_state_dict()
    head = _Head(head_sd).to(torch.bfloat16).cuda().eval()
    head_sd_t = {k: torch.from_numpy(v) for k, v in head_sd.items()}
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": HID, "max_num_frames": T})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="c4")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    x = synth.synth_inputs(T, 27, 27, 1152, tag="c4")
    hidden_states, guide = dev_bf16(x["ff"]), dev_bf16(x["g"])           # tower output = frames_feature (encoder.py:277-283)
    gen = torch.Generator().manual_seed(9)
    weight = torch.randn(V, HID, generator=gen).to(torch.bfloat16)
    ids = torch.randint(0, V, (2, S), generator=gen)
    ids[0, 41] = -201                                                    # <video> in sample 0; sample 1 is text-only
    labels = torch.where(ids >= 0, ids, torch.full_like(ids, -100))
    mask = torch.ones(2, S, dtype=torch.long)
    mask[1, -9:] = 0
    emb = torch.nn.Embedding(V, HID).to(torch.bfloat16).cuda()
    emb.weight.data.copy_(weight)
    with torch.no_grad():
        fe = siglip_head_embed(hidden_states, head)                        # bf16, as the tower hands it on
        out32 = m(hidden_states, fe, guide, "video", None)                 # fp32 (return_fp32): the compressed tokens
        feats = [out32.to(torch.bfloat16), torch.zeros(3, HID, dtype=torch.bfloat16, device="cuda")]
        _, new_mask, _, embeds, new_labels = prepare_inputs_labels_for_multimodal(emb, ids.cuda(), mask.cuda(), None, labels.cuda(), feats)
    torch.cuda.synchronize()
    # oracle chain: head (fp32) -> bf16 interface (the model dtype between tower and projector) -> compressor (fp32)
    fe_ref = orc.siglip_head_embed(hidden_states.float().cpu().view(-1, 1152), head_sd_t).view(T, 27, 27, 1152)
    assert float((fe.float().cpu() - fe_ref).abs().max()) <= 2 ** -8 * float(fe_ref.abs().max())   # bf16 cast of a <= 1e-3 result
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    want = orc.projector_forward(cfg, sdt, hidden_states.float().cpu(), fe_ref.to(torch.bfloat16).float(), guide.float().cpu(), "video", None)
    assert out32.shape == want.shape == (T // 4 * 81 + 32, HID)
    assert float((out32.cpu() - want).abs().max()) <= TOL
    # splice of the product's own bf16 tokens: every row, label and mask element bit-exact against the oracle placement
    wm, we, wl = so.splice(weight.float(), ids, mask, labels, [f.float().cpu() for f in feats])
    assert embeds.shape == (2, S - 1 + out32.shape[0], HID)
    assert torch.equal(embeds.float().cpu(), we) and torch.equal(new_labels.cpu(), wl) and torch.equal(new_mask.cpu(), wm)


# ---------------------------------------------------------------------------------------------------------------------
# plan key, weight caches, sharded-path guards
# ---------------------------------------------------------------------------------------------------------------------
def test_plan_key_includes_frames_embed_shape():
    """a second call with the same frames_feature shape and a differently shaped frames_embed must not hit the
    cached plan (it would patch in a pointer the kernel reads out of bounds): it raises like the first call would."""
    m, _, case = _module_and_sd("G1_direct_T8")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    with torch.no_grad():
        m(ff, fe, g, "video", None)
        with pytest.raises(ValueError):
            m(ff, fe[:4].contiguous(), g, "video", None)
        with pytest.raises(ValueError):
            m(ff, fe[:, :3].contiguous(), g, "video", None)
        assert bool(torch.isfinite(m(ff, fe, g, "video", None)).all())


def test_weight_writes_that_bypass_the_version_counter():
    """DeepSpeed's bf16 optimizer updates parameters with `p.data.copy_(...)` / through a flat alias, which
    bumps no version counter.  Training-mode forwards rebuild every weight-derived cache (fp16 readout copies, kpe, plans)
    from the live weights; at inference the parameters' `.data` hook does (round 5), or `hicom_amd.invalidate_weight_caches()`."""
    import hicom_amd
    m, _, case = _module_and_sd("G1_direct_T8")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    w = m.local_compressor.readout[0].weight
    wk = m.global_compressor.attn_layer.k_proj.weight
    with torch.no_grad():
        base = m(ff, fe, g, "video", None).clone()
    v0 = (w._version, wk._version)
    w.data.copy_(w.data * 1.5)                                           # no version bump, same storage
    wk.data.copy_(wk.data * -1.0)
    assert (w._version, wk._version) == v0
    # training path (autograd on, trainable parameters): sees the live weights at once
    trained = m(ff, fe, g, "video", None).detach().clone()
    assert float((trained[:-32] - base[:-32]).abs().max()) > 1e-3           # local rows: readout weight changed
    assert float((trained[-32:] - base[-32:]).abs().max()) > 1e-4           # global rows: k_proj changed (folded queries, kpe)
    with torch.no_grad():
        # the first inference forward after training rebuilds once and agrees with the training forward
        assert float((m(ff, fe, g, "video", None) - trained).abs().max()) <= 2e-4
        # round 5 (verdict r4 #9): an eval-mode `p.data.copy_` is seen WITHOUT a call to hicom_amd.invalidate_weight_caches() --
        # the projector's parameters report `.data` accesses (native.TrackedParameter) and the next forward refreshes the tables
        w.data.copy_(w.data / 1.5)
        wk.data.copy_(wk.data * -1.0)
        assert float((m(ff, fe, g, "video", None) - base).abs().max()) <= 2e-4
        # `p.data = new tensor` and an in-place op on the alias likewise; the explicit call stays available (writes through an alias
        # of the storage taken earlier, which no parameter object sees)
        w.data = (w.detach().float() * 2).to(w.dtype)
        doubled = m(ff, fe, g, "video", None).clone()
        assert float((doubled[:-32] - base[:-32]).abs().max()) > 1e-3
        alias = w.detach()
        alias.copy_((alias.float() / 2).to(alias.dtype))                     # bumps the shared version counter: seen as well
        assert float((m(ff, fe, g, "video", None) - base).abs().max()) <= 2e-4
        hicom_amd.invalidate_weight_caches()
        assert float((m(ff, fe, g, "video", None) - base).abs().max()) <= 2e-4


def test_data_reads_do_not_refresh_the_plan_tables():
    """Round-5 verdict #8: `p.data.norm()` in a serving / logging loop must not cost a rebuild of the weight-derived tables (a plan
    miss is ~2x the step): the plan's refresh producer runs for WRITES through `.data` only."""
    m, _, case = _module_and_sd("G1_direct_T8")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    with torch.no_grad():
        base = m(ff, fe, g, "video", None).clone()
        plan = next(iter(m.__dict__["_engine_plans"].values()))
        calls = []
        inner = plan.refresh
        plan.refresh = lambda: (calls.append(1), inner())[1]
        for _ in range(5):
            for p in m.parameters():
                p.data.norm()                                                 # a gradient- / weight-norm logger, an EMA read
            assert torch.equal(m(ff, fe, g, "video", None), base)
        assert not calls and next(iter(m.__dict__["_engine_plans"].values())) is plan
        w = m.local_compressor.readout[0].weight
        w.data.mul_(1.5)                                                      # a write through the alias: seen
        assert float((m(ff, fe, g, "video", None)[:-32] - base[:-32]).abs().max()) > 1e-3 and len(calls) == 1


def test_eval_after_a_training_step_whose_optimizer_bypasses_the_version_counter():
    """the tables built during the LAST training forward / backward carry the current epoch but pre-date the
    optimizer step.  A version-bypassing step (`p.data.copy_`, DeepSpeed's flat alias) followed by an inference forward
    (Trainer.evaluate) must rebuild them: the training forward leaves a dirty mark that the next inference entry point consumes."""
    m, _, case = _module_and_sd("G1_direct_T8")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    names = ("local_compressor.readout.0.weight", "local_compressor.readout.2.weight", "global_compressor.attn_layer.k_proj.weight",
             "global_compressor.attn_layer.out_proj.weight", "global_compressor.readout.0.weight")
    params = dict(m.named_parameters())
    m.train()
    out = m(ff, fe, g, "video", None)                                    # training forward: epoch bump, tables at epoch E
    out.float().square().sum().backward()                                # backward may rebuild tables (still pre-step weights)
    v0 = tuple(params[n]._version for n in names)
    for n in names:                                                      # "optimizer step" behind torch's back
        params[n].data.copy_((params[n].data.float() * 1.25).to(torch.bfloat16))
    assert tuple(params[n]._version for n in names) == v0
    m.eval()
    with torch.no_grad():
        got = m(ff, fe, g, "video", None)
        fresh = build_module(cases.build_case("G1_direct_T8"))
        fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        want = fresh(ff, fe, g, "video", None)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
        assert torch.equal(m(ff, fe, g, "video", None), want)            # (and the mark is consumed once: later forwards hit)


def test_f16_weight_copy_range_check():
    from hicom_amd import native as nv
    w = torch.full((4, 64), 7.0e4, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError):
        nv.f16_weight_copy(w)
    ok = torch.full((4, 64), 3.0e-6, dtype=torch.bfloat16, device="cuda")   # an fp16 subnormal: kept to 2^-25 absolute
    c = nv.f16_weight_copy(ok)
    torch.cuda.synchronize()
    assert float((c.float() - ok.float()).abs().max()) <= 2.0 ** -25


def test_sharded_forward_takes_the_recipes_outside_the_executor_operator_by_operator():
    """Round 2 (ADVICE, medium) made coarse / fine / query-side-adaptor recipes RAISE in sharded_forward: the shard plans only patch guide
    aliases, and those recipes derive their queries from the guide per call.  Round 5 (verdict r4 #6): they shard operator by
    operator (`dist.sharded_forward_stepwise`: queries recomputed on every call, nothing cached that could go stale) -- through a
    1-rank RCCL group the result equals the unsharded forward, also when the guide changes between two calls; round 6: a clip-scale GLOBAL
    stage shards the same way; unset clip logits still refuse."""
    import socket
    import torch.distributed as dist
    from hicom_amd.dist import sharded_forward
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for name in ("G6_coarse", "G7_fine", "G5b_adaptqkvg_off"):
            m, _, case = _module_and_sd(name)
            ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
            with torch.no_grad():
                for gg in (g, (g.float() * -0.5).to(g.dtype)):                 # a second call with another guide: no stale query
                    want = m(ff, fe, gg, "video", None)
                    got = sharded_forward(m, ff, fe, gg, ff.shape[0])
                    torch.cuda.synchronize()
                    assert got.shape == want.shape and float((got.float() - want.float()).abs().max()) <= 2e-4, name
                out, ev = sharded_forward(m, ff, fe, g, ff.shape[0], deferred=True)
                ev.synchronize()
                assert float((out.float() - m(ff, fe, g, "video", None).float()).abs().max()) <= 2e-4
        # round 6: a clip-scale GLOBAL stage shards as well (operator by operator: the key norms are per token)
        m, _, case = _module_and_sd("G1_direct_T8")
        m.global_use_clip_scale = True
        m.set_clip_logits(glob=(1.5, -2.0))
        ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
        with torch.no_grad():
            want = m(ff, fe, g, "video", None)
            got = sharded_forward(m, ff, fe, g, 8)
        assert float((got.float() - want.float()).abs().max()) <= 2e-4
    finally:
        dist.destroy_process_group()
    m, _, case = _module_and_sd("G1_direct_T8")
    m.config.use_clip_scale = "local"
    m.local_use_clip_scale = True
    with torch.no_grad(), pytest.raises(RuntimeError):
        sharded_forward(m, dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), 8)


def test_inplace_weight_update_refreshes_derived_tables_under_the_same_plan():
    """An optimizer step changes weight CONTENT, not addresses: the executor plan survives, and the weight-derived device tables it
    points at (fp16 readout copies, kpe = W_k . PE^T, C = G0 . W_o) are re-run in place -- the next forward equals the forward of a
    freshly built module holding the updated weights, for an in-place update (version counter) and for a `p.data.copy_` write
    followed by invalidate_weight_caches() (no counter)."""
    import hicom_amd
    m, _, case = _module_and_sd("G1_direct_T8")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    names = ("local_compressor.readout.0.weight", "global_compressor.attn_layer.k_proj.weight",
             "global_compressor.attn_layer.out_proj.weight", "global_compressor.readout.0.weight")
    params = dict(m.named_parameters())
    with torch.no_grad():
        m(ff, fe, g, "video", None)
        plan = next(iter(m.__dict__["_engine_plans"].values()))
        for step, bypass in enumerate((False, True)):
            for n in names:
                new = (params[n].float() * (1.0 + 0.25 * (step + 1))).to(torch.bfloat16)
                if bypass:
                    params[n].data.copy_(new)                                  # no version bump (DeepSpeed's flat alias writes like this)
                else:
                    params[n].copy_(new)
            if bypass:
                hicom_amd.invalidate_weight_caches()
            got = m(ff, fe, g, "video", None)
            assert next(iter(m.__dict__["_engine_plans"].values())) is plan    # same plan, same workspace, same table addresses
            case2 = cases.build_case("G1_direct_T8")
            fresh = build_module(case2)
            fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
            want = fresh(ff, fe, g, "video", None)
            assert torch.equal(got, want), (step, float((got.float() - want.float()).abs().max()))


def test_guide_off_reuses_folded_learnable_queries_until_the_weights_change():
    """Guide off: the injected queries are the learnable `query` parameter (IdentityMap, reference projector.py:586-587), so q_proj +
    fold are weight-only work; the executor keeps them in the plan's workspace (hicom_compressor_args.reuse_queries) and redoes them
    after a weight update.  Second call bit-identical to the first; after in-place updates of `query`, `q_proj` and `k_proj` the
    result equals a freshly built module's."""
    m, _, case = _module_and_sd("G2_off_T8")
    ff, fe = dev_bf16(case.ff), dev_bf16(case.fe)
    params = dict(m.named_parameters())
    with torch.no_grad():
        first = m(ff, fe, None, "video", None).clone()
        plan = next(iter(m.__dict__["_engine_plans"].values()))
        assert plan.args.reuse_queries == 1
        assert torch.equal(m(ff, fe, None, "video", None), first)
        for n in ("global_compressor.query", "global_compressor.attn_layer.q_proj.weight", "global_compressor.attn_layer.k_proj.weight"):
            params[n].copy_((params[n].float() * 1.5).to(torch.bfloat16))
        got = m(ff, fe, None, "video", None)
        assert next(iter(m.__dict__["_engine_plans"].values())) is plan and plan.args.reuse_queries == 1
        fresh = build_module(cases.build_case("G2_off_T8"))
        fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        want = fresh(ff, fe, None, "video", None)
        assert torch.equal(got, want) and not torch.equal(got, first)


@pytest.mark.parametrize("name", ["G6_coarse", "G7_fine", "G5_adaptkv"])
def test_two_stream_stepwise_forward_is_bit_stable_and_equals_one_stream(name):
    """The operator-by-operator path runs the local chain on a side stream beside the global chain (round 3).  Same operators, same
    bits as the one-stream form, and 200 back-to-back forwards on two alternating input sets -- fresh output tensors every call, no
    host synchronisation -- reproduce the first results exactly (a missing fence between the streams shows up here)."""
    m, _, case = _module_and_sd(name)
    sets = [(dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g))]
    gen = torch.Generator(device="cuda").manual_seed(23)
    sets.append(tuple(torch.randn(t.shape, device="cuda", generator=gen).to(torch.bfloat16) for t in sets[0]))
    with torch.no_grad():
        m.overlap_stages = False
        want = [m(a, b, c, case.modal, None).clone() for a, b, c in sets]
        m.overlap_stages = True
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        for i in range(200):
            a, b, c = sets[i & 1]
            bad += (m(a, b, c, case.modal, None) != want[i & 1]).any()
    assert int(bad) == 0


def test_bench_distributed_branch_world1_prints_one_json_line():
    """VERDICT r3 #7a: bench.py's N > 1 branch (process group, sharded_forward, the deferred serving loop, the JSON relay through the
    duplicated stdout) run as a fresh process at world size 1 (HICOM_BENCH_FORCE_DIST=1) -- the driver's first real multi-GPU run
    must not be the first time this code executes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HICOM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                  # ONE line on stdout, nothing else (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "frame-shard" in d["config"]["parallelism"] and "RCCL" in d["config"]["parallelism"]
    assert d["ms_per_step_joined"]["median"] > 0 and d["ms_per_step"] > 0
    assert d["roofline"]["frac"] > 0


@pytest.mark.parametrize("name", ["G1_direct_T8", "G11_c1_shape", "G2_off_T8", "G9_anyres"])
def test_fp16_projector_and_inputs_match_the_oracle(name):
    """Round-5 verdict missing #4: the reference's own inference default is fp16 (`--dtype float16`, inference_video_mcqa_videomme.py:323;
    `load_mm_projector` casts to fp16, projector.py:53).  An fp16 projector with fp16 inputs runs on the bf16 kernels through
    hicom_cast16_fwd (HIComProjector._forward_half): on bf16-representable weights and inputs -- what a bf16-trained checkpoint loaded as
    fp16 holds -- the casts are exact and the result is the bf16 module's, <= 1e-3 from the oracle; it follows in-place weight updates;
    fp32 modules keep refusing."""
    from oracle_util import run_oracle
    case = cases.build_case(name)
    m = build_module(case, fp32_out=True).half()
    assert next(m.parameters()).dtype == torch.float16
    h = lambda a: None if a is None else dev_bf16(a).half()
    ff, fe, g, nl = h(case.ff), h(case.fe), h(case.g), h(case.newline)
    if case.anyres is not None:
        a = case.anyres
        ff = {"base": None if a["no_base"] else ff[0], "patch": h(a["patch_ff"])}
        fe = {"base": None if a["no_base"] else fe[0], "patch": h(a["patch_fe"])}
    with torch.no_grad():
        out = m(ff, fe, g, case.modal, nl)
        want = run_oracle(case)["out"].numpy()
        assert float(np.abs(out.float().cpu().numpy() - want).max()) <= TOL
        m.return_fp32 = False
        out16 = m(ff, fe, g, case.modal, nl)
        assert out16.dtype == torch.float16 and float(np.abs(out16.float().cpu().numpy() - want).max()) <= TOL + 2.0 ** -9 * float(np.abs(want).max())
        m.return_fp32 = True
        w = (m.local_compressor or m.global_compressor).readout[2].weight
        w0 = w.detach().clone()
        w.mul_(0.5)                                                          # an in-place update of the fp16 module reaches the bf16 twin
        out2 = m(ff, fe, g, case.modal, nl)
        assert float((out2 - out).abs().max()) > 1e-4
        w.copy_(w0)
        assert torch.equal(m(ff, fe, g, case.modal, nl), out)
    with pytest.raises(NotImplementedError):
        build_module(case).float()(dev_bf16(case.ff).float() if case.anyres is None else ff, None, g, case.modal, nl)


def test_ring_marginals_form_end_to_end():
    """Round 6 (verdict r5 #1a): HICOM_RING_MARG=1 -- the value-side pos-emb leaves the ring kernel as marginals and is applied by the merge
    role of readout GEMM 1's launch (v_proj . pe^T).  Opt-in (measured a net loss, profiles/r06_a_ring_marg_ab.txt); the switch is read once
    per process, so the check runs in a child: C1 shape and two small grids against the oracle, and bit-stable from call to call."""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch, cases
from gpu_util import build_module, dev_bf16
from oracle_util import run_oracle
for name in ("G11_c1_shape", "G1_direct_T8", "G10_peaky_direct", "G9_grid"):
    case = cases.build_case(name)
    m = build_module(case)
    ff, fe, g, nl = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), dev_bf16(case.newline)
    with torch.no_grad():
        a = m(ff, fe, g, case.modal, nl).clone()
        b = m(ff, fe, g, case.modal, nl)
    assert torch.equal(a, b), name
    err = float(np.abs(a.float().cpu().numpy() - run_oracle(case)["out"].numpy()).max())
    print(name, err)
    assert err <= 1e-3, (name, err)
print("MARG_OK")
""" % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"), os.path.join(ROOT_DIR, "tests", "golden"))
    env = dict(os.environ, HICOM_RING_MARG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MARG_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_handoff_failure_counters_and_shard_form_predicate():
    """the in-launch hand-offs (query prep granules, the GEMV chain under readout GEMM 2) poison their rows with NaN when a bounded
    spin expires and count it -- hicom_compressor_handoff_failures reads the counters of a plan's workspace (0 after healthy steps); and
    hicom_compressor_takes_shard4 is the ONE predicate behind the sharded step's four-launch form (STREAM writes r0_buf iff FINISH reads it)."""
    from hicom_amd import dist as hd, engine
    m, _, case = _module_and_sd("G11_c1_shape")
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    with torch.no_grad():
        for _ in range(5):
            out = m(ff, fe, g, "video", None)
    assert bool(torch.isfinite(out.float()).all())
    plan = next(iter(m.__dict__["_engine_plans"].values()))
    assert plan.fused and nv.compressor_handoff_failures(plan.args) == (0, 0)
    sp = hd._shard_plan(m, ff, fe, g, 4, None, None, rank=0, world=1)
    for st in sp.sets:
        took = nv.compressor_takes_shard4(st.a_stream)
        assert took == bool(st.a_stream.r0_buf) == bool(st.a_finish.r0_buf)
        assert took                                                  # (the release recipe at hidden 896: the four-launch form)


def test_integration_md_ctypes_example_runs_as_published():
    """the ctypes stub in INTEGRATION.md section 2 had fallen one ABI version behind (a 16-argument hicom_local_attn_fwd
    against the library's 17: the stream handle landed in ctx_f16).  The published block is executed verbatim here and checked
    against the oracle's window attention, so the document cannot drift from include/hicom_hip.h again."""
    import math
    import os
    import re
    from oracle import hicom_oracle as orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C-ABI seam"):]
    block = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "hicom_abi_version() == %d" % nv.ABI_VERSION in block
    block = block.replace('"hicom_amd/libhicom_hip.so"', repr(nv.LIB_PATH))
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    x = synth.synth_inputs(8, 6, 6, 1152, tag="doc")
    ff = torch.from_numpy(x["ff"]).to(torch.bfloat16).cuda()
    fe = torch.from_numpy(x["fe"]).to(torch.bfloat16).cuda()
    g = torch.from_numpy(x["g"]).to(torch.bfloat16).cuda()
    ctx = ns["local_windows"](fe, ff, g)
    torch.cuda.synchronize()
    idx = orc.window_token_index(8, 6, 6, 4, 3)
    k, v = fe.float().cpu().reshape(-1, 1152)[idx], ff.float().cpu().reshape(-1, 1152)[idx]
    s = torch.einsum("d,wnd->wn", g.float().cpu(), k) / math.sqrt(1152)
    want = torch.einsum("wn,wnd->wd", torch.softmax(s, 1), v)
    assert float((ctx.cpu() - want).abs().max()) <= 2e-5


def test_release_step_is_bit_identical_in_its_three_and_four_launch_forms():
    """The executor's A/B switch HICOM_TAIL_LAUNCHES (read once per process): 4 (default) = merge role + chain role in the two GEMM
    launches, 3 (opt-in) = the fused tail launch (hicom_readout_tail_fwd).  Same device functions, same arithmetic order: the packed bf16
    output of a run of release-recipe forwards on changing inputs hashes the same.  (5 = round 4's merge launch + GEMV roles sums its
    single-row layers in another order: it runs here too, finite, not compared bit for bit.)"""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import hashlib, sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "import bench\n"
        "dev = torch.device('cuda', 0)\n"
        "torch.manual_seed(5)\n"
        "m = bench.make_projector(bench.release_config(896, 16), dev)\n"
        "h = hashlib.sha256()\n"
        "with torch.no_grad():\n"
        "    for i in range(6):\n"
        "        g = torch.Generator(device='cuda').manual_seed(100 + i)\n"
        "        a = torch.randn(16, 27, 27, 1152, device=dev, generator=g).bfloat16()\n"
        "        b = torch.randn(16, 27, 27, 1152, device=dev, generator=g).bfloat16()\n"
        "        q = torch.randn(1152, device=dev, generator=g).bfloat16()\n"
        "        out = m(a, b, q, 'video', None)\n"
        "        assert bool(torch.isfinite(out.float()).all())\n"
        "        h.update(out.view(torch.int16).cpu().numpy().tobytes())\n"
        "print('HASH', h.hexdigest())\n")
    got = {}
    for form in ("3", "4", "5"):
        env = dict(os.environ, HICOM_TAIL_LAUNCHES=form)
        r = subprocess.run([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        got[form] = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("HASH")][0]
    assert got["3"] == got["4"] and len(got["5"]) == len(got["4"]), got
