"""End-to-end parity (GPU): the drop-in HIComProjector through the C ABI vs
 (a) the reference's own fp32 outputs (tests/golden/golden_v1.npz) and (b) the CPU oracle,
plus size-independent properties at the benchmark's full 64 x 729 x 1152 size.

Tolerance: BASELINE.json's north_star asks for <= 1e-3 max-abs against the reference's fp32
evaluation on bf16-representable inputs/weights; the kernels' fp32 output is what is compared
(the final cast to bf16 alone costs up to 2^-9 |out|)."""
import numpy as np
import pytest
import torch

import cases
from gpu_util import build_module, dev_bf16, run_native
from oracle_util import run_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-3
# Two HIP paths to the same result (fused executor vs operator-by-operator, dense vs frame-sharded) agree to fp32 noise in
# everything but the local readout: the hot path carries the window contexts / hidden activations as ONE fp16 plane
# (2^-12 relative rounding per activation, hicom_readout16_gemm_fwd) where the stepwise path uses fp32 / bf16 hi+lo, and
# a 1e-7 difference upstream can flip such a rounding.  Still 5x inside the parity tolerance against the oracle.
PATH_TOL = 2e-4

NATIVE_CASES = ["G1_direct_T8", "G2_off_T8", "G2b_off_string", "G3_direct_T7", "G3d_direct_T2", "G3e_off_T3_h2",
                "G3c_off_T10_hw75", "G4_direct_T1", "G4b_image_newline", "G9_grid", "G9_frame", "G9_one_token",
                "G9_flat", "G9_anyres", "G9_anyres_nobase", "G9_local_only", "G9_global_only", "G9_local22",
                "G10_peaky_direct", "G10b_peaky_off", "G11_c1_shape",
                # injector / adaptor variants: FiLM+LN, 64-token MHA, alpha-blended q/k/v/guide adaptors
                "G5_adaptkv", "G5b_adaptqkvg_off", "G6_coarse", "G7_fine", "G7b_guide_override",
                # CLIP-tower branch: every width-768 kernel instantiation
                "G12_clip768_direct", "G12b_clip768_off"]


@pytest.mark.parametrize("name", NATIVE_CASES)
def test_matches_reference_golden(name, golden):
    case = cases.build_case(name)
    got = run_native(case)["out"].float().cpu().numpy()
    if case.sampled:
        assert tuple(golden[f"{name}/out_shape"]) == got.shape
        r, c = cases.sample_index(*got.shape)
        err = np.abs(got[r, c] - golden[f"{name}/out_samples"]).max()
        want = run_oracle(case)["out"].numpy()            # full tensor through the pinned oracle
        err = max(err, np.abs(got - want).max())
    else:
        ref = golden[f"{name}/out"]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
    assert err <= TOL, f"{name}: max-abs {err:.3e}"


# Heavy-tailed channels (round-5 verdict, weak #2): 12 channels x 60 with a non-zero mean in both visual tensors, the statistics of real
# SigLIP hidden_states[-2] (reference encoder.py:253-259).  The outputs grow with them (max |out| 2.1 at hidden 64, 11.5 at the C1 shape)
# and so does every rounding: the HIP path carries window contexts / hidden activations as ONE fp16 plane (2^-12 relative), the
# reference's own bf16 inference arithmetic rounds at 2^-9.  Tolerance: HEAVY_REL of max |out| (the absolute 1e-3 of the N(0,1) cases is
# the same bar at max |out| ~ 1), asserted beside the reference-bf16-vs-reference-fp32 figure stored in the fixture, which the HIP path
# must beat by HEAVY_VS_REF_BF16.
HEAVY_REL = 1e-3
HEAVY_VS_REF_BF16 = 8.0


@pytest.mark.parametrize("name", ["G13_outlier_direct", "G13b_outlier_off", "G13c_outlier_c1"])
def test_heavy_tailed_channels_match_reference_golden(name, golden):
    case = cases.build_case(name)
    got = run_native(case)["out"].float().cpu().numpy()
    ref_bf16_err, out_max = (float(v) for v in golden[f"{name}/ref_bf16_max_abs"])
    if case.sampled:
        assert tuple(golden[f"{name}/out_shape"]) == got.shape
        r, c = cases.sample_index(*got.shape)
        err = np.abs(got[r, c] - golden[f"{name}/out_samples"]).max()
        want = run_oracle(case)["out"].numpy()            # full tensor through the pinned oracle
        err = max(err, np.abs(got - want).max())
    else:
        ref = golden[f"{name}/out"]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
    print(f"{name}: max-abs {err:.3e}  max|out| {out_max:.3f}  relative {err / out_max:.3e}  reference bf16 vs fp32 {ref_bf16_err:.3e}")
    assert err <= HEAVY_REL * max(out_max, 1.0), f"{name}: max-abs {err:.3e} of max|out| {out_max:.3f}"
    assert err * HEAVY_VS_REF_BF16 <= ref_bf16_err, f"{name}: {err:.3e} against the reference's own bf16 deviation {ref_bf16_err:.3e}"


@pytest.mark.parametrize("name", ["G8_clip_scale", "G8b_clip_coarse", "G8c_clip_fine", "G8d_clip_direct_adaptg", "G8e_clip_adaptkv"])
def test_clip_scale_local_matches_golden(golden, name):
    """Clip-scale on the LOCAL stage (reference projector.py:527-529, :549); with an injector or an adapted guide the guide rows are
    normalised BEFORE injection and the injected query is not normalised again (G8b / G8c / G8d)."""
    case = cases.build_case(name)
    got = run_native(case)["local"].float().cpu().numpy()
    ref = golden[f"{name}/local"]
    assert got.shape == ref.shape and np.abs(got - ref).max() <= TOL


@pytest.mark.parametrize("name", ["G8_clip_scale", "G8b_clip_coarse", "G8c_clip_fine", "G8d_clip_direct_adaptg"])
def test_clip_scale_global_matches_golden(golden, name):
    """Clip-scale on the GLOBAL stage (reference projector.py:184-191: queries and PROJECTED keys L2-normalised over the full
    width, logits * exp(logit_scale) + logit_bias): the stored `global` vector, direct compressor call as the fixture makes it;
    G8b / G8c: 32 distinct injected queries (coarse / fine), G8d: the adapted guide."""
    case = cases.build_case(name)
    m = build_module(case)
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    gs, gb = (torch.tensor(v, device="cuda") for v in case.logit["glob"])
    with torch.no_grad():
        got = m.global_compressor(ff, fe, g, case.modal, gs, gb)
    ref = golden[f"{name}/global"]
    assert tuple(got.shape) == ref.shape and np.abs(got.float().cpu().numpy() - ref).max() <= TOL


def test_use_clip_scale_config_constructs_and_runs(golden):
    """A projector configured with use_clip_scale='local,global' constructs (the reference reads the logits from the SigLIP
    checkpoint there and keeps them as PARAMETERS `local_logit_scale` ... of the projector, reference projector.py:655-670), refuses to
    run while they are unset, and reproduces G8's local and global vectors through HIComProjector.forward -- with the logits given by
    set_clip_logits() and with the logits arriving through load_state_dict() of a checkpoint that carries them (round 5: the
    state-dict schema of a clip-scale projector equals the reference's)."""
    import hicom_amd
    case = cases.build_case("G8_clip_scale")
    case.cfg.use_clip_scale = "local,global"
    sd = {k: torch.from_numpy(v.copy()) for k, v in case.sd.items()}
    m = hicom_amd.build_vision_projector(case.cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert sorted(missing) == ["global_logit_bias", "global_logit_scale", "local_logit_bias", "local_logit_scale"] and not unexpected
    m = m.to(torch.bfloat16).cuda().eval()
    m.return_fp32 = True
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    loc, glo = golden["G8_clip_scale/local"], golden["G8_clip_scale/global"]
    nl = loc.reshape(-1, loc.shape[-1]).shape[0]
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            m(ff, fe, g, case.modal, None)
        m.set_clip_logits(local=case.logit["local"], glob=case.logit["glob"])
        out = m(ff, fe, g, case.modal, None).float().cpu().numpy()
    assert out.shape[0] == nl + glo.shape[0]
    assert np.abs(out[:nl] - loc.reshape(nl, -1)).max() <= TOL and np.abs(out[nl:] - glo).max() <= TOL
    # the reference's checkpoint format: the four logits are entries of the state dict
    full = {k: v.float().cpu() for k, v in m.state_dict().items()}
    assert float(full["local_logit_scale"]) == case.logit["local"][0] and float(full["global_logit_bias"]) == case.logit["glob"][1]
    m2 = hicom_amd.build_vision_projector(case.cfg)
    m2.load_state_dict(full, strict=True)
    m2 = m2.to(torch.bfloat16).cuda().eval()
    m2.return_fp32 = True
    with torch.no_grad():
        out2 = m2(ff, fe, g, case.modal, None).float().cpu().numpy()
        assert np.array_equal(out2, out)
        # an in-place change of a logit parameter is seen by the next forward (the plans bake the floats in)
        m2.local_logit_scale.fill_(0.5)
        out3 = m2(ff, fe, g, case.modal, None).float().cpu().numpy()
    assert np.abs(out3[:nl] - out[:nl]).max() > 1e-3 and np.array_equal(out3[nl:], out[nl:])


@pytest.mark.parametrize("name", ["G3b_direct_T5_raises", "G4c_image_T2_raises"])
def test_reference_error_cases(name):
    case = cases.build_case(name)
    with pytest.raises((RuntimeError, ValueError)):
        run_native(case)


def test_bf16_output_is_the_cast_of_fp32():
    case = cases.build_case("G1_direct_T8")
    a = run_native(case, fp32_out=True)["out"]
    b = run_native(case, fp32_out=False)["out"]
    assert b.dtype == torch.bfloat16 and torch.equal(a.to(torch.bfloat16), b)


def test_direct_mode_global_rows_identical():
    case = cases.build_case("G1_direct_T8")
    out = run_native(case)["out"]
    g = out[-32:]
    assert torch.equal(g, g[:1].expand_as(g))             # 32 bit-identical rows


def test_post_process_visual_feature_standalone():
    import hicom_amd
    from types import SimpleNamespace
    from oracle import hicom_oracle as orc
    feat = torch.randn(2, 3, 5, 64).to(torch.bfloat16)
    nl = torch.randn(64).to(torch.bfloat16)
    for merge, pos in (("spatial_unpad", "grid"), ("spatial_unpad", "frame"), ("spatial_unpad", "one_token"),
                       ("spatial_unpad", "no_token"), ("flat", "grid")):
        cfg = SimpleNamespace(mm_patch_merge_type=merge, mm_newline_position=pos)
        want = orc.post_process(cfg, feat.float(), "video", nl.float(), False)
        got = hicom_amd.post_process_visual_feature(cfg, feat.cuda(), "video", nl.cuda(), False)
        assert torch.equal(got.float().cpu(), want)        # pure data movement: bit exact
    cfg = SimpleNamespace(mm_patch_merge_type="spatial", mm_newline_position="grid")
    want = orc.post_process(cfg, feat[:1].float(), "image", nl.float(), True)
    got = hicom_amd.post_process_visual_feature(cfg, feat[:1].cuda(), "image", nl.cuda(), True)
    assert torch.equal(got.float().cpu(), want)


# ---- full benchmark size (C2: 64 x 729 x 1152, H = 896): properties the domain offers ----------
@pytest.fixture(scope="module")
def c2():
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": 896, "max_num_frames": 64})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="c2")
    x = synth.synth_inputs(64, 27, 27, 1152, tag="c2")
    case = SimpleNamespace(cfg=cfg, sd=sd)
    m = build_module(case)
    return m, dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"]), case


def test_c2_shape_and_first_group_matches_oracle(c2):
    """The first 4-frame group's 81 local tokens depend only on frames 0-3: check them, at full
    size, against the oracle evaluated on those 4 frames (which it finishes in seconds)."""
    m, ff, fe, g, case = c2
    with torch.no_grad():
        out = m(ff, fe, g, "video", None)
    assert out.shape == (64 // 4 * 81 + 32, 896) and bool(torch.isfinite(out).all())
    from oracle import hicom_oracle as orc
    sd = {k: torch.from_numpy(v) for k, v in case.sd.items()}
    spec = orc.parse_projector_type(case.cfg.mm_projector_type)["local"]
    want = orc.local_forward(spec, "direct", sd, "local_compressor", ff[:4].float().cpu(), fe[:4].float().cpu(),
                             g.float().cpu(), "video").reshape(81, 896)
    assert float((out[:81].cpu() - want).abs().max()) <= TOL
    assert torch.equal(out[-32:], out[-32:-31].expand(32, -1))


@pytest.mark.parametrize("T,hidden,with_global_oracle", [(32, 3584, True), (128, 896, False)])
def test_other_baseline_configs_against_oracle(T, hidden, with_global_oracle):
    """BASELINE.json's other single-GPU shapes: C4 (32 frames, the 7B LLM's hidden size 3584) and one GPU's share of
    C5 (128 frames): first and last 4-frame group against the oracle on those frames, the global rows against the
    oracle on the whole clip (C4: 23k keys, seconds on the host) or against the frame-sharded composition (C5)."""
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": hidden, "max_num_frames": max(64, T)})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="c%d" % T)
    x = synth.synth_inputs(T, 27, 27, 1152, tag="c%d" % T)
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    ff, fe, g = dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"])
    nw = T // 4 * 81
    with torch.no_grad():
        out = m(ff, fe, g, "video", None)
        again = m(ff, fe, g, "video", None)
    assert out.shape == (nw + 32, hidden) and bool(torch.isfinite(out).all()) and torch.equal(out, again)
    assert torch.equal(out[-32:], out[-32:-31].expand(32, -1))
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    spec = orc.parse_projector_type(cfg.mm_projector_type)
    for lo in (0, T - 4):
        want = orc.local_forward(spec["local"], "direct", sdt, "local_compressor", ff[lo:lo + 4].float().cpu(),
                                 fe[lo:lo + 4].float().cpu(), g.float().cpu(), "video").reshape(81, hidden)
        assert float((out[lo // 4 * 81:lo // 4 * 81 + 81].cpu() - want).abs().max()) <= TOL, lo
    if with_global_oracle:
        want = orc.global_forward(spec["global"], "direct", sdt, "global_compressor", ff.float().cpu(), g.float().cpu())
        assert float((out[-32:].cpu() - want.reshape(32, hidden)).abs().max()) <= TOL
    else:
        gc = m.global_compressor
        with torch.no_grad():
            q_in, n_rows = gc.injected_queries(g)
            parts = [gc.partial_context(ff[lo:lo + T // 4], q_in, t_offset=lo) for lo in range(0, T, T // 4)]
            glob = torch.empty((32, hidden), dtype=torch.float32, device="cuda")
            gc.finish(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), q_in, glob, 0, n_rows)
        assert float((glob - out[-32:]).abs().max()) <= PATH_TOL


def test_c2_frame_shards_compose(c2):
    """Sharding the 64 frames 4-ways (absolute frame offsets) and combining the partial softmax
    states reproduces the unsharded result; local tokens of a shard equal the matching slice."""
    m, ff, fe, g, _ = c2
    gc, lc = m.global_compressor, m.local_compressor
    with torch.no_grad():
        full = m(ff, fe, g, "video", None)
        q_in, n_rows = gc.injected_queries(g)
        mls, accs = [], []
        for r in range(4):
            sl = slice(16 * r, 16 * r + 16)
            ml, acc, _ = gc.partial_context(ff[sl], q_in, t_offset=16 * r)
            mls.append(ml), accs.append(acc)
        out = torch.empty((32, 896), dtype=torch.float32, device="cuda")
        gc.finish(torch.stack(mls), torch.stack(accs), q_in, out, 0, n_rows)
        assert float((out - full[-32:]).abs().max()) <= PATH_TOL
        ctx, grid = lc.window_context(ff[16:32], fe[16:32], g, "video", None, None)
        loc = torch.empty((ctx.shape[0], 896), dtype=torch.float32, device="cuda")
        lc.readout_into(ctx, loc, 0, 0)
        # stepwise local path = VALU window kernel, forward = fused MFMA kernel: same math, fp32 noise
        assert float((loc - full[4 * 81:8 * 81]).abs().max()) <= PATH_TOL


def test_c2_uniform_attention_known_answer(c2):
    """Zero guide => every local logit is 0 => each local context is the plain mean of its 36 value
    rows (checked through linearity of the readout on the mean rows)."""
    m, ff, fe, g, _ = c2
    lc = m.local_compressor
    with torch.no_grad():
        ctx, _ = lc.window_context(ff, fe, torch.zeros_like(g), "video", None, None)
        x = ff.float().view(16, 4, 9, 3, 9, 3, 1152).permute(0, 2, 4, 1, 3, 5, 6).reshape(1296, 36, 1152)
        assert float((ctx - x.mean(dim=1)).abs().max()) <= 2e-6


@pytest.mark.parametrize("name", ["G1_direct_T8", "G2_off_T8", "G9_grid", "G9_one_token", "G3_direct_T7", "G9_local_only",
                                  "G9_global_only", "G10b_peaky_off",
                                  # round 5: injected / adapted queries made in front of the call, handed in as f32 rows (external_queries)
                                  "G6_coarse", "G7_fine", "G7b_guide_override", "G5b_adaptqkvg_off"])
def test_executor_equals_stepwise(name):
    """The one-call native executor (two streams, fused fold) and the operator-by-operator path agree."""
    case = cases.build_case(name)
    m = build_module(case)
    assert m._executor_covers()
    ff, fe, g, nl = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), dev_bf16(case.newline)
    with torch.no_grad():
        a = m(ff, fe, g, case.modal, nl)
        b = m.forward_stepwise(ff, fe, g, case.modal, nl)
        a2 = m(ff, fe, g, case.modal, nl)                  # workspace reuse: same bits on the second call
        assert len(m.__dict__.get("_engine_plans", {})) == 1 and next(iter(m._engine_plans.values())).hits == 2
        if g is not None and m._external_queries():
            # the plan's query rows follow the guide of EACH call: another guide, then the first one again
            g2 = (g.float() * -0.5 + 0.25).to(g.dtype)
            c = m(ff, fe, g2, case.modal, nl)
            c_step = m.forward_stepwise(ff, fe, g2, case.modal, nl)
            assert float((c - c_step).abs().max()) <= PATH_TOL and float((c - a).abs().max()) > 1e-3
            assert torch.equal(m(ff, fe, g, case.modal, nl), a)
    assert a.shape == b.shape and float((a - b).abs().max()) <= PATH_TOL
    assert torch.equal(a, a2)


def test_forward_deferred_equals_forward(c2):
    """forward_deferred (no join of the side stream at the end of a call) gives, for DIFFERENT videos issued back
    to back, exactly the bits of the joined forward of each -- also when interleaved with joined calls."""
    m, ff, fe, g, _ = c2
    vids = [(ff, fe), (fe, ff), (ff.flip(0).contiguous(), fe), (ff, fe)]
    with torch.no_grad():
        want = [m(a, b, g, "video", None).clone() for a, b in vids]
        for rep in range(3):
            got = [m.forward_deferred(a, b, g, "video", None) for a, b in vids for _ in range(2)]
            mid = m(fe, ff, g, "video", None)                       # a joined call right behind deferred ones
            more = [m.forward_deferred(a, b, g, "video", None) for a, b in vids]
            torch.cuda.synchronize()
            assert torch.equal(mid, want[1])
            for k, (o, ev) in enumerate(got):
                assert ev.query() and torch.equal(o, want[k // 2]), (rep, k)
            for k, (o, ev) in enumerate(more):
                assert torch.equal(o, want[k]), (rep, "more", k)


def test_alternating_guides_share_one_workspace(c2):
    """Cached plans of two guides alternate on one workspace (different folded queries, deferred and joined calls
    back to back): every result equals the isolated forward of its guide."""
    m, ff, fe, g, _ = c2
    g2 = (g.float() * -0.7 + 0.05).to(g.dtype)
    with torch.no_grad():
        want = {}
        for name, guide in (("g", g), ("g2", g2)):
            torch.cuda.synchronize()
            want[name] = m(ff, fe, guide, "video", None).clone()
            torch.cuda.synchronize()
        assert not torch.equal(want["g"], want["g2"])
        seq = ["g", "g2", "g2", "g", "g2", "g", "g", "g2"] * 3
        outs = [m(ff, fe, g if k == "g" else g2, "video", None) for k in seq]
        outs_d = [m.forward_deferred(ff, fe, g if k == "g" else g2, "video", None)[0] for k in seq]
        torch.cuda.synchronize()
        for k, o, od in zip(seq, outs, outs_d):
            assert torch.equal(o, want[k]) and torch.equal(od, want[k]), k


def test_random_mix_of_shapes_guides_and_joins(c2):
    """150 back-to-back calls drawn at random from {three videos of two shapes} x {two guides} x {deferred, joined},
    every call with FRESH input tensors (clones: new pointers, as a serving loop hands over): every result equals, bit
    for bit, the isolated forward of its (video, guide) -- plans, workspaces and events are shared across all of them."""
    import random
    m, ff, fe, g, _ = c2
    gen = torch.Generator(device="cuda").manual_seed(5)
    small = (torch.randn(16, 27, 27, 1152, device="cuda", generator=gen).bfloat16(),
             torch.randn(16, 27, 27, 1152, device="cuda", generator=gen).bfloat16())
    vids = {"a": (ff, fe), "b": small, "c": (fe, ff)}
    gs = {"g1": g, "g2": (g.float() * -0.7 + 0.05).to(g.dtype)}
    with torch.no_grad():
        want = {}
        for v, (x, y) in vids.items():
            for k, gg in gs.items():
                want[v, k] = m(x, y, gg, "video", None).clone()
                torch.cuda.synchronize()
        n_plans = len(m.__dict__["_engine_plans"])
        rnd = random.Random(11)
        seq = [(rnd.choice("abc"), rnd.choice(["g1", "g2"]), rnd.random() < 0.7, rnd.random() < 0.5) for _ in range(150)]
        outs = []
        for v, k, deferred, fresh in seq:
            x, y = vids[v]
            gg = gs[k]
            if fresh and v == "b":
                x, y, gg = x.clone(), y.clone(), gg.clone()          # new buffers: same plan, patched pointers
            elif fresh:
                gg = gg.clone()
            if deferred:
                outs.append(m.forward_deferred(x, y, gg, "video", None)[0])
            else:
                outs.append(m(x, y, gg, "video", None))
            del x, y, gg
        torch.cuda.synchronize()
        bad = [i for i, ((v, k, _, _), o) in enumerate(zip(seq, outs)) if not torch.equal(o, want[v, k])]
        assert len(m.__dict__["_engine_plans"]) == n_plans           # no plan was rebuilt for a new buffer
    assert not bad, bad[:10]


def test_weight_update_invalidates_plan(c2):
    """In-place updates and `.data` swaps of a parameter are seen by the next call (plans bake weight pointers)."""
    m, ff, fe, g, _ = c2
    with torch.no_grad():
        base = m(ff[:8], fe[:8], g, "video", None).clone()
        w = m.local_compressor.readout[2].bias
        orig = w.detach().clone()
        w.add_(1.0)
        moved = m(ff[:8], fe[:8], g, "video", None).clone()
        w.copy_(orig)
        assert float((moved[:-32] - base[:-32] - 1.0).abs().max()) < 2e-2
        old = w.data
        w.data = (old.float() + 2.0).to(old.dtype)                   # pointer swap, version counter untouched
        swapped = m(ff[:8], fe[:8], g, "video", None).clone()
        w.data = old
        assert float((swapped[:-32] - base[:-32] - 2.0).abs().max()) < 2e-2
        assert torch.equal(m(ff[:8], fe[:8], g, "video", None), base)


def test_graph_replay_equals_eager(c2):
    """hipGraph capture / replay of a plan (two streams, fork / join as captured event nodes) reproduces eager bits."""
    m, ff, fe, g, _ = c2
    with torch.no_grad():
        want = m(ff[:16], fe[:16], g, "video", None).clone()
        m.graph_replay = True
        try:
            outs = [m(ff[:16], fe[:16], g, "video", None) for _ in range(4)]
            torch.cuda.synchronize()
        finally:
            m.graph_replay = False
            m._invalidate_plans()
        for o in outs:
            assert torch.equal(o, want)


def test_grad_mode_builds_a_graph_or_refuses(c2):
    """With autograd on and trainable parameters, forward() must never hand back a silently detached tensor."""
    m, ff, fe, g, _ = c2
    out = m(ff[:4], fe[:4], g, "video", None)                        # grad mode on, parameters require grad
    assert out.requires_grad and out.grad_fn is not None
    with pytest.raises(RuntimeError):
        m.local_compressor(ff[:4], fe[:4], g, "video")
    with pytest.raises(RuntimeError):
        m.forward_deferred(ff[:4], fe[:4], g, "video", None)
    with torch.no_grad():
        assert not m(ff[:4], fe[:4], g, "video", None).requires_grad


def test_c2_full_output_matches_oracle(c2):
    """ALL 1328 x 896 outputs of the benchmark configuration against the fp32 oracle of the whole 64-frame clip
    (reference projector.py:676-708; ~1 s of host time): every window of every workgroup of the ring kernel, the
    global rows, the packing."""
    m, ff, fe, g, case = c2
    from oracle import hicom_oracle as orc
    with torch.no_grad():
        out = m(ff, fe, g, "video", None)
        sd = {k: torch.from_numpy(v) for k, v in case.sd.items()}
        want = orc.projector_forward(case.cfg, sd, ff.float().cpu(), fe.float().cpu(), g.float().cpu(), "video", None)
    assert out.shape == want.shape == (1328, 896)
    err = (out.cpu() - want).abs()
    assert float(err.max()) <= TOL, (float(err.max()), int(err.argmax()) // 896)


def _emulate_ranks(m, ff, fe, g, world, finish_rank):
    """The N > 1 device path without a collective: every "rank" runs ITS shard's STREAM phase (absolute frame offsets,
    state + local tokens into its send buffer, merge on the comm stream); the all-gather is emulated by copying the
    send buffers into `finish_rank`'s receive buffer; that rank's FINISH phase produces the full output."""
    from hicom_amd import dist as hd, native as nv
    T = ff.shape[0]
    per = T // world
    sends, keep = [], []
    for r in range(world):
        a, b = ff[r * per:(r + 1) * per], fe[r * per:(r + 1) * per]              # dense slices along dim 0
        plan = hd._shard_plan(m, a, b, g, T, None, None, rank=r, world=world)
        st = plan.sets[0]
        plan.set_inputs(st, a, b, g, st.out)
        nv.compressor_fwd(st.a_stream)
        torch.cuda.synchronize()
        sends.append(st.mine.clone())
        keep.append(plan)
    plan = keep[finish_rank]
    st = plan.sets[0]
    for r in range(world):
        st.everyone[r].copy_(sends[r])
    st.out.fill_(float("nan"))
    torch.cuda.synchronize()
    nv.compressor_fwd(st.a_finish)
    torch.cuda.synchronize()
    return st.out.clone()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_device_path_emulated_on_one_gpu(c2, world):
    """2- and 4-rank worlds on one GPU (see _emulate_ranks) against the dense forward of all frames."""
    m, ff, fe, g, _ = c2
    with torch.no_grad():
        want = m(ff, fe, g, "video", None)
        got = _emulate_ranks(m, ff, fe, g, world, 0)
    # (not bit-equal: a shard's workgroups hold fewer windows each, so windows meet the 16-token tiles at other
    # offsets and their sums associate differently)
    assert got.shape == want.shape and float((got - want).abs().max()) <= PATH_TOL


@pytest.mark.parametrize("world", [2, 4])
def test_adaptkv_recipe_in_the_executor_and_sharded(world):
    """The second released recipe (`local43_adaptkv_global32`, use_guide = direct: k / v adaptor MLPs + LayerNorm blend over ALL
    tokens, reference projector.py:431-457, :533-534) is first-class: ONE C call (hicom_compressor_fwd runs the four
    dense GEMMs and the blend-fused window attention), equal to the operator-by-operator path and to the oracle on every output,
    and it shards over frames (the adaptors are token-wise: shard-local) -- 2- and 4-rank emulation against the dense forward."""
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    T, H, W = 16, 9, 9
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": "local43_adaptkv_global32", "hidden_size": 256, "max_num_frames": T})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="akv")
    for k in list(sd):
        if k.endswith("_alpha"):
            sd[k] = np.full_like(sd[k], 0.4)                          # (the reference initialises the blends at 0: dead branches)
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    gen = torch.Generator(device="cuda").manual_seed(37)
    ff = torch.randn(T, H, W, 1152, device="cuda", generator=gen).bfloat16()
    fe = torch.randn(T, H, W, 1152, device="cuda", generator=gen).bfloat16()
    g = torch.randn(1152, device="cuda", generator=gen).bfloat16()
    with torch.no_grad():
        got = m(ff, fe, g, "video", None)
        assert next(iter(m.__dict__["_engine_plans"].values())).hits == 1        # the one-call executor took it
        m.use_executor = False
        step = m(ff, fe, g, "video", None)
        m.use_executor = True
        assert torch.equal(m(ff, fe, g, "video", None), got)                      # (second call: plan hit, same bits)
        shard = _emulate_ranks(m, ff, fe, g, world, world - 1)
        o, ev = m.forward_deferred(ff, fe, g, "video", None)
        ev.synchronize()
    assert float((got - step).abs().max()) <= PATH_TOL and float((o - got).abs().max()) <= PATH_TOL
    assert shard.shape == got.shape and float((shard - got).abs().max()) <= PATH_TOL
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    want = orc.projector_forward(cfg, sdt, ff.float().cpu(), fe.float().cpu(), g.float().cpu(), "video", None)
    assert float((got.cpu() - want).abs().max()) <= TOL


def test_guide_off_wide_global_kernel_matches_narrow():
    """Guide off at 16 frames of the full grid (288 folded query rows): the wide global stream kernel (two row groups
    per workgroup, three-deep ring) against the one-row-group kernel it replaces, and the 81 x 4 local tokens of the
    first groups + the 32 distinct global rows against the oracle."""
    import os
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": "local43_global32", "use_guide": None,
                             "hidden_size": 896, "max_num_frames": 64})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="off16")
    x = synth.synth_inputs(16, 27, 27, 1152, tag="off16")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    ff, fe, g = dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"])
    with torch.no_grad():
        wide = m(ff, fe, g, "video", None).clone()
        os.environ["HICOM_GLOBAL_NARROW"] = "1"
        try:
            narrow = m(ff, fe, g, "video", None).clone()
        finally:
            del os.environ["HICOM_GLOBAL_NARROW"]
        torch.cuda.synchronize()
    assert wide.shape == (4 * 81 + 32, 896)
    assert float((wide - narrow).abs().max()) <= PATH_TOL
    assert not torch.equal(wide[-32:-31], wide[-31:-30])                 # 32 DISTINCT global rows in this mode
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    spec = orc.parse_projector_type(cfg.mm_projector_type)
    want = orc.global_forward(spec["global"], None, sdt, "global_compressor", ff.float().cpu(), None)
    assert float((wide[-32:].cpu() - want.reshape(32, 896)).abs().max()) <= TOL


@pytest.mark.parametrize("T,t_offset", [(16, 0), (64, 0), (12, 40)])
def test_global_stream_in_kernel_marginals_match_the_logit_tensor_path(T, t_offset, monkeypatch):
    """Guide off (288 folded query rows, value-side pos-emb): the many-row stream kernel accumulates the t / y / x marginals of the
    softmax weights itself (hicom_global_stream_marg_fwd + hicom_global_merge_marg_fwd, no [288, N] logit tensor) -- against the
    path that writes the logits and takes per-frame marginals from them afterwards (hicom_global_stream_fwd +
    hicom_global_merge_fwd), same (M, L), contexts to fp32 noise, with a frame offset (a shard of a longer clip), and with the
    logits requested as well (the training forward): then they equal the other path's bit for bit."""
    from types import SimpleNamespace
    from hicom_amd import native as nv, synth
    from oracle import hicom_oracle as orc
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": "local43_global32", "use_guide": None,
                             "hidden_size": 896, "max_num_frames": 64})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="marg")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    gc = m.global_compressor
    gen = torch.Generator(device="cuda").manual_seed(5)
    ff = torch.randn(T, 27, 27, 1152, device="cuda", generator=gen).bfloat16()
    with torch.no_grad():
        q_in, _ = gc.injected_queries(None)
        N = T * 729
        assert nv.global_stream_has_marg(N, 1152, 288, 27, 27, nv.global_stream_nparts(N, 288))
        ml, acc, none = gc.partial_context(ff, q_in, t_offset=t_offset)
        ml2, acc2, sc2 = gc.partial_context(ff, q_in, t_offset=t_offset, need_scores=True)
        assert none is None and sc2 is not None and torch.equal(ml, ml2) and torch.equal(acc, acc2)
        monkeypatch.setattr(nv, "global_stream_has_marg", lambda *a: False)
        ml3, acc3, sc3 = gc.partial_context(ff, q_in, t_offset=t_offset)
        torch.cuda.synchronize()
    assert torch.equal(ml, ml3) and torch.equal(sc2[:288, :N], sc3[:288, :N])
    ctx, ctx3 = acc / ml[:, 1:2], acc3 / ml3[:, 1:2]
    assert bool(torch.isfinite(ctx).all()) and float((ctx - ctx3).abs().max()) <= 2e-5
    # the positional part is not negligible here: without it the contexts differ visibly
    pe = gc.pos_and_kpe(t_offset + T, 27, 27, ff.device)[0]
    assert float(pe.abs().max()) > 0.5


def test_global_stream_backward_many_row_form_matches_the_one_row_group_kernel(monkeypatch):
    """Attention backward over the token stream at 288 rows (guide off / coarse / fine): the many-row kernel (two row groups per
    workgroup, dS handed over through LDS, positional marginals of dS taken in the kernel -- no [rows, N] dS tensor) against the
    one-row-group kernel that writes dS: same sum dS . x per row, same t / y / x marginals."""
    import os
    from hicom_amd import native as nv
    g = torch.Generator(device="cuda").manual_seed(8)
    T, H, W, E, R = 12, 27, 27, 1152, 288
    N = T * H * W
    x = torch.randn(N, E, device="cuda", generator=g).bfloat16()
    dctx = torch.randn(R, E, device="cuda", generator=g) * 0.05
    dhi, dlo = torch.empty(R, E, device="cuda", dtype=torch.bfloat16), torch.empty(R, E, device="cuda", dtype=torch.bfloat16)
    nv.split_bf16(dctx, R, dhi, dlo)
    P = T + H + W
    pos_b = torch.randn(R, P, device="cuda", generator=g) * 0.1
    stride = (N + 15) // 16 * 16
    s_in = torch.randn(R, stride, device="cuda", generator=g)
    ml = torch.stack([s_in[:, :N].max(1).values, torch.exp(s_in[:, :N] - s_in[:, :N].max(1, keepdim=True).values).sum(1)], 1).contiguous()
    delta = torch.randn(R, device="cuda", generator=g) * 0.1
    nparts = nv.global_stream_nparts(N, R)
    assert nv.global_stream_has_marg(N, E, R, H, W, nparts)
    part = torch.empty(nparts, R, E, device="cuda")
    pm = torch.empty(nparts, R, nv.global_stream_marg_width(H, W), device="cuda")
    nv.global_stream_bwd(x, N, dhi, dlo, pos_b, H, W, 0, T, T + H, s_in, ml, delta, None, part, R, part_marg=pm)
    mT = torch.zeros(R, T, device="cuda")
    mT.index_add_(1, nv.marg_frame_index(N, H, W, nparts, x.device).reshape(-1), pm[:, :, :8].permute(1, 0, 2).reshape(R, -1))
    mY, mX = pm[:, :, 16:16 + H].sum(0), pm[:, :, 48:48 + W].sum(0)
    # with dS written as well: same partial sums bit for bit
    part2, ds2 = torch.empty_like(part), torch.empty(R, stride, device="cuda")
    nv.global_stream_bwd(x, N, dhi, dlo, pos_b, H, W, 0, T, T + H, s_in, ml, delta, ds2, part2, R, part_marg=torch.empty_like(pm))
    assert torch.equal(part, part2)
    os.environ["HICOM_GLOBAL_NARROW"] = "1"
    try:
        part1, ds1 = torch.empty_like(part), torch.empty(R, stride, device="cuda")
        nv.global_stream_bwd(x, N, dhi, dlo, pos_b, H, W, 0, T, T + H, s_in, ml, delta, ds1, part1, R)
    finally:
        del os.environ["HICOM_GLOBAL_NARROW"]
    torch.cuda.synchronize()
    scale = float(ds1[:, :N].abs().max())
    assert float((ds2[:, :N] - ds1[:, :N]).abs().max()) <= 1e-5 * scale
    a, b = part.sum(0), part1.sum(0)
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    dS = ds1[:, :N].view(R, T, H, W)
    for got, want in ((mT, dS.sum((2, 3))), (mY, dS.sum((1, 3))), (mX, dS.sum((1, 2)))):
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-6


@pytest.mark.parametrize("per,finish_rank", [(64, 5), (128, 6)])
def test_c3_c5_eight_rank_emulation(per, finish_rank):
    """BASELINE configs[2] (512 frames over 8 GPUs, 64 per GPU) and configs[4] (1024 frames, 128 per GPU) on one GPU:
    the eight ranks' STREAM phases one after the other (frame offsets up to 896), the all-gather replaced by copies,
    one rank's FINISH phase -- against the dense forward of all frames (which takes the multi-round form of the stream
    kernel), and the first / a middle / the last 4-frame group plus the 32 global rows (softmax over ALL keys of the clip)
    against the oracle."""
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    world = 8
    T = world * per
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": 896, "max_num_frames": T})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="c3")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    gen = torch.Generator(device="cuda").manual_seed(31)
    ff = torch.randn(T, 27, 27, 1152, device="cuda", generator=gen).bfloat16()
    fe = torch.randn(T, 27, 27, 1152, device="cuda", generator=gen).bfloat16()
    g = torch.randn(1152, device="cuda", generator=gen).bfloat16()
    with torch.no_grad():
        want = m(ff, fe, g, "video", None)
        assert want.shape == (T // 4 * 81 + 32, 896) and bool(torch.isfinite(want).all())
        got = _emulate_ranks(m, ff, fe, g, world, finish_rank)
    assert got.shape == want.shape and float((got - want).abs().max()) <= PATH_TOL
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    spec = orc.parse_projector_type(cfg.mm_projector_type)
    for lo in (0, (world - 3) * per + 8, T - 4):                 # last two sit in shards with frame offset >= 5/8 T
        ref = orc.local_forward(spec["local"], "direct", sdt, "local_compressor", ff[lo:lo + 4].float().cpu(),
                                fe[lo:lo + 4].float().cpu(), g.float().cpu(), "video").reshape(81, 896)
        assert float((got[lo // 4 * 81:lo // 4 * 81 + 81].cpu() - ref).abs().max()) <= TOL, lo
    # global rows against the ORACLE on all 373k (C3) / 746k (C5) keys, not only against the build's own dense forward:
    # the reference's un-folded formulation 32 frames at a time with the softmax carried across the chunks
    # (oracle.global_forward_chunked, pinned against global_forward and the golden rows on the CPU): ~2 / 4 TFLOP of host
    # GEMMs, well under 1 GB of host memory
    ref = orc.global_forward_chunked(spec["global"], "direct", sdt, "global_compressor", ff.float().cpu(), g.float().cpu(), chunk_frames=32)
    assert float((got[-32:].cpu() - ref.reshape(32, 896)).abs().max()) <= TOL


@pytest.mark.parametrize("world,per,newline", [(2, 8, None), (4, 4, None), (8, 8, None), (2, 8, "grid")])
def test_direct_all_gather_form_emulated(world, per, newline):
    """Round 6: the FINISH call of the frame-sharded step enqueues its all-gather ITSELF (hicom_compressor_args.ag_fn: RCCL's ncclAllGather
    through the process group's communicator; without newline rows the token blocks go straight into the output's rows and a second
    all-gather of the same group carries the states).  On the pool's one-GPU boxes that path only ever runs at world size 1 through
    real RCCL, so here every rank of a 2 / 4 / 8-rank world runs it on ONE GPU with a TEST DOUBLE behind the same three C entry points:
    the double copies every rank's send buffer into its slot of the receive buffer (what an all-gather does) -- offsets, strides, the
    grouped form, the state sets and the FINISH phase are the product's.  Every rank's result against the unsharded forward."""
    import ctypes
    from types import SimpleNamespace
    from hicom_amd import dist as hd, native as nv_, synth
    from oracle import hicom_oracle as orc
    T = world * per
    over = {"hidden_size": 128, "max_num_frames": max(16, T)}
    if newline:
        over.update(mm_patch_merge_type="spatial_unpad", mm_newline_position=newline)
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, **over})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="dag")
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    x = synth.synth_inputs(T, 27, 27, 1152, tag=f"dag{world}")
    ff, fe, g = dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"])
    nl = dev_bf16(synth.normal_like((128,), 993)) if newline else None
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    sends = {}                                              # send pointer -> (kind, rank); kind -> [pointer of every rank]
    by_kind = {}
    calls = []

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
    def fake_all_gather(send, recv, count, dtype, comm, stream):
        kind, _ = sends[send]
        assert dtype == 1 and comm == 0x5EED                # ncclUint8, the token handed in as the communicator
        calls.append(kind)
        for r, src in enumerate(by_kind[kind]):
            if hip.hipMemcpyAsync(recv + r * count, src, count, 3, stream) != 0:      # hipMemcpyDeviceToDevice
                return 1
        return 0

    depth = [0]

    @ctypes.CFUNCTYPE(ctypes.c_int)
    def fake_group_start():
        depth[0] += 1
        return 0

    @ctypes.CFUNCTYPE(ctypes.c_int)
    def fake_group_end():
        depth[0] -= 1
        return 0
    addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value
    coll = (addr(fake_all_gather), addr(fake_group_start), addr(fake_group_end), 0x5EED)
    with torch.no_grad():
        want = m(ff, fe, g, "video", nl)
        plans = [hd._shard_plan(m, ff[r * per:(r + 1) * per], fe[r * per:(r + 1) * per], g, T, nl, None, rank=r, world=world, collective=coll)
                 for r in range(world)]
        outs = []
        for r, plan in enumerate(plans):
            st = plan.sets[0]
            assert st.direct_ag and st.tok_direct == (newline is None)
            out = torch.empty((plan.n_rows_total, plan.hidden), dtype=plan.odt, device="cuda")
            plan.set_inputs(st, ff[r * per:(r + 1) * per], fe[r * per:(r + 1) * per], g, out)
            outs.append(out)
            tok_ptr, state_ptr = st.mine.data_ptr() + plan.pack.tok_off, st.mine.data_ptr()
            if st.tok_direct:
                sends[tok_ptr], sends[state_ptr] = ("tokens", r), ("states", r)
                by_kind.setdefault("tokens", []).append(tok_ptr)
                by_kind.setdefault("states", []).append(state_ptr)
            else:
                sends[state_ptr] = ("packed", r)
                by_kind.setdefault("packed", []).append(state_ptr)
            nv_.compressor_fwd(st.a_stream)                 # STREAM phase of rank r (main stream; the comm stream waits for its event)
        torch.cuda.synchronize()                            # (every rank's send buffer is complete: what the collective's own sync guarantees)
        for r, plan in enumerate(plans):
            st = plan.sets[0]
            nv_.compressor_fwd(st.a_finish)                 # FINISH phase of rank r: the double "gathers", the states merge, the tokens land
            if plan.lay.newline_rows:
                first = plan.lay.newline_rows[0]
                step = plan.lay.newline_rows[1] - first if len(plan.lay.newline_rows) > 1 else 1
                nv_.scatter_rows(nl.view(1, -1), outs[r], first, len(plan.lay.newline_rows), row_step=step, stream=plan.comm.cuda_stream)
        torch.cuda.synchronize()
    assert depth[0] == 0 and len(calls) == world * (2 if newline is None else 1)
    for r in range(world):
        assert outs[r].shape == want.shape
        assert float((outs[r].float() - want.float()).abs().max()) <= PATH_TOL, r
    for plan in plans:
        plan.release()


def test_sharded_forward_world1_equals_forward(c2):
    """sharded_forward with a 1-rank RCCL group (STREAM phase -> all-gather -> FINISH phase) reproduces
    the single-call forward at the full C2 size."""
    import socket
    import torch.distributed as dist
    from hicom_amd.dist import sharded_forward
    m, ff, fe, g, _ = c2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        with torch.no_grad():
            want = m(ff, fe, g, "video", None)
            want2 = m(fe, ff, g, "video", None)
            junk = torch.full_like(want, float("nan"))      # the block the allocator hands out next is poisoned
            torch.cuda.synchronize()
            del junk
            got = sharded_forward(m, ff, fe, g, 64)
            torch.cuda.synchronize()
            assert got.shape == want.shape and float((got - want).abs().max()) <= PATH_TOL
            # pipelined form: two buffer sets alternate, results live in the set; every row is rewritten each step
            for k in range(6):
                a, b, w = (ff, fe, want) if k % 3 else (fe, ff, want2)
                o, ev = sharded_forward(m, a, b, g, 64, deferred=True)
                ev.synchronize()
                assert float((o - w).abs().max()) <= PATH_TOL, k
                o.fill_(float("nan"))
            torch.cuda.synchronize()
            # alternating input buffers AND guides, deferred, NO host sync between the calls: every plan
            # owns its workspaces, so a call never streams into partial states a comm stream is still merging
            g2 = (g.float() * -0.7 + 0.05).to(g.dtype)
            want_g2 = m(ff, fe, g2, "video", None)
            want2_g2 = m(fe, ff, g2, "video", None)
            half_a, half_b = ff[:32].contiguous(), fe[:32].contiguous()
            want_half = m(half_a, half_b, g, "video", None)
            outs = []
            for k in range(24):
                a, b = (ff, fe) if k % 2 == 0 else (fe, ff)
                gg = g if (k // 2) % 2 == 0 else g2
                if k % 6 == 5:
                    o, ev = sharded_forward(m, half_a, half_b, g, 32, deferred=True)     # another plan (shape) in between
                    torch.cuda.current_stream().wait_event(ev)
                    outs.append((o.clone(), want_half))
                    continue
                o, ev = sharded_forward(m, a.clone(), b.clone(), gg.clone(), 64, deferred=True)   # fresh buffers every call
                w = {(0, 0): want, (1, 0): want2, (0, 1): want_g2, (1, 1): want2_g2}[k % 2, (k // 2) % 2]
                torch.cuda.current_stream().wait_event(ev)
                outs.append((o.clone(), w))
            torch.cuda.synchronize()
            for k, (o, w) in enumerate(outs):
                assert float((o - w).abs().max()) <= PATH_TOL, ("alternating", k)
    finally:
        dist.destroy_process_group()


def test_release_step_is_bit_stable_over_back_to_back_launches(c2):
    """The release step hands data between workgroups INSIDE launches (query_prep's {epoch, value} granules, the readout GEMMs'
    co-scheduled GEMV roles) and keeps an arrival-counter epoch across launches.  600 joined forwards on three rotating input sets
    and guides, no host synchronisation in between, every result compared bit for bit with the first one of its set: a rare
    stale read or a torn hand-off shows up as a mismatch here (it would pass every single-launch parity test)."""
    m, ff, fe, g, _ = c2
    gen = torch.Generator(device="cuda").manual_seed(17)
    sets = [(ff, fe, g)]
    for _ in range(2):
        sets.append((torch.randn(ff.shape, device="cuda", generator=gen).to(torch.bfloat16),
                     torch.randn(fe.shape, device="cuda", generator=gen).to(torch.bfloat16),
                     torch.randn(g.shape, device="cuda", generator=gen).to(torch.bfloat16)))
    with torch.no_grad():
        want = [m(a, b, c, "video", None).clone() for a, b, c in sets]
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        for i in range(600):
            a, b, c = sets[i % 3]
            bad += (m(a, b, c, "video", None) != want[i % 3]).any()
    assert int(bad) == 0
    # the local-logits variant of the step (frames_embed never read) under the same loop
    from hicom_amd import native as nv
    with torch.no_grad():
        lls = [(b.float() @ c.float()).contiguous() for _, b, c in sets]                 # [T,27,27] raw dot products (test-side torch)
        want_l = [m(a, None, c, "video", None, local_logits=l).clone() for (a, _, c), l in zip(sets, lls)]
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        for i in range(300):
            a, _, c = sets[i % 3]
            bad += (m(a, None, c, "video", None, local_logits=lls[i % 3]) != want_l[i % 3]).any()
    assert int(bad) == 0
    for k in range(3):
        assert float((want_l[k].float() - want[k].float()).abs().max()) <= 2e-2          # bf16 outputs of two summation orders


def test_large_video_local_tokens_are_frame_group_local(c2):
    """256 frames (4x the benchmark; 860 MB of visual tokens): the local tokens of a 4-frame group depend on that group's frames only
    (reference projector.py:544-558), whatever the partition of the windows over workgroups -- the first 1296 rows of the 256-frame
    result equal the 64-frame result on the same frames (to 2e-4: two summation orders), the 32 global rows are 32 copies of one row (direct mode) and differ from
    the 64-frame ones (they see all 256 frames)."""
    m, ff, fe, g, _ = c2
    gen = torch.Generator(device="cuda").manual_seed(29)
    big_ff = torch.cat([ff] + [torch.randn(ff.shape, device="cuda", generator=gen).to(torch.bfloat16) for _ in range(3)])
    big_fe = torch.cat([fe] + [torch.randn(fe.shape, device="cuda", generator=gen).to(torch.bfloat16) for _ in range(3)])
    with torch.no_grad():
        small = m(ff, fe, g, "video", None)
        big = m(big_ff, big_fe, g, "video", None)
    torch.cuda.synchronize()
    assert big.shape == (256 // 4 * 81 + 32, small.shape[1]) and bool(torch.isfinite(big).all())
    # (a window's 36 tokens fall into different 16-token tiles when a workgroup owns 21 windows instead of 6: the online softmax sums
    # in another order -- an fp32 difference of ~1e-7 that the fp16 activation plane of the readout can turn into ~5e-5)
    assert float((big[:1296].float() - small[:1296].float()).abs().max()) <= 2e-4
    assert all(torch.equal(big[-32], big[-32 + k]) for k in range(1, 32))
    assert not torch.equal(big[-32:], small[-32:])


@pytest.mark.parametrize("recipe,world,newline", [("coarse", 2, None), ("coarse", 4, "grid"), ("fine", 2, "frame"), ("fine", 4, None),
                                                  ("adaptqg_coarse", 2, None), ("clip_local_coarse", 4, None),
                                                  ("clip_global_direct", 4, None), ("clip_global_off", 2, "grid")])
def test_sharded_stepwise_recipes_emulated_on_one_gpu(recipe, world, newline):
    """Round 5 (verdict r4 #6): coarse / fine injection and the query-side adaptors shard over frames, operator by operator
    (`dist.stepwise_shard_send` / `stepwise_shard_finish`, the two halves of `sharded_forward_stepwise` around its one all-gather).
    2- and 4-rank worlds on one GPU at the 27 x 27 grid -- the all-gather replaced by stacking the ranks' send buffers -- against the
    ORACLE (reference projector.py:369-397 injectors, :539-542 pooled window queries, absolute frame indices in the positional terms)
    and against the unsharded HIP forward; with and without newline rows in the packed output."""
    from types import SimpleNamespace
    from hicom_amd import dist as hd, synth
    from oracle import hicom_oracle as orc
    T, H, W = 16, 27, 27
    over = {"hidden_size": 128, "max_num_frames": 32}
    if recipe in ("coarse", "fine"):
        over.update(mm_projector_type="local43_global32", use_guide=recipe)
    elif recipe == "adaptqg_coarse":
        over.update(mm_projector_type="local43_adaptqg_global32_adaptg", use_guide="coarse")
    elif recipe == "clip_global_direct":                       # round 6: clip-scale on the GLOBAL stage shards too (key norms are per token)
        over.update(mm_projector_type="local43_global32", use_guide="direct", use_clip_scale="global")
    elif recipe == "clip_global_off":
        over.update(mm_projector_type="local43_global32", use_guide=None, use_clip_scale="global")
    else:
        over.update(mm_projector_type="local43_global32", use_guide="coarse", use_clip_scale="local")
    if newline:
        over.update(mm_patch_merge_type="spatial_unpad", mm_newline_position=newline)
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, **over})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag="shard_" + recipe)
    m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
    x = synth.synth_inputs(T, H, W, 1152, tag="shard_" + recipe, guide_len=64 if recipe == "fine" else 0)
    ff, fe, g = dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"])
    nl = dev_bf16(synth.normal_like((128,), 991)) if newline else None
    logit = None
    if recipe == "clip_local_coarse":
        logit = (2.0, -3.0)
        m.set_clip_logits(local=logit)
    if recipe in ("clip_global_direct", "clip_global_off"):
        logit = (1.5, -2.0)
        m.set_clip_logits(glob=logit)
    per = T // world
    with torch.no_grad():
        want = m(ff, fe, g, "video", nl)
        sends = []
        for r in range(world):
            lay, mine, meta = hd.stepwise_shard_send(m, ff[r * per:(r + 1) * per], fe[r * per:(r + 1) * per], g, T, r, world)
            sends.append(mine)
        everyone = torch.stack(sends)
        got = hd.stepwise_shard_finish(m, lay, everyone, meta, world, nl)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    assert float((got.float() - want.float()).abs().max()) <= PATH_TOL
    if logit is None:
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        ref = orc.projector_forward(cfg, sdt, ff.float().cpu(), fe.float().cpu(), g.float().cpu(), "video",
                                    None if nl is None else nl.float().cpu())
        assert float((got.float().cpu() - ref).abs().max()) <= TOL


@pytest.mark.parametrize("ptype,mode,guide_len,T,h,w,modal,newline", [
    ("local43", "coarse", 0, 4, 6, 6, "video", False),            # local stage alone: the call's local chain injects
    ("local43", "fine", 17, 8, 9, 6, "video", False),             # fewer than 64 text tokens
    ("global32", "fine", 64, 4, 6, 6, "video", False),            # global stage alone
    ("global16", "coarse", 0, 3, 5, 7, "video", False),
    ("local43_global32", "coarse", 0, 1, 6, 9, "image", True),    # image modal with the newline token (T = 1: the logit-tensor form of the merge)
    ("local43_global32", None, 0, 1, 9, 6, "image", False),       # guide off on a single image (found broken by the case above: round 5 fix)
    ("local43_global32", None, 0, 2, 27, 27, "video", False),
    ("local22_global8", "fine", 40, 6, 8, 8, "video", False)])
def test_in_call_injection_shapes_against_the_oracle(ptype, mode, guide_len, T, h, w, modal, newline):
    """Round 5: coarse / fine injection run by hicom_compressor_fwd itself (`inj_l` / `inj_g`) on configurations the golden set does not
    hold: one stage alone, short guides, image inputs with the newline token, other window / query geometries -- against the pinned
    oracle (tolerance of the golden cases) and against the operator-by-operator path."""
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    tag = f"inj:{ptype}:{mode}:{T}:{h}:{w}"
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": ptype, "use_guide": mode,
                             "mm_newline_position": "grid" if newline else cases.DEFAULT_CFG.get("mm_newline_position", "no_token")})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
    x = synth.synth_inputs(T, h, w, cases.D, tag=tag, guide_len=guide_len)
    nl_np = synth.normal_like((cfg.hidden_size,), synth.seed_of(tag + ":newline")) if newline else None
    case = SimpleNamespace(cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"], modal=modal, newline=nl_np, anyres=None, logit=None)
    m = build_module(case)
    assert m._executor_covers() and not m._external_queries()
    ff, fe, g, nl = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), dev_bf16(case.newline)
    with torch.no_grad():
        got = m(ff, fe, g, modal, nl)
        step = m.forward_stepwise(ff, fe, g, modal, nl)
    want = run_oracle(case)["out"].numpy()
    assert got.shape == step.shape == tuple(want.shape)
    assert float((got - step).abs().max()) <= PATH_TOL
    assert np.abs(got.float().cpu().numpy() - want).max() <= TOL


@pytest.mark.parametrize("seed,large,count,need", [(20251004, False, 28, 18), (7, False, 28, 16), (99, True, 10, 6)])
def test_random_configuration_sweep_against_the_oracle(seed, large, count, need):
    """Seeded sweep over (geometry, projector type, injection mode, modal, newline position): the drop-in forward against the pinned
    oracle on configurations nobody wrote down -- single images, tiny grids, odd frame counts, one stage alone.  Where the oracle
    (= the reference's behaviour) raises, the HIP path must raise too; everywhere else: same shape, max-abs <= 1e-3."""
    import random
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    rng = random.Random(seed)
    ptypes = ["local43_global32", "local43_global32", "local22_global8", "local41_global16", "local43", "global32", "local23_global4"]
    ran = raised = 0
    for k in range(count):
        ptype = rng.choice(ptypes)
        mode = rng.choice(["direct", None, "coarse", "fine", "direct"])
        modal = rng.choice(["video", "video", "image"])
        T = 1 if modal == "image" else rng.choice([1, 4, 8, 16] if large else [1, 2, 3, 4, 5, 8, 12])
        h, w = (rng.choice([9, 14, 27]), rng.choice([9, 18, 27])) if large else (rng.choice([2, 3, 4, 6, 7, 9, 12]), rng.choice([2, 3, 5, 6, 8, 9]))
        nlpos = rng.choice(["no_token", "grid", "frame", "one_token"])
        merge = rng.choice(["spatial_unpad", "flat"])
        glen = rng.choice([5, 33, 64]) if mode == "fine" else 0
        tag = f"sweep{seed}:{k}" if seed != 20251004 else f"sweep{k}"
        cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": ptype, "use_guide": mode, "mm_newline_position": nlpos,
                                 "mm_patch_merge_type": merge})
        sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
        x = synth.synth_inputs(T, h, w, cases.D, tag=tag, guide_len=glen)
        nl_np = synth.normal_like((cfg.hidden_size,), synth.seed_of(tag + ":newline"))
        case = SimpleNamespace(cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"], modal=modal, newline=nl_np, anyres=None, logit=None)
        what = (k, ptype, mode, modal, T, h, w, nlpos, merge, glen)
        try:
            want = run_oracle(case)["out"].numpy()
        except Exception as e:                                        # the reference refuses this geometry (e.g. 5 frames under a 4-frame kernel)
            m = build_module(case)
            with pytest.raises(Exception):
                with torch.no_grad():
                    m(dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), modal, dev_bf16(case.newline))
            raised += 1
            continue
        m = build_module(case)
        with torch.no_grad():
            got = m(dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), modal, dev_bf16(case.newline))
        assert tuple(got.shape) == tuple(want.shape), what
        err = float(np.abs(got.float().cpu().numpy() - want).max())
        assert err <= TOL, (what, err)
        ran += 1
    assert ran >= need, (ran, raised)


@pytest.mark.parametrize("hidden,T,mode,fp32_out", [(64, 4, "direct", True), (128, 8, None, False), (1024, 4, "direct", True), (1536, 8, "direct", False),
                                                     (2048, 16, "direct", False), (2048, 32, None, True), (3584, 32, "direct", False), (3584, 8, "coarse", False),
                                                     (3584, 64, "direct", True), (4096, 32, "direct", False)])
def test_hidden_width_sweep_against_the_oracle(hidden, T, mode, fp32_out):
    """LLM widths other than the benchmark's 896 on the real 27 x 27 grid: every readout form (fp16 planes with 64- and 128-column
    tiles, the chain role up to 1536, the GEMV role above), bf16 and fp32 rows.  (Widths that are no multiple of 64 are refused by
    the one-call executor with a message -- no LLM has one.)"""
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    tag = f"width:{hidden}:{T}:{mode}"
    cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "hidden_size": hidden, "use_guide": mode, "max_num_frames": max(T, 32)})
    sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
    x = synth.synth_inputs(T, 27, 27, cases.D, tag=tag)
    case = SimpleNamespace(cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"], modal="video", newline=None, anyres=None, logit=None)
    m = build_module(case, fp32_out=fp32_out)
    with torch.no_grad():
        got = m(dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), "video", None)
        again = m(dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), "video", None)
    want = run_oracle(case)["out"].numpy()
    assert tuple(got.shape) == tuple(want.shape) and got.dtype == (torch.float32 if fp32_out else torch.bfloat16)
    assert torch.equal(got, again)
    tol = TOL if fp32_out else TOL + 2.0 ** -8 * float(np.abs(want).max())          # (+ the bf16 rounding of the rows)
    assert float(np.abs(got.float().cpu().numpy() - want).max()) <= tol


def test_random_anyres_sweep_against_the_oracle():
    """Seeded sweep over anyres dict inputs (reference projector.py:679-689): base image (or none) + patch grid of another size, every
    injection mode, both merge types -- the operator-by-operator path against the pinned oracle."""
    import random
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    rng = random.Random(4242)
    ran = 0
    for k in range(20):
        ptype = rng.choice(["local43_global32", "local22_global8", "local43", "local43_global32", "global32"])
        mode = rng.choice(["direct", None, "coarse", "fine"])
        h, w = rng.choice([3, 6, 9]), rng.choice([3, 6, 9])
        ph, pw = rng.choice([3, 6, 9, 12]), rng.choice([6, 9, 12, 18])
        no_base = rng.random() < 0.3
        glen = rng.choice([7, 64]) if mode == "fine" else 0
        merge = rng.choice(["spatial_unpad", "flat"])
        tag = f"anyres{k}"
        cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": ptype, "use_guide": mode, "mm_patch_merge_type": merge})
        sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
        x = synth.synth_inputs(1, h, w, cases.D, tag=tag, guide_len=glen)
        pz = synth.synth_inputs(1, ph, pw, cases.D, tag=tag + ":patch")
        nl_np = synth.normal_like((cfg.hidden_size,), synth.seed_of(tag + ":newline"))
        case = SimpleNamespace(cfg=cfg, sd=sd, ff=x["ff"], fe=x["fe"], g=x["g"], modal="image", newline=nl_np, logit=None,
                               anyres=dict(patch_ff=pz["ff"][0], patch_fe=pz["fe"][0], no_base=no_base))
        what = (k, ptype, mode, h, w, ph, pw, no_base, merge)
        try:
            want = run_oracle(case)["out"].numpy()
        except Exception:
            continue
        got = run_native(case)["out"].float().cpu().numpy()
        assert got.shape == want.shape, what
        assert float(np.abs(got - want).max()) <= TOL, what
        ran += 1
    assert ran >= 9, ran


@pytest.mark.parametrize("name", ["G2_off_T8", "G6_coarse", "G7_fine", "G5b_adaptqkvg_off", "G5_adaptkv"])
def test_graph_replay_of_the_other_recipes_equals_eager(name):
    """hipGraph capture / replay of the plans of the non-release recipes: guide off (folded queries kept across calls), coarse / fine
    (injector chains inside the C call), query adaptors (their producers are captured in front of the call), k / v adaptors -- the
    replayed result follows the inputs of each call and equals the eager one bit for bit."""
    case = cases.build_case(name)
    m = build_module(case)
    ff, fe, g = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g)
    ff2 = (ff.float() * 0.5 + 0.1).to(ff.dtype)
    with torch.no_grad():
        want, want2 = m(ff, fe, g, case.modal, None).clone(), m(ff2, fe, g, case.modal, None).clone()
        m.graph_replay = True
        try:
            a, b, c = m(ff, fe, g, case.modal, None), m(ff2, fe, g, case.modal, None), m(ff, fe, g, case.modal, None)
            a2 = m(ff, fe, g, case.modal, None)
            torch.cuda.synchronize()
        finally:
            m.graph_replay = False
            m._invalidate_plans()
    assert torch.equal(a, want) and torch.equal(c, want) and torch.equal(a2, want) and torch.equal(b, want2)
    assert not torch.equal(want, want2)


def test_random_shard_emulation_sweep():
    """Seeded sweep of the frame-sharded device path (see _emulate_ranks) over world sizes, frames per rank, grids, hidden widths and the
    guide-off / direct / k-v-adaptor recipes the executor's STREAM / FINISH phases run: every emulated world against the dense forward
    of all frames."""
    import random
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    rng = random.Random(77)
    for k in range(10):
        world = rng.choice([2, 3, 4, 8])
        per = 4 * rng.choice([1, 2, 4])
        h, w = rng.choice([(27, 27), (9, 9), (6, 12), (27, 27)])
        ptype, mode = rng.choice([("local43_global32", "direct"), ("local43_global32", None), ("local43_adaptkv_global32", "direct"),
                                  ("local43_global32", "direct")])
        hidden = rng.choice([64, 896, 3584]) if (h, w) == (27, 27) else 64
        T = world * per
        tag = f"shard{k}"
        cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": ptype, "use_guide": mode, "hidden_size": hidden,
                                 "max_num_frames": max(T, 32)})
        sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
        m = build_module(SimpleNamespace(cfg=cfg, sd=sd))
        gen = torch.Generator(device="cuda").manual_seed(500 + k)
        ff = torch.randn(T, h, w, 1152, device="cuda", generator=gen).bfloat16()
        fe = torch.randn(T, h, w, 1152, device="cuda", generator=gen).bfloat16()
        g = torch.randn(1152, device="cuda", generator=gen).bfloat16()
        with torch.no_grad():
            want = m(ff, fe, g, "video", None)
            got = _emulate_ranks(m, ff, fe, g, world, rng.randrange(world))
        what = (k, world, per, h, w, ptype, mode, hidden)
        assert got.shape == want.shape and bool(torch.isfinite(got).all()), what
        assert float((got - want).abs().max()) <= PATH_TOL * max(1.0, float(want.abs().max())), (what, float((got - want).abs().max()))


def test_random_clip_scale_sweep_against_the_oracle():
    """Seeded sweep of the clip-scale variants (reference projector.py:184-191, :527-529, :549) through HIComProjector.forward: which
    stages use the SigLIP logits, their values, injection mode, adaptors, geometry -- against the oracle's projector forward."""
    import random
    import hicom_amd
    from types import SimpleNamespace
    from hicom_amd import synth
    from oracle import hicom_oracle as orc
    from oracle_util import to_t
    rng = random.Random(909)
    ran = 0
    for k in range(16):
        ptype = rng.choice(["local43_global32", "local43_global32", "local22_global8", "local43_adaptkv_global32", "local43", "global32"])
        mode = rng.choice(["direct", None, "coarse", "fine", "direct"])
        which = rng.choice(["local", "global", "local,global"])
        T, h, w = rng.choice([1, 4, 8]), rng.choice([3, 6, 9]), rng.choice([3, 6, 9])
        glen = rng.choice([5, 64]) if mode == "fine" else 0
        q16 = lambda lo, hi: round(rng.uniform(lo, hi) * 16) / 16            # (bf16-representable: the module keeps its logit parameters in bf16)
        logit = {"local": (q16(0.5, 2.5), q16(-4, 1)), "global": (q16(0.5, 2.5), q16(-4, 1))}
        tag = f"clip{k}"
        cfg = SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_projector_type": ptype, "use_guide": mode, "use_clip_scale": which})
        sd = synth.synth_state_dict(orc.param_shapes(cfg), tag=tag)
        x = synth.synth_inputs(T, h, w, cases.D, tag=tag, guide_len=glen)
        what = (k, ptype, mode, which, T, h, w)
        used = {s: tuple(torch.tensor(v) for v in logit[s]) for s in ("local", "global") if s in which.split(",")}
        try:
            want = orc.projector_forward(cfg, {n: to_t(v) for n, v in sd.items()}, to_t(x["ff"]), to_t(x["fe"]), to_t(x["g"]), "video", None,
                                         logit=used).numpy()
        except Exception:
            continue
        m = hicom_amd.build_vision_projector(cfg)
        m.load_state_dict({n: torch.from_numpy(v.copy()) for n, v in sd.items()}, strict=False)
        m = m.to(torch.bfloat16).cuda().eval()
        m.return_fp32 = True
        for sub in (m.local_compressor, m.global_compressor):
            if sub is not None:
                sub.return_fp32 = True
        m.set_clip_logits(local=logit["local"] if "local" in which else None, glob=logit["global"] if "global" in which else None)
        with torch.no_grad():
            got = m(dev_bf16(x["ff"]), dev_bf16(x["fe"]), dev_bf16(x["g"]), "video", None).float().cpu().numpy()
        assert got.shape == want.shape, what
        assert float(np.abs(got - want).max()) <= TOL, (what, float(np.abs(got - want).max()))
        ran += 1
    assert ran >= 10, ran


@pytest.mark.parametrize("name", ["G9_anyres", "G9_anyres_nobase"])
def test_anyres_dict_through_the_executor_equals_stepwise(name):
    """Round 5: the anyres dict input takes hicom_compressor_fwd once per segment (base image: local stage; patch grid: local stage with
    the anyres packing + global stage, rows behind the base image's) instead of one C call per operator: same rows (<= PATH_TOL; the
    one-call form carries its activations as fp16 planes), same bits from call to call, and a second image of another size on the
    same module."""
    case = cases.build_case(name)
    m = build_module(case)
    a = case.anyres
    ff, fe, g, nl = dev_bf16(case.ff), dev_bf16(case.fe), dev_bf16(case.g), dev_bf16(case.newline)
    fd = {"base": None if a["no_base"] else ff[0], "patch": dev_bf16(a["patch_ff"])}
    ed = {"base": None if a["no_base"] else fe[0], "patch": dev_bf16(a["patch_fe"])}
    with torch.no_grad():
        got = m(fd, ed, g, case.modal, nl)
        step = m.forward_stepwise(fd, ed, g, case.modal, nl)
        again = m(fd, ed, g, case.modal, nl)
        fd2 = {"base": fd["base"], "patch": fd["patch"][:6, :3].contiguous()}
        ed2 = {"base": ed["base"], "patch": ed["patch"][:6, :3].contiguous()}
        other = m(fd2, ed2, g, case.modal, nl)
        other_step = m.forward_stepwise(fd2, ed2, g, case.modal, nl)
    assert len(m.__dict__.get("_engine_plans", {})) >= 2
    assert got.shape == step.shape and float((got - step).abs().max()) <= PATH_TOL and torch.equal(got, again)
    assert other.shape == other_step.shape and float((other - other_step).abs().max()) <= PATH_TOL
