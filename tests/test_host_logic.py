"""CPU tests of the host-side logic (no GPU): geometry vs the oracle's independent restatement,
state-dict schema, type-string grammar, packing layouts, C-ABI exports."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cases
import hicom_amd
from hicom_amd import geometry as geo
from hicom_amd import native, synth
from oracle import hicom_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 13])
def test_axis_tiling_matches_oracle_rule(k):
    for n in range(1, 70):
        try:
            starts, klen = orc.window_starts(n, k)
        except RuntimeError:
            with pytest.raises(RuntimeError):
                geo.axis_tiling(n, k)
            continue
        a = geo.axis_tiling(n, k)
        assert a.starts == starts and a.k == klen and a.nwin == len(starts)
        assert a.starts[-1] + a.k <= n and a.starts[0] == 0


def test_known_overlap_cases():
    assert geo.axis_tiling(7, 4).starts == [0, 3]            # frames {0..3},{3..6}
    assert geo.axis_tiling(27, 2).starts[-1] == 25            # local22 on 27: last window overlaps by one
    for t in (5, 6, 9):
        with pytest.raises(RuntimeError):
            geo.axis_tiling(t, 4)
    assert geo.axis_tiling(3, 4) == geo.AxisTiling(3, 3, 1, 1)


@pytest.mark.parametrize("s", ["local43_global32_coarse", "local43_adaptkv_global32", "local22_global16",
                               "local413_global8", "local43_guidecoarse_global32_guideoff",
                               "local43_adaptqkvg_global32_adaptg", "global32", "local43"])
def test_type_string_grammar_matches_oracle(s):
    l, g = geo.parse_mm_projector_type(s)
    o = orc.parse_projector_type(s)
    if o["local"] is None:
        assert l is None
    else:
        ol = o["local"]
        assert (l.temporal_kernel_size, l.spatial_kernel_size) == (ol["kt"], ol["ks"])
        assert (l.adapt_q, l.adapt_k, l.adapt_v, l.adapt_guide) == (ol["adapt_q"], ol["adapt_k"], ol["adapt_v"], ol["adapt_guide"])
        assert l.force_use_guide == ol["force_use_guide"]
    if o["global"] is None:
        assert g is None
    else:
        assert (g.num_queries, g.adapt_guide, g.force_use_guide) == (o["global"]["num_queries"], o["global"]["adapt_guide"], o["global"]["force_use_guide"])


def test_pack_layout_row_counts():
    # SURVEY §8a8 probe: [2,9,9,*] -> 180 / 164 / 163 / 162 rows
    for pos, rows in (("grid", 180), ("frame", 164), ("one_token", 163), ("no_token", 162)):
        lay = geo.pack_layout("spatial_unpad", pos, "video", 2, 9, 9, True, False)
        assert lay.n_rows == rows
        packed = sorted(set(lay.row_of(m) for m in range(lay.n_tokens)) | set(lay.newline_rows))
        assert packed == list(range(rows))
    assert geo.pack_layout("flat", "grid", "video", 2, 9, 9, False, False).n_rows == 162
    assert geo.pack_layout("spatial", "grid", "image", 1, 9, 9, True, True).n_rows == 90
    assert geo.pack_layout("spatial", "grid", "image", 1, 9, 9, True, False).n_rows == 82
    with pytest.raises(ValueError):
        geo.pack_layout("spatial", "grid", "image", 2, 9, 9, True, False)


def test_pack_layout_matches_oracle_rows():
    d = 4
    for pos in ("grid", "frame", "one_token", "no_token"):
        cfg = SimpleNamespace(mm_patch_merge_type="spatial_unpad", mm_newline_position=pos)
        feat = torch.arange(2 * 3 * 5 * d, dtype=torch.float32).reshape(2, 3, 5, d)
        nl = torch.full((d,), -1.0)
        want = orc.post_process(cfg, feat, "video", nl, False)
        lay = geo.pack_layout("spatial_unpad", pos, "video", 2, 3, 5, True, False)
        got = torch.zeros(lay.n_rows, d)
        flat = feat.reshape(-1, d)
        for m in range(lay.n_tokens):
            got[lay.row_of(m)] = flat[m]
        for r in lay.newline_rows:
            got[r] = nl
        assert torch.equal(got, want)


def test_pos_tables_match_oracle():
    T, H, W, D = 5, 4, 3, 1152
    tab = geo.stacked_pos_tables(8, H, W, D)
    full = orc.pos_table(T, H, W, D).numpy()
    mine = tab[:T, None, None, :] + tab[8:8 + H][None, :, None, :] + tab[8 + H:8 + H + W][None, None, :, :]
    assert np.abs(mine - full).max() < 5e-7


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if not c.get("expect_raises")])
def test_state_dict_schema(name):
    cfg = SimpleNamespace(**cases.case_config(name))
    m = hicom_amd.build_vision_projector(cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == orc.param_shapes(cfg)
    m.load_state_dict({k: torch.zeros(s) for k, s in got.items()}, strict=True)
    m.to(torch.bfloat16)
    m.requires_grad_(False)


def test_init_law_and_plain_projectors():
    cfg = SimpleNamespace(**cases.DEFAULT_CFG)
    m = hicom_amd.build_vision_projector(cfg)
    assert float(m.global_compressor.query.detach().abs().max()) == 0.0          # zero-init (ref :583)
    w = m.local_compressor.readout[0].weight
    assert 0.015 < float(w.std()) < 0.025 and float(m.local_compressor.readout[0].bias.abs().max()) == 0
    assert isinstance(hicom_amd.build_vision_projector(SimpleNamespace(mm_projector_type="linear", mm_hidden_size=8, hidden_size=4)), torch.nn.Linear)
    seq = hicom_amd.build_vision_projector(SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=8, hidden_size=4))
    assert [type(x).__name__ for x in seq] == ["Linear", "GELU", "Linear"]
    with pytest.raises(NotImplementedError):
        hicom_amd.build_vision_projector(SimpleNamespace(**{**cases.DEFAULT_CFG, "mm_vision_tower": "other"}))


def test_cpu_tensors_fail_loudly():
    """No CPU fallback: the product path refuses CPU tensors instead of computing with torch."""
    case = cases.build_case("G1_direct_T8")
    m = hicom_amd.build_vision_projector(case.cfg).to(torch.bfloat16)
    t = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    with pytest.raises(native.HicomNativeError):
        m(t(case.ff), t(case.fe), t(case.g), "video", None)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hicom_hip.h")).read()
    declared = set(re.findall(r"\b(hicom_[a-z0-9_]+)\s*\(", header))
    assert declared == set(native.EXPORTS)
    lib = ctypes.CDLL(native.LIB_PATH)       # loading needs no GPU; no compute call is made
    for name in declared:
        assert hasattr(lib, name), name
    lib.hicom_abi_version.restype = ctypes.c_int
    assert lib.hicom_abi_version() == native.ABI_VERSION


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hicom_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|import_module\([\"']oracle|oracle/", src, re.M), f


def test_synth_is_bit_stable():
    a = synth.normal_like((4, 5), 123)
    assert a.dtype == np.float32 and np.array_equal(a, synth.round_to_bf16(a))
    assert abs(float(synth.normal_like((200000,), 7).std()) - 1.0) < 0.02


def test_weights_epoch_protocol():
    """Training forwards bump the weights epoch and leave a dirty mark; the first inference entry point afterwards bumps once
    more (tables built between a forward and the optimizer step must not survive it); later inference calls do not."""
    from hicom_amd import native as nv
    nv.begin_inference()                     # (consume a mark an earlier test's training-mode call may have left)
    e0 = nv.weights_epoch()
    nv.begin_inference()
    assert nv.weights_epoch() == e0
    nv.note_training_forward()
    assert nv.weights_epoch() == e0 + 1
    nv.note_training_forward()
    assert nv.weights_epoch() == e0 + 2
    nv.begin_inference()
    assert nv.weights_epoch() == e0 + 3
    nv.begin_inference()
    assert nv.weights_epoch() == e0 + 3
    nv.invalidate_weight_caches()
    assert nv.weights_epoch() == e0 + 4


def test_data_alias_reads_do_not_move_the_weights_epoch_and_writes_do():
    """Round-5 verdict #8 / ADVICE r5: `p.data` READS (norm loggers, EMA, deepcopy, DeepSpeed bookkeeping) must not invalidate the
    weight-derived tables; WRITES through the alias (`p.data.copy_`, an in-place op on a held alias or on a view of it, `p.data = w`)
    must -- the alias's own version counter tells them apart (native.TrackedParameter, native._scan_data_aliases)."""
    import copy
    from hicom_amd import native as nv
    lin = torch.nn.Linear(4, 4)
    nv.track_parameters(lin)
    assert type(lin.weight) is nv.TrackedParameter
    e0 = nv.weights_epoch()
    for _ in range(50):
        lin.weight.data.norm()
        lin.bias.data.float().sum()
    copy.deepcopy(lin)
    assert nv.weights_epoch() == e0 and len(nv._DATA_ALIASES) == 0          # reads: nothing moved, nothing hoarded
    lin.weight.data.copy_(torch.zeros(4, 4))
    assert nv.weights_epoch() == e0 + 1
    held = lin.weight.data
    assert nv.weights_epoch() == e0 + 1 and len(nv._DATA_ALIASES) == 1      # a held alias stays watched
    held.mul_(2)
    assert nv.weights_epoch() == e0 + 2
    assert nv.weights_epoch() == e0 + 2                                     # counted once
    view = lin.bias.data[1:3]
    view.zero_()                                                             # a view shares the alias's version counter
    assert nv.weights_epoch() == e0 + 3
    del held, view
    assert nv.weights_epoch() == e0 + 3 and len(nv._DATA_ALIASES) == 0
    lin.weight.data = torch.ones(4, 4)
    assert nv.weights_epoch() == e0 + 4


def test_device_event_factory_has_the_surface_the_callers_use():
    """hicom_amd/events.py: a stream-to-stream event is either the raw HIP event (hipEventDisableSystemFence) or -- no HIP device, no library by that
    name, HICOM_EVENT_NOFENCE=0 -- a torch event; both expose record / wait / cuda_event (engine._DeviceResources, dist._shard_plan,
    projector._two_stream_forward)."""
    import importlib
    from hicom_amd import events
    ev = events.device_event()
    assert hasattr(ev, "record") and hasattr(ev, "wait")
    os.environ["HICOM_EVENT_NOFENCE"] = "0"
    try:
        importlib.reload(events)
        import torch
        assert isinstance(events.device_event(), torch.cuda.Event)
    finally:
        os.environ.pop("HICOM_EVENT_NOFENCE", None)
        importlib.reload(events)
