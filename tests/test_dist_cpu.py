"""World-size-2 gloo test (CPU) of the frame-shard plan and the single-collective exchange.

The HIP kernels cannot run here, so each rank's shard results are produced by the oracle (as the
checker's stand-in for the kernels); what is under test is the product's plan, buffer packing,
all-gather ordering and the online-softmax combine identity the multi-GPU path relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from hicom_amd.dist import ExchangeSets, FrameShardPlan, PackLayout, exchange, gather_packed
from oracle import hicom_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_state(sd, ff, guide, t0, T_total, nh=9):
    """(M, L, ACC) of one frame shard for the direct-mode query, from oracle math (fp64)."""
    E = ff.shape[-1]
    H, W = ff.shape[1:3]
    pos = orc.pos_table(T_total, H, W, E)[t0:t0 + ff.shape[0]].double()
    x = (ff.double() + pos).reshape(-1, E)
    pre = "global_compressor.attn_layer"
    q = orc.linear(guide.double()[None], {k: v.double() for k, v in sd.items()}, pre + ".q_proj")
    k = orc.linear(x, {k_: v.double() for k_, v in sd.items()}, pre + ".k_proj")
    hd = E // nh
    s = torch.einsum("hd,nhd->hn", q.reshape(nh, hd), k.reshape(-1, nh, hd)) * hd ** -0.5
    M = s.max(dim=1).values
    p = torch.exp(s - M[:, None])
    return M, p.sum(dim=1), p @ x        # [nh], [nh], [nh, E]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = cases.build_case("G1_direct_T8")
        sd = {k: torch.from_numpy(v) for k, v in case.sd.items()}
        ff, fe, g = (torch.from_numpy(a) for a in (case.ff, case.fe, case.g))
        T = ff.shape[0]
        plan = FrameShardPlan(T, world, 4)
        t0, t1 = plan.frame_range(rank)
        spec = orc.parse_projector_type(case.cfg.mm_projector_type)
        loc = orc.local_forward(spec["local"], "direct", sd, "local_compressor", ff[t0:t1], fe[t0:t1], g, "video")
        loc = loc.reshape(-1, loc.shape[-1]).float()
        assert loc.shape[0] == plan.windows_per_rank(2, 2)
        M, L, ACC = _shard_state(sd, ff[t0:t1], g, t0, T)
        state = torch.cat([torch.stack([M, L], dim=1).reshape(-1), ACC.reshape(-1)]).float()
        states, tokens = exchange(state, loc)
        assert states.shape == (world, state.numel()) and tokens.shape == (world * loc.shape[0], loc.shape[1])
        assert torch.equal(states[rank], state) and torch.equal(tokens[rank * loc.shape[0]:(rank + 1) * loc.shape[0]], loc)
        # the same layout object drives the HIP path: its kernels write the send buffer in place and read the gathered one
        # at these offsets (state_set_stride / place_src / place_block_stride in hicom_amd/dist.py)
        lay = PackLayout(state.numel(), loc.shape[0], loc.shape[1], 2)
        assert lay.tok_off % 16 == 0 and lay.total % 16 == 0 and lay.set_stride_floats * 4 == lay.total
        mine = lay.new_buffer("cpu")
        lay.state_view(mine).copy_(state)
        lay.tokens_view(mine, torch.bfloat16).copy_(loc.bfloat16())
        everyone = gather_packed(mine, lay.new_buffer("cpu", world))
        flat = everyone.view(-1)
        for r in range(world):             # what hicom_global_combine_strided_fwd / hicom_place_blocks_fwd address
            st_r = flat[r * lay.total: r * lay.total + lay.state_bytes].view(torch.float32)
            tk_r = flat[r * lay.total + lay.tok_off: r * lay.total + lay.tok_off + lay.tok_bytes].view(torch.bfloat16)
            assert torch.equal(st_r, states[r]) and torch.equal(tk_r.view(loc.shape), tokens[r * loc.shape[0]:(r + 1) * loc.shape[0]].bfloat16())
        # combine exactly as hicom_global_combine_fwd does
        R, E = ACC.shape
        ml = states[:, :2 * R].reshape(world, R, 2).double()
        acc = states[:, 2 * R:].reshape(world, R, E).double()
        Ms = ml[..., 0].max(dim=0).values
        w = torch.exp(ml[..., 0] - Ms)
        ctx = (w[..., None] * acc).sum(0) / (w * ml[..., 1]).sum(0)[:, None]
        # finish the global rows with oracle math and compare with the unsharded oracle
        dsd = {k: v.double() for k, v in sd.items()}
        pre = "global_compressor.attn_layer"
        hd = E // R
        o = torch.cat([dsd[pre + ".v_proj.weight"][h * hd:(h + 1) * hd] @ ctx[h] for h in range(R)]) + dsd[pre + ".v_proj.bias"]
        att = orc.linear(o[None], dsd, pre + ".out_proj")
        glob = orc.mlp2(g.double()[None] + att, dsd, "global_compressor.readout")
        full = orc.projector_forward(case.cfg, sd, ff, fe, g, "video", None)
        assert float((tokens - full[:tokens.shape[0]]).abs().max()) < 1e-5
        assert float((glob.float() - full[-1:]).abs().max()) < 1e-5
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _worker_two_steps(rank, world, port, tmp):
    """Three consecutive steps through the plan's TWO exchange-buffer pairs (`ExchangeSets`, the object sharded_forward steps
    through): step i gathers into pair i mod 2 while the consumer of step i - 1 still holds VIEWS into the other pair -- what the
    GPU path's deferred FINISH phase does.  Checks the alternation, that a step never disturbs the previous step's gathered
    data, and that pair 0 is reused (and correctly refilled) by step 2."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, rows, cols = 2 * 9 + 9 * 64, 6, 32
        lay = PackLayout(S, rows, cols, 2)
        xs = ExchangeSets(lay, "cpu", world)
        assert len(xs.pairs) == 2

        def payload(step, r):
            g = torch.Generator().manual_seed(1000 * step + r)
            return torch.randn(S, generator=g), torch.randn(rows, cols, generator=g).bfloat16()

        held = []                                     # (step, index, state view, token view) of earlier steps: views, not copies
        for step in range(3):
            i = xs.advance()
            assert i == step % 2
            mine, everyone = xs.pairs[i]
            st, tk = payload(step, rank)
            lay.state_view(mine).copy_(st)
            lay.tokens_view(mine, torch.bfloat16).copy_(tk)
            gather_packed(mine, everyone)
            flat = everyone.view(-1)
            views = []
            for r in range(world):                    # the offsets the FINISH-phase kernels address (state_set_stride, place_src)
                st_r = flat[r * lay.total: r * lay.total + lay.state_bytes].view(torch.float32)
                tk_r = flat[r * lay.total + lay.tok_off: r * lay.total + lay.tok_off + lay.tok_bytes].view(torch.bfloat16).view(rows, cols)
                want_st, want_tk = payload(step, r)
                assert torch.equal(st_r, want_st) and torch.equal(tk_r, want_tk)
                views.append((st_r, tk_r))
            # the previous step's gathered data (other pair) is untouched by this step's fill + collective
            for (pstep, pi, pviews) in held[-1:]:
                assert pi != i
                for r in range(world):
                    want_st, want_tk = payload(pstep, r)
                    assert torch.equal(pviews[r][0], want_st) and torch.equal(pviews[r][1], want_tk)
            held.append((step, i, views))
        # step 2 reused pair 0: step 0's views now show step 2's data (the documented lifetime of a deferred result: two steps)
        assert held[0][1] == held[2][1] == 0
        assert all(torch.equal(held[0][2][r][0], payload(2, r)[0]) for r in range(world))
        open(os.path.join(tmp, f"ok2_{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_steps_alternate_buffer_sets(tmp_path):
    world = 2
    mp.spawn(_worker_two_steps, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok2_{r}").exists() for r in range(world))


def test_two_rank_exchange_and_combine(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_plan_rejects_ragged_splits():
    assert FrameShardPlan(512, 8, 4).frame_range(3) == (192, 256)
    assert FrameShardPlan(64, 1, 4).windows_per_rank(9, 9) == 1296
    with pytest.raises(ValueError):
        FrameShardPlan(60, 8, 4)
