"""Frame-sharded execution over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference never shards a video; this is the data-parallel extension the
north star asks for.  What makes it exact:
  * local windows never straddle a group of `kt` consecutive frames, so any partition on
    multiples of kt frames gives independent local work per rank;
  * the global softmax spans all frames: each rank streams its frames once and keeps the
    online-softmax state (M, L, ACC[R, E]) of its shard, with positional terms indexed by the
    ABSOLUTE frame number; the states are all-gathered and combined (same kernel that merges
    workgroup partials), then every rank finishes the 32 global rows redundantly.

Exchange = ONE all-gather of a flat per-rank buffer [state | local tokens] (a few MB): on a
fully-connected xGMI node that is 7 concurrent peer writes, not a ring.

`FrameShardPlan` and `exchange()` are device-agnostic (exercised with gloo on CPU in
tests/test_dist_cpu.py); `sharded_forward()` is the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import geometry as geo


@dataclass(frozen=True)
class FrameShardPlan:
    total_frames: int
    world: int
    kt: int

    def __post_init__(self):
        if self.total_frames % (self.world * self.kt):
            raise ValueError(f"{self.total_frames} frames do not split into {self.world} shards of whole "
                             f"{self.kt}-frame window groups")

    @property
    def frames_per_rank(self) -> int:
        return self.total_frames // self.world

    def frame_range(self, rank: int) -> Tuple[int, int]:
        f = self.frames_per_rank
        return rank * f, (rank + 1) * f

    def windows_per_rank(self, wh: int, ww: int) -> int:
        return self.frames_per_rank // self.kt * wh * ww


def exchange(state: torch.Tensor, local_tokens: torch.Tensor, group=None):
    """All-gathers (state f32 [S], local tokens [Nw_rank, H]) of every rank with ONE collective.

    Returns (states [world, S] f32, tokens [world * Nw_rank, H]) in rank order = frame order.
    (Copying reference form used by the CPU/gloo test; the HIP path below gathers a pre-packed
    send buffer that the kernels wrote in place.)"""
    world = dist.get_world_size(group)
    mine, s_bytes, pad = pack_buffer(state.numel(), local_tokens.shape, local_tokens.dtype, state.device)
    mine[:s_bytes].view(torch.float32).copy_(state.reshape(-1))
    mine[s_bytes + pad:].view(local_tokens.dtype).copy_(local_tokens.reshape(-1))
    everyone = gather_buffers(mine, group)
    states = everyone[:, :s_bytes].contiguous().view(torch.float32).view(world, -1)
    tokens = everyone[:, s_bytes + pad:].contiguous().view(local_tokens.dtype).view(world * local_tokens.shape[0], -1)
    return states, tokens


def pack_buffer(state_floats: int, token_shape, token_dtype, device):
    """Per-rank send buffer [state f32 | pad to 16 B | local tokens]; returns (buffer, state bytes, pad)."""
    s_bytes = state_floats * 4
    pad = (-s_bytes) % 16
    t_bytes = token_shape[0] * token_shape[1] * torch.empty((), dtype=token_dtype).element_size()
    tail = (-(s_bytes + pad + t_bytes)) % 16
    return torch.empty(s_bytes + pad + t_bytes + tail, dtype=torch.uint8, device=device), s_bytes, pad


def gather_buffers(mine: torch.Tensor, group=None) -> torch.Tensor:
    world = dist.get_world_size(group)
    flat = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(flat, mine, group=group)      # rank-major concatenation
    return flat.view(world, mine.numel())


class _ShardSet:
    """One set of exchange buffers + argument blocks (two sets alternate so that the all-gather of step i can
    still be reading its send buffer while step i+1 streams into the other one)."""
    __slots__ = ("mine", "everyone", "tok_off", "a_stream", "a_finish", "ev_stream", "ev_tok")


class _ShardPlan:
    __slots__ = ("sets", "n", "comm", "res", "lay", "nw", "hidden", "odt", "n_rows_total", "world")


def _shard_plan(projector, ff_shard, fe_shard, guide_embed, total_frames, image_newline, group):
    from . import engine
    from . import native as nv
    from .projector import _out_dtype
    lc, gc = projector.local_compressor, projector.global_compressor
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = ff_shard.device
    cur = torch.cuda.current_stream(dev)
    key = ("shard", ff_shard.data_ptr(), tuple(ff_shard.shape), None if fe_shard is None else fe_shard.data_ptr(),
           None if guide_embed is None else guide_embed.data_ptr(), None if image_newline is None else image_newline.data_ptr(),
           _out_dtype(projector), world, rank, total_frames, cur.cuda_stream, gc._cache_gen, engine._param_stamp(projector))
    plans = projector.__dict__.setdefault("_engine_plans", {})
    plan = plans.get(key)
    if plan is not None:
        return plan
    shard = FrameShardPlan(total_frames, world, lc.temporal_kernel_size)
    t0, t1 = shard.frame_range(rank)
    if ff_shard.shape[0] != t1 - t0:
        raise ValueError(f"rank {rank} must hold frames [{t0},{t1})")
    T, H, W, E = ff_shard.shape
    hidden = lc.readout[2].out_features
    odt = _out_dtype(projector)
    at, ay, ax = lc.tilings(T, H, W, "video")
    nw = at.nwin * ay.nwin * ax.nwin
    q_in, n_rows = gc.injected_queries(guide_embed)
    R = q_in.shape[0] * gc.attn_layer.num_heads
    S = 2 * R + R * E                                   # (M, L) pairs, then ACC, of one shard
    S_pad = (S + 3) // 4 * 4
    lay = projector._layout((at.nwin * world, ay.nwin, ax.nwin), "video", image_newline is not None, False)
    plan = _ShardPlan()
    plan.comm = torch.cuda.Stream(device=dev, priority=-1)
    plan.res = engine._resources(dev)                  # side stream + fork/join events of the caller's stream
    plan.lay, plan.nw, plan.hidden, plan.odt, plan.world = lay, nw, hidden, odt, world
    plan.n_rows_total = lay.n_rows + n_rows
    plan.n = 0
    plan.sets = []
    probe = torch.empty((plan.n_rows_total, hidden), dtype=odt, device=dev)      # any valid `out` for the argument blocks
    for _ in range(2):
        st = _ShardSet()
        # ONE exchange buffer per rank: [state f32 | pad to 16 B | local tokens] -> one collective per step (the host
        # cost of a torch.distributed call, not the wire, is what a second collective would add)
        st.mine, s_bytes, pad = pack_buffer(S, (nw, hidden), odt, dev)
        st.mine.zero_()
        st.tok_off = s_bytes + pad
        st.everyone = torch.empty((world, st.mine.numel()), dtype=torch.uint8, device=dev)
        state_mine = st.mine[:s_bytes].view(torch.float32)
        tok_mine = st.mine[st.tok_off:st.tok_off + nw * hidden * probe.element_size()].view(odt).view(nw, hidden)
        st.a_stream = engine.build_args(projector, ff_shard, fe_shard, guide_embed, "video", None, probe, None, t_offset=t0,
                                        phases=nv.PHASE_STREAM, local_out=tok_mine, state_out=state_mine,
                                        global_row0=lay.n_rows)
        engine.attach_execution(st.a_stream, dev, key_extra=("shard",))
        st.a_finish = engine.build_args(projector, ff_shard, fe_shard, guide_embed, "video", None, probe, None, t_offset=t0,
                                        phases=nv.PHASE_FINISH, local_out=tok_mine, state_out=state_mine,
                                        state_sets=st.everyone, state_set_stride=st.mine.numel() // 4, nsets=world,
                                        global_row0=lay.n_rows)
        # the FINISH phase runs on the comm stream (own workspace: it reads only the gathered states)
        engine.attach_execution(st.a_finish, dev, key_extra=("shard-finish",), main_stream=plan.comm, res=plan.res)
        st.a_stream._keep = st.a_finish._keep = None
        st.ev_stream, st.ev_tok = torch.cuda.Event(), torch.cuda.Event()
        st.ev_tok.record(cur)
        plan.sets.append(st)
    if len(plans) >= engine._MAX_PLANS:
        plans.pop(next(iter(plans)))
    plans[key] = plan
    return plan


def sharded_forward(projector, ff_shard, fe_shard, guide_embed, total_frames: int,
                    image_newline: Optional[torch.Tensor] = None, group=None, deferred: bool = False):
    """HIComProjector.forward for modal='video' with the frames split evenly over the ranks of
    `group`; every rank passes ITS frames and receives the full [n_tok, hidden] result.

    main stream : STREAM phase of the native executor only (query prep, stream kernel, readout GEMMs; its side
                  stream merges the partials into this rank's softmax state) -- local tokens and state go
                  straight into ONE send buffer [state | tokens]
    comm stream : ONE all-gather (RCCL), the FINISH phase (combine the states, the small global chain -> 32
                  global rows) and ONE launch that places every rank's token block in the packed output.
    deferred=False: the caller's stream waits for the comm stream before returning (plain tensor semantics).
    deferred=True : returns (out, event); the token rows are complete once `event` has fired.  Back-to-back steps
                  then overlap the token exchange of step i with the streaming of step i+1 (two buffer sets)."""
    from . import native as nv
    lc, gc = projector.local_compressor, projector.global_compressor
    if lc is None or gc is None:
        raise NotImplementedError("sharded_forward expects both compressors")
    if not all(t is None or t.is_contiguous() for t in (ff_shard, fe_shard, guide_embed, image_newline)):
        raise ValueError("sharded_forward: contiguous inputs only")
    dev = ff_shard.device
    plan = _shard_plan(projector, ff_shard, fe_shard, guide_embed, total_frames, image_newline, group)
    st = plan.sets[plan.n & 1]
    plan.n += 1
    main, comm = torch.cuda.current_stream(dev), plan.comm
    out = torch.empty((plan.n_rows_total, plan.hidden), dtype=plan.odt, device=dev)
    out.record_stream(comm)
    main.wait_event(st.ev_tok)                     # this buffer set's previous exchange (two steps ago) has drained
    st.a_stream.out = st.a_finish.out = out.data_ptr()
    nv.compressor_fwd(st.a_stream)                 # main: prep, stream kernel, readout GEMMs   side: merge -> state
    st.ev_stream.record(main)
    esz = out.element_size()
    with torch.cuda.stream(comm):                  # c10d orders a collective after the CURRENT stream
        comm.wait_event(st.ev_stream)              # this rank's state and local tokens are complete
        dist.all_gather_into_tensor(st.everyone.view(-1), st.mine, group=group)
        nv.compressor_fwd(st.a_finish)             # combine + the global chain -> the 32 global rows (stream baked in: comm)
        nv.place_blocks(st.everyone.data_ptr() + st.tok_off, plan.nw, plan.world, st.mine.numel(), plan.hidden * esz, out, 0,
                        nl_group=plan.lay.nl_group, stream=comm.cuda_stream)
        if plan.lay.newline_rows:
            first = plan.lay.newline_rows[0]
            step = plan.lay.newline_rows[1] - first if len(plan.lay.newline_rows) > 1 else 1
            nv.scatter_rows(image_newline.view(1, -1), out, first, len(plan.lay.newline_rows), row_step=step)
        st.ev_tok.record(comm)
    if deferred:
        return out, st.ev_tok
    main.wait_event(st.ev_tok)
    return out
