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

`FrameShardPlan`, `PackLayout` and `gather_packed()` are device-agnostic and are THE pack / all-gather / unpack
code of both paths: the HIP path (`sharded_forward`) has its kernels write straight into a `PackLayout` send
buffer and read the gathered buffer at the layout's offsets; the CPU/gloo test (tests/test_dist_cpu.py) fills and
reads the same layout with copies (`exchange`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .events import device_event


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


@dataclass(frozen=True)
class PackLayout:
    """Per-rank exchange buffer: [state f32 x state_floats | pad to 16 B | local tokens rows x cols | pad to 16 B]."""
    state_floats: int
    token_rows: int
    token_cols: int
    token_itemsize: int

    @property
    def state_bytes(self) -> int:
        return self.state_floats * 4

    @property
    def tok_off(self) -> int:
        return self.state_bytes + (-self.state_bytes) % 16

    @property
    def tok_bytes(self) -> int:
        return self.token_rows * self.token_cols * self.token_itemsize

    @property
    def total(self) -> int:
        n = self.tok_off + self.tok_bytes
        return n + (-n) % 16

    @property
    def set_stride_floats(self) -> int:
        """Distance, in floats, between two ranks' states inside the gathered buffer."""
        return self.total // 4

    def new_buffer(self, device, world: int = 0) -> torch.Tensor:
        shape = (self.total,) if world == 0 else (world, self.total)
        return torch.zeros(shape, dtype=torch.uint8, device=device)

    def state_view(self, buf: torch.Tensor) -> torch.Tensor:
        """f32 view of the state part: [state_floats] of a send buffer, [world, state_floats] of a gathered one."""
        return buf[..., :self.state_bytes].view(torch.float32) if buf.ndim == 1 else \
            buf[:, :self.state_bytes].contiguous().view(torch.float32)

    def tokens_view(self, buf: torch.Tensor, dtype) -> torch.Tensor:
        """Token rows: [rows, cols] of a send buffer (a view), [world * rows, cols] of a gathered one (rank order =
        frame order)."""
        a, b = self.tok_off, self.tok_off + self.tok_bytes
        if buf.ndim == 1:
            return buf[a:b].view(dtype).view(self.token_rows, self.token_cols)
        return buf[:, a:b].contiguous().view(dtype).view(buf.shape[0] * self.token_rows, self.token_cols)


class ExchangeSets:
    """The (send buffer, gathered buffer) pairs of a plan, used in turn: step i fills and gathers pair i mod n while the
    consumer of step i - 1 (the comm stream's FINISH phase on the GPU) may still be reading the other pair.  Device-agnostic:
    `sharded_forward` and the world-2 gloo test step through the same object."""

    def __init__(self, lay: "PackLayout", device, world: int, n: int = 2):
        self.pairs = [(lay.new_buffer(device), lay.new_buffer(device, world)) for _ in range(n)]
        self.steps = 0

    def advance(self) -> int:
        """Index of the pair of the step that starts now."""
        i = self.steps % len(self.pairs)
        self.steps += 1
        return i


def gather_packed(mine: torch.Tensor, everyone: torch.Tensor, group=None) -> torch.Tensor:
    """THE collective of the path: all-gathers every rank's packed buffer into `everyone` [world, total] (rank-major),
    ordered on the current stream of `mine`'s device."""
    dist.all_gather_into_tensor(everyone.view(-1), mine, group=group)
    return everyone


def exchange(state: torch.Tensor, local_tokens: torch.Tensor, group=None):
    """All-gathers (state f32 [S], local tokens [Nw_rank, H]) of every rank with ONE collective.
    Returns (states [world, S] f32, tokens [world * Nw_rank, H]) in rank order = frame order."""
    world = dist.get_world_size(group)
    lay = PackLayout(state.numel(), local_tokens.shape[0], local_tokens.shape[1], local_tokens.element_size())
    mine = lay.new_buffer(state.device)
    lay.state_view(mine).copy_(state.reshape(-1))
    lay.tokens_view(mine, local_tokens.dtype).copy_(local_tokens)
    everyone = gather_packed(mine, lay.new_buffer(state.device, world), group)
    return lay.state_view(everyone), lay.tokens_view(everyone, local_tokens.dtype)


class _ShardSet:
    """One set of exchange buffers + argument blocks + workspaces (two sets alternate so that the all-gather of step i
    can still be reading its send buffer while step i+1 streams into the other one)."""
    __slots__ = ("mine", "everyone", "a_stream", "a_finish", "ws_stream", "ws_finish", "ev_stream", "ev_tok", "out", "fused", "r0", "direct_ag", "on_comm",
                 "states_all", "tok_direct")


class _ShardPlan:
    __slots__ = ("sets", "xs", "comm", "res", "lay", "pack", "nw", "hidden", "odt", "n_rows_total", "world", "rank", "sig",
                 "guide_fields")

    def set_inputs(self, st: _ShardSet, ff, fe, guide, out):
        """Patches this call's tensors into both argument blocks of a buffer set."""
        for a in (st.a_stream, st.a_finish):
            a.ff = ff.data_ptr()
            if fe is not None:
                a.fe = fe.data_ptr()
            if guide is not None:
                for f in self.guide_fields:
                    setattr(a, f, guide.data_ptr())
            a.out = out.data_ptr()
        if st.tok_direct:
            st.a_finish.ag_recv = out.data_ptr()       # (the token all-gather lands in the output's first world * nw rows)

    def release(self):
        """Before the plan's buffers go back to the allocator: everything its comm stream still does with them."""
        self.comm.synchronize()


_MAX_SHARD_PLANS = 8
_N_SETS = 4
_COMM_PTRS = {}          # (process group, device) -> ncclComm_t of its RCCL backend, or None


def _rccl_comm(group, dev, world) -> Optional[int]:
    """ncclComm_t of `group`'s RCCL backend as an integer (torch: ProcessGroupNCCL._comm_ptr()), or None when the backend is not RCCL / the
    torch build does not expose it / HICOM_SHARD_DIRECT_AG=0 -- the step then issues its all-gather through torch.distributed.  The
    communicator is created lazily by torch: one (tiny) c10d collective on this device makes sure it exists."""
    import os
    if os.environ.get("HICOM_SHARD_DIRECT_AG", "1") == "0" or not dist.is_initialized():
        return None
    g = group if group is not None else dist.group.WORLD
    try:
        if dist.get_backend(g) != "nccl":
            return None
        backend = g._get_backend(torch.device(dev))
        # (one probe collective per backend INSTANCE -- its `uid` tells a re-created process group from a destroyed one whose id() was reused)
        uid = getattr(backend, "uid", None)
        key = (uid, str(dev))
        if uid is not None and key in _COMM_PTRS:
            return _COMM_PTRS[key]
        probe = torch.zeros(world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(probe, probe[:1].clone(), group=g)
        torch.cuda.current_stream(dev).synchronize()
        ptr = int(backend._comm_ptr()) or None
    except Exception:  # noqa: BLE001  (a torch without _comm_ptr, a wrapped backend: fall back to c10d)
        return None
    if uid is not None:
        if len(_COMM_PTRS) > 16:
            _COMM_PTRS.clear()
        _COMM_PTRS[key] = ptr
    return ptr


def _shard_plan(projector, ff_shard, fe_shard, guide_embed, total_frames, image_newline, group, rank=None, world=None,
                collective=None) -> _ShardPlan:
    """Plan of one (problem shape, rank, world, caller stream, weight state): `_N_SETS` buffer sets, each with its own argument
    blocks, exchange buffers and workspaces, and the plan's comm stream.  The input pointers are patched per call
    (`set_inputs`).  `rank` / `world` default to the process group's; the single-GPU tests of the N > 1 device path pass
    them explicitly (no collective) -- or, with `collective` = (all-gather fn, group-start fn, group-end fn, comm token) as C function
    addresses, a TEST DOUBLE of RCCL's entry points: the FINISH call then takes exactly the path it takes under a real process group
    (hicom_compressor_args.ag_*), with the double doing the copies (tests/test_gpu_parity.py::test_direct_all_gather_form_emulated)."""
    from . import engine
    from . import native as nv
    from .projector import _out_dtype
    lc, gc = projector.local_compressor, projector.global_compressor
    dev = ff_shard.device
    cur = torch.cuda.current_stream(dev)
    _explicit = rank
    if rank is None or world is None:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    key = (tuple(ff_shard.shape), fe_shard is not None, None if guide_embed is None else tuple(guide_embed.shape),
           image_newline is not None, _out_dtype(projector), world, rank, total_frames, cur.cuda_stream, id(group), collective)
    plans = projector.__dict__.setdefault("_shard_plans", {})
    plan = plans.get(key)
    sig = engine.weights_sig(projector)
    if plan is not None and plan.sig == sig:
        return plan
    if plan is not None:
        plans.pop(key).release()
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
    lay = projector._layout((at.nwin * world, ay.nwin, ax.nwin), "video", image_newline is not None, False)
    plan = _ShardPlan()
    import os
    # the comm stream carries the FINISH phase of step i under the streaming of step i + 1.  HIGH priority (the device offers -1 and 0):
    # measured at world size 1, pipelined loop -- -1: 82.5-82.9 us per step, 0: 111.8-113.3 (the chain launch's 32 workgroups hand a vector
    # over inside the launch: at equal priority some of them queue behind the ring kernel's workgroups while the resident ones spin).
    # HICOM_COMM_PRIORITY: dev / A-B switch.
    prio = int(os.environ.get("HICOM_COMM_PRIORITY", "-1"))
    try:
        plan.comm = torch.cuda.Stream(device=dev, priority=prio)
    except Exception:  # noqa: BLE001  (a priority outside the device's range)
        plan.comm = torch.cuda.Stream(device=dev)
    plan.res = engine._resources(dev)                  # side stream + fork/join events of the caller's stream
    plan.lay, plan.nw, plan.hidden, plan.odt, plan.world, plan.rank = lay, nw, hidden, odt, world, rank
    plan.n_rows_total = lay.n_rows + n_rows
    plan.sets = []
    # ONE exchange buffer per rank: [state (M, L) pairs + ACC | local tokens] -> one collective per step (the host cost
    # of a torch.distributed call, not the wire, is what a second collective would add)
    plan.pack = pack = PackLayout(2 * R + R * E, nw, hidden, torch.empty((), dtype=odt).element_size())
    probe = torch.empty((plan.n_rows_total, hidden), dtype=odt, device=dev)      # any valid `out` for the argument blocks
    # FOUR buffer sets, and the HOST is what keeps a set's next step behind its previous exchange (sharded_forward: a host-side wait on
    # ev_tok when the set is still busy, which also bounds how far the enqueueing thread runs ahead of the device): with two sets the
    # dependency had to be a device-side wait on the MAIN stream in every steady-state step -- a barrier packet there is a ~10-us bubble
    # between readout GEMM 2 and the next step's query prep (tools/shard_trace.py: 89.9 us from ring to ring against 75 of kernels)
    plan.xs = ExchangeSets(pack, dev, world, n=_N_SETS)
    comm_ptr = _rccl_comm(group, dev, world) if _explicit is None else None      # (explicit rank / world: the one-GPU emulations, no collective)
    if collective is not None:
        comm_ptr = collective[3]
    for mine, everyone in plan.xs.pairs:
        st = _ShardSet()
        st.mine, st.everyone = mine, everyone
        state_mine, tok_mine = pack.state_view(st.mine), pack.tokens_view(st.mine, odt)
        st.a_stream = engine.build_args(projector, ff_shard, fe_shard, guide_embed, "video", None, probe, None, t_offset=t0,
                                        phases=nv.PHASE_STREAM, local_out=tok_mine, state_out=state_mine,
                                        global_row0=lay.n_rows)
        # one workspace per buffer set AND plan: with MERGE_ON_NEXT the comm stream merges this set's partial states
        # while the main stream already runs the next step (on the other set, or of another plan)
        st.ws_stream = engine.attach_execution(st.a_stream, dev)
        st.a_finish = engine.build_args(projector, ff_shard, fe_shard, guide_embed, "video", None, probe, None, t_offset=t0,
                                        phases=nv.PHASE_FINISH, local_out=tok_mine, state_out=state_mine,
                                        state_sets=st.everyone, state_set_stride=pack.set_stride_floats, nsets=world,
                                        global_row0=lay.n_rows)
        # the FINISH phase runs on the comm stream (it reads only the gathered states)
        st.ws_finish = engine.attach_execution(st.a_finish, dev, main_stream=plan.comm, res=plan.res)
        plan.guide_fields = st.a_stream._guide_ptr_fields
        st.a_stream._keep = st.a_finish._keep = None
        st.out = torch.empty((plan.n_rows_total, hidden), dtype=odt, device=dev)
        st.ev_stream, st.ev_tok = device_event(), torch.cuda.Event()      # (ev_stream: stream-to-stream on this device -- no system-scope fence, events.py)
        st.ev_stream.record(cur)                   # (hipEventRecord from C needs created events)
        st.ev_tok.record(cur)
        # what follows each phase on its stream rides in the same C call (every separate host call is 3-6 us and the
        # sharded step is host-bound): STREAM -> record ev_stream, comm waits for it; FINISH -> place every rank's
        # token block into the packed output, record ev_tok
        st.a_stream.ev_done, st.a_stream.stream_next = st.ev_stream.cuda_event, plan.comm.cuda_stream
        st.fused = nv.compressor_is_fused(st.a_stream)
        st.r0 = None
        if st.fused:
            # four-launch form of the sharded step (executor.hip: shard4 / finish4): r0 travels from the STREAM phase's query prep to the
            # FINISH phase's chain launch through this buffer (their workspaces are separate)
            st.r0 = torch.zeros(hidden, dtype=torch.float32, device=dev)
            st.a_stream.r0_buf = st.a_finish.r0_buf = st.r0.data_ptr()
        if st.fused:
            # release recipe: no side stream -- the merge of the partials runs on the comm stream, in front of the
            # all-gather (the fork / join / ev_merge event traffic was ~17 us of host time on a host-bound step)
            st.a_stream.phases = nv.PHASE_STREAM | nv.PHASE_MERGE_ON_NEXT
        st.a_finish.place_src = st.everyone.data_ptr() + pack.tok_off
        st.a_finish.place_block_rows, st.a_finish.place_nblocks = nw, world
        st.a_finish.place_block_stride = pack.total
        st.a_finish.nl_group = lay.nl_group
        st.a_finish.ev_done = st.ev_tok.cuda_event
        if st.fused and not nv.compressor_takes_shard4(st.a_stream):
            # FINISH's four-launch form consumes the r0 that ONLY the STREAM call's four-launch form writes -- both or neither
            st.a_stream.r0_buf = st.a_finish.r0_buf = None
        # the all-gather enqueued by the FINISH call itself (RCCL through the group's own communicator): one host call per step
        st.on_comm = True
        st.direct_ag = comm_ptr is not None
        st.states_all, st.tok_direct = None, False
        if st.direct_ag:
            fn_ag, fn_gs, fn_ge = collective[:3] if collective is not None else nv.rccl_fns()
            st.a_finish.ag_fn, st.a_finish.ag_comm = fn_ag, comm_ptr
            st.a_finish.ag_send, st.a_finish.ag_recv, st.a_finish.ag_bytes = st.mine.data_ptr(), st.everyone.data_ptr(), pack.total
            if not lay.newline_rows and lay.nl_group == 0 and os.environ.get("HICOM_SHARD_PLACE", "0") != "1":
                # no newline rows: the ranks' token blocks are CONSECUTIVE rows of the output -- the token all-gather writes them there
                # itself, a second all-gather in the same RCCL group carries the states: no placement launch (a 2.3 x world MB copy on the
                # comm stream beside the next step's ring kernel; HICOM_SHARD_PLACE=1: the one-buffer form + placement, A/B switch)
                st.tok_direct = True
                st.states_all = torch.zeros((world, pack.state_floats), dtype=torch.float32, device=dev)
                st.a_finish.ag_group_start, st.a_finish.ag_group_end = fn_gs, fn_ge
                st.a_finish.ag_send, st.a_finish.ag_bytes = st.mine.data_ptr() + pack.tok_off, pack.tok_bytes
                st.a_finish.ag_send2, st.a_finish.ag_recv2, st.a_finish.ag_bytes2 = st.mine.data_ptr(), st.states_all.data_ptr(), pack.state_bytes
                st.a_finish.state_sets, st.a_finish.state_set_stride = st.states_all.data_ptr(), pack.state_floats
                st.a_finish.place_src = None
        plan.sets.append(st)
    plan.sig = engine.weights_sig(projector)
    if len(plans) >= _MAX_SHARD_PLANS:
        plans.pop(next(iter(plans))).release()
    plans[key] = plan
    return plan


def _set_stream(stream):
    """torch.cuda.set_stream without the context manager's device / current-stream lookups (a few us each)."""
    torch._C._cuda_setStream(stream_id=stream.stream_id, device_index=stream.device_index, device_type=stream.device_type)


def sharded_forward(projector, ff_shard, fe_shard, guide_embed, total_frames: int,
                    image_newline: Optional[torch.Tensor] = None, group=None, deferred: bool = False):
    """HIComProjector.forward for modal='video' with the frames split evenly over the ranks of
    `group`; every rank passes ITS frames and receives the full [n_tok, hidden] result.

    main stream : STREAM phase of the native executor only (query prep, stream kernel, readout GEMMs) -- local tokens
                  and the shard's softmax state go straight into ONE send buffer [state | tokens]
    comm stream : merge of the partial states, ONE all-gather (RCCL), the FINISH phase (combine the states, the small
                  global chain -> 32 global rows) and ONE launch that places every rank's token block in the packed
                  output.
    deferred=False: the caller's stream waits for the comm stream before returning (plain tensor semantics).
    deferred=True : returns (out, event); the token rows are complete once `event` has fired.  Back-to-back steps
                  then overlap the token exchange of step i with the streaming of step i+1 (two buffer sets); `out`
                  belongs to the buffer set and is overwritten by the second-next deferred call on this shape."""
    from . import native as nv
    lc, gc = projector.local_compressor, projector.global_compressor
    if lc is None or gc is None:
        raise NotImplementedError("sharded_forward expects both compressors")
    projector._check_clip_logits()
    if torch.is_grad_enabled() and projector._needs_grad(ff_shard, fe_shard, guide_embed, image_newline):
        raise RuntimeError("sharded_forward is an inference path: call it under torch.no_grad() / inference_mode()")
    nv.begin_inference()
    if not projector._executor_covers() or not projector._queries_native() or projector.global_logit is not None:
        # coarse / fine / query-side adaptor recipes (reference projector.py:369-397, :431-441): their injected queries are computed
        # per call from the guide -- redundantly on every rank for the 32 global rows, from the shard's own frames for the pooled
        # window queries (window-local: projector.py:539-542) -- so they shard operator by operator (round 5)
        out = sharded_forward_stepwise(projector, ff_shard, fe_shard, guide_embed, total_frames, image_newline, group)
        if deferred:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(ff_shard.device))
            return out, ev
        return out
    if not all(t is None or t.is_contiguous() for t in (ff_shard, fe_shard, guide_embed, image_newline)):
        raise ValueError("sharded_forward: contiguous inputs only")
    dev = ff_shard.device
    plan = _shard_plan(projector, ff_shard, fe_shard, guide_embed, total_frames, image_newline, group)
    st = plan.sets[plan.xs.advance()]
    main, comm = torch.cuda.current_stream(dev), plan.comm
    if deferred:
        # pipelined serving: the result lives in this buffer set (no allocator traffic on the host-bound path) and is
        # overwritten by the comm stream two steps later, i.e. after everything the caller has queued on ITS stream
        # before that step
        out = st.out
    else:
        out = torch.empty((plan.n_rows_total, plan.hidden), dtype=plan.odt, device=dev)
        out.record_stream(comm)
    # this buffer set's previous exchange (_N_SETS steps ago) has to have drained before the step overwrites the set.  Enforced on the
    # HOST: a device-side wait costs the main stream a barrier packet -- a ~10-us bubble in its launch pipeline, every step, because the
    # enqueueing thread runs ahead of the device and the event is never complete yet at enqueue time.  Blocking here instead bounds that
    # run-ahead to _N_SETS steps and keeps the main stream free of cross-stream waits.
    if not st.ev_tok.query():
        st.ev_tok.synchronize()
    if deferred != st.on_comm:
        _route_finish(plan, st, main, deferred)
    plan.set_inputs(st, ff_shard, fe_shard, guide_embed, out)
    # main: prep, stream kernel, readout GEMMs, ev_stream; the comm stream waits for it and merges the partials -> state
    if st.direct_ag and not plan.lay.newline_rows:
        # ONE host call for the step: STREAM on the main stream, then -- on the comm stream, behind ev_stream -- ncclAllGather, the merge
        # of the gathered states, the chain, the token placement and ev_tok (executor.hip)
        nv.compressor_fwd2(st.a_stream, st.a_finish)
    else:
        nv.compressor_fwd(st.a_stream)
        _comm_step(plan, st, out, image_newline, group, main)
    if deferred:
        # the comm stream reads the guide (residual of out_proj) and the newline token after this call has returned
        for t in (guide_embed, image_newline):
            if t is not None:
                t.record_stream(comm)
        return out, st.ev_tok
    return out                                     # (joined: the whole step ran on the caller's stream, _route_finish)


def _route_finish(plan, st, main, on_comm: bool):
    """Where a buffer set's FINISH phase runs.  Pipelined serving (deferred=True): on the plan's comm stream, behind ev_stream, under the
    next step's streaming.  Joined call: on the CALLER's stream, right behind STREAM -- the result is needed at once, and two cross-stream
    hops (main -> comm -> main, ~10 us each on this platform) bought nothing: world-1 joined step 133 -> ~100 us."""
    from . import native as nv
    st.a_finish.stream_main = (plan.comm if on_comm else main).cuda_stream
    if on_comm:
        st.a_stream.phases &= ~nv.PHASE_NEXT_IS_MAIN
    else:
        st.a_stream.phases |= nv.PHASE_NEXT_IS_MAIN    # (the executor then puts no event wait between the phases; stream_next is ignored)
    st.on_comm = on_comm


def _comm_step(plan, st, out, image_newline, group, restore=None):
    """Comm-stream half of a step: ONE all-gather, then (one C call) combine + the global chain -> the 32 global
    rows, every rank's token block into the packed output, ev_tok."""
    from . import native as nv
    comm = plan.comm if st.on_comm else (restore if restore is not None else torch.cuda.current_stream(st.mine.device))
    if not st.direct_ag:
        if st.on_comm:
            _set_stream(comm)                      # c10d orders a collective after the CURRENT (thread-local) stream
        try:
            gather_packed(st.mine, st.everyone, group)
        finally:
            if st.on_comm and restore is not None:
                _set_stream(restore)
    nv.compressor_fwd(st.a_finish)                 # (direct_ag: enqueues ncclAllGather itself, in front of everything else)                 # (records ev_tok = its ev_done behind the token placement, from C)
    fenced = bool(st.a_finish.ev_done)
    if plan.lay.newline_rows:
        first = plan.lay.newline_rows[0]
        step = plan.lay.newline_rows[1] - first if len(plan.lay.newline_rows) > 1 else 1
        nv.scatter_rows(image_newline.view(1, -1), out, first, len(plan.lay.newline_rows), row_step=step, stream=comm.cuda_stream)
        fenced = False                             # the newline rows were written behind the C call's record
    if not fenced:
        st.ev_tok.record(comm)                     # ev_tok always covers the LAST write of the step into `out`


# ---- operator-by-operator sharded path: every recipe (coarse / fine injection, query-side adaptors, clip-scale on the local stage) ----

def stepwise_shard_send(projector, ff_shard, fe_shard, guide_embed, total_frames: int, rank: int, world: int):
    """STREAM half of a shard, one C-ABI call per operator: the shard's window contexts -> readout -> token rows, and the shard's
    online-softmax state of the global stage with ABSOLUTE frame indices for the positional terms; both packed into ONE send buffer.
    Returns (PackLayout, send buffer uint8 [total], (q_in, n_rows, nw, grid)).  The injected queries are computed here from the guide:
    the 32 global rows' redundantly on every rank (reference projector.py:642), the per-window ones from the shard's own frames
    (pooled queries are window-local, :539-542)."""
    from .projector import _out_dtype
    lc, gc = projector.local_compressor, projector.global_compressor
    shard = FrameShardPlan(total_frames, world, lc.temporal_kernel_size)
    t0, t1 = shard.frame_range(rank)
    if ff_shard.shape[0] != t1 - t0:
        raise ValueError(f"rank {rank} must hold frames [{t0},{t1})")
    dev, odt = ff_shard.device, _out_dtype(projector)
    hidden = lc.readout[2].out_features
    ctx, grid = lc.window_context(ff_shard, fe_shard, guide_embed, "video", *projector._logit_args("local"))
    nw = grid[0] * grid[1] * grid[2]
    q_in, n_rows = gc.injected_queries(guide_embed)
    # (clip-scale on the global stage, reference projector.py:184-191: the key norms are per token, i.e. shard-local like the windows; the
    # shard states merge like any others -- round 6)
    ml, acc, _ = gc.partial_context(ff_shard, q_in, t_offset=t0, logit_scale=projector._logit_args("global")[0])
    R, E = acc.shape
    lay = PackLayout(2 * R + R * E, nw, hidden, torch.empty((), dtype=odt).element_size())
    mine = lay.new_buffer(dev)
    state = lay.state_view(mine)
    state[:2 * R].copy_(ml.reshape(-1))
    state[2 * R:].copy_(acc.reshape(-1))
    lc.readout_into(ctx, lay.tokens_view(mine, odt), 0, 0)            # plain rows: the packing (newline groups) happens at placement
    return lay, mine, (q_in, n_rows, nw, grid)


def stepwise_shard_finish(projector, lay: PackLayout, everyone: torch.Tensor, meta, world: int, image_newline=None) -> torch.Tensor:
    """FINISH half on the gathered buffers [world, total]: every rank's token block into the packed output (reference row order
    `[local (t h w) (+ newline rows) ; global]`, projector.py:707, mm_utils.py:92-140), the states combined, the global chain."""
    from . import native as nv
    from .projector import _out_dtype
    lc, gc = projector.local_compressor, projector.global_compressor
    q_in, n_rows, nw, grid = meta
    dev, odt = everyone.device, _out_dtype(projector)
    hidden = lc.readout[2].out_features
    playout = projector._layout((grid[0] * world, grid[1], grid[2]), "video", image_newline is not None, False)
    out = torch.empty((playout.n_rows + n_rows, hidden), dtype=odt, device=dev)
    nv.place_blocks(everyone.data_ptr() + lay.tok_off, nw, world, lay.total, hidden * out.element_size(), out, 0, playout.nl_group)
    if playout.newline_rows:
        first = playout.newline_rows[0]
        step = playout.newline_rows[1] - first if len(playout.newline_rows) > 1 else 1
        nv.scatter_rows(image_newline.contiguous().view(1, -1), out, first, len(playout.newline_rows), row_step=step)
    states = lay.state_view(everyone)                                    # [world, 2R + R E] f32
    R = q_in.shape[0] * gc.attn_layer.num_heads
    E = gc.embed_dim
    ml_sets = states[:, :2 * R].reshape(world, R, 2).contiguous()
    acc_sets = states[:, 2 * R:].reshape(world, R, E).contiguous()
    gc.finish(ml_sets, acc_sets, q_in, out, playout.n_rows, n_rows)
    return out


def sharded_forward_stepwise(projector, ff_shard, fe_shard, guide_embed, total_frames: int, image_newline=None, group=None):
    """sharded_forward for the recipes outside the one-call executor: send half, ONE all-gather (RCCL), finish half, all on the
    caller's stream.  Same exchange format (`PackLayout`, `gather_packed`) as the release path."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lay, mine, meta = stepwise_shard_send(projector, ff_shard, fe_shard, guide_embed, total_frames, rank, world)
    everyone = gather_packed(mine, lay.new_buffer(mine.device, world), group)
    return stepwise_shard_finish(projector, lay, everyone, meta, world, image_newline)
