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

    Returns (states [world, S] f32, tokens [world * Nw_rank, H]) in rank order = frame order."""
    world = dist.get_world_size(group)
    s_bytes = state.numel() * 4
    t_bytes = local_tokens.numel() * local_tokens.element_size()
    pad = (-s_bytes) % 16
    mine = torch.empty(s_bytes + pad + t_bytes, dtype=torch.uint8, device=state.device)
    mine[:s_bytes].view(torch.float32).copy_(state.reshape(-1))
    mine[s_bytes + pad:].view(local_tokens.dtype).copy_(local_tokens.reshape(-1))
    flat = torch.empty(world * mine.numel(), dtype=torch.uint8, device=state.device)
    dist.all_gather_into_tensor(flat, mine, group=group)      # rank-major concatenation
    everyone = flat.view(world, mine.numel())
    states = everyone[:, :s_bytes].contiguous().view(torch.float32).view(world, -1)
    tokens = everyone[:, s_bytes + pad:].contiguous().view(local_tokens.dtype).view(world * local_tokens.shape[0], -1)
    return states, tokens


def sharded_forward(projector, ff_shard, fe_shard, guide_embed, total_frames: int,
                    image_newline: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """HIComProjector.forward for modal='video' with the frames split evenly over the ranks of
    `group`; every rank passes ITS frames and receives the full [n_tok, hidden] result."""
    from . import native as nv
    from .projector import _out_dtype
    lc, gc = projector.local_compressor, projector.global_compressor
    if lc is None or gc is None:
        raise NotImplementedError("sharded_forward expects both compressors")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = FrameShardPlan(total_frames, world, lc.temporal_kernel_size)
    t0, t1 = plan.frame_range(rank)
    if ff_shard.shape[0] != t1 - t0:
        raise ValueError(f"rank {rank} must hold frames [{t0},{t1})")
    dev = ff_shard.device
    hidden = lc.readout[2].out_features
    odt = _out_dtype(projector)
    # local tokens of this shard (unpacked; packing happens after the gather)
    ctx, grid = lc.window_context(ff_shard, fe_shard, guide_embed, "video",
                                  projector.local_logit_scale, projector.local_logit_bias)
    loc = torch.empty((ctx.shape[0], hidden), dtype=odt, device=dev)
    lc.readout_into(ctx, loc, 0, 0)
    # global online-softmax state of this shard
    gc._check_native(projector.global_logit_scale)
    q_in, n_rows = gc.injected_queries(guide_embed)
    ml, acc, _ = gc.partial_context(ff_shard, q_in, t_offset=t0)
    R, E = acc.shape
    state = torch.cat([ml.reshape(-1), acc.reshape(-1)])
    states, tokens = exchange(state, loc, group)
    ml_sets = states[:, :2 * R].contiguous().view(world, R, 2)
    acc_sets = states[:, 2 * R:].contiguous().view(world, R, E)
    lay = projector._layout((grid[0] * world, grid[1], grid[2]), "video", image_newline is not None, False)
    out = torch.empty((lay.n_rows + n_rows, hidden), dtype=odt, device=dev)
    if lay.n_rows == lay.n_tokens:
        out[:lay.n_tokens].copy_(tokens)
    else:
        nv.scatter_rows(tokens, out, 0, lay.n_tokens, nl_group=lay.nl_group)
        first = lay.newline_rows[0]
        step = lay.newline_rows[1] - first if len(lay.newline_rows) > 1 else 1
        nv.scatter_rows(image_newline.contiguous().view(1, -1), out, first, len(lay.newline_rows), row_step=step)
    gc.finish(ml_sets, acc_sets, q_in, out, lay.n_rows, n_rows)
    return out
