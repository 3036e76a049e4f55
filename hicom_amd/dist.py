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


def sharded_forward(projector, ff_shard, fe_shard, guide_embed, total_frames: int,
                    image_newline: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """HIComProjector.forward for modal='video' with the frames split evenly over the ranks of
    `group`; every rank passes ITS frames and receives the full [n_tok, hidden] result.

    STREAM phase of the native executor writes this rank's local tokens and global softmax state
    straight into the send buffer; one RCCL all-gather; FINISH phase combines the states and writes
    the global rows; one row-scatter places the gathered local tokens."""
    from . import engine
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
    ff_shard = ff_shard.contiguous()
    fe_shard = fe_shard.contiguous() if fe_shard is not None else None
    T, H, W, E = ff_shard.shape
    hidden = lc.readout[2].out_features
    odt = _out_dtype(projector)
    at, ay, ax = lc.tilings(T, H, W, "video")
    nw = at.nwin * ay.nwin * ax.nwin
    q_in, n_rows = gc.injected_queries(guide_embed)
    R = q_in.shape[0] * gc.attn_layer.num_heads
    mine, s_bytes, pad = pack_buffer(2 * R + R * E, (nw, hidden), odt, dev)
    lay = projector._layout((at.nwin * world, ay.nwin, ax.nwin), "video", image_newline is not None, False)
    out = torch.empty((lay.n_rows + n_rows, hidden), dtype=odt, device=dev)
    a = engine.build_args(projector, ff_shard, fe_shard, guide_embed, "video", None, out, None, t_offset=t0,
                          phases=nv.PHASE_STREAM, local_out=mine[s_bytes + pad:], state_out=mine[:s_bytes],
                          global_row0=lay.n_rows)
    engine.attach_execution(a, dev, key_extra=("shard",))
    nv.compressor_fwd(a)
    everyone = gather_buffers(mine, group)
    a.phases = nv.PHASE_FINISH
    a.state_sets, a.nsets, a.state_set_stride = everyone.data_ptr(), world, mine.numel() // 4
    nv.compressor_fwd(a)
    tokens = everyone[:, s_bytes + pad:]
    # gathered local tokens -> packed rows (one strided row-scatter per rank block keeps this a kernel)
    for r in range(world):
        blk = tokens[r, :nw * hidden * out.element_size()].view(odt).view(nw, hidden)
        m0 = r * nw                   # shard boundaries coincide with newline-group boundaries
        nv.scatter_rows(blk, out, m0 + (m0 // lay.nl_group if lay.nl_group else 0), nw, nl_group=lay.nl_group)
    if lay.newline_rows:
        first = lay.newline_rows[0]
        step = lay.newline_rows[1] - first if len(lay.newline_rows) > 1 else 1
        nv.scatter_rows(image_newline.contiguous().view(1, -1), out, first, len(lay.newline_rows), row_step=step)
    return out
