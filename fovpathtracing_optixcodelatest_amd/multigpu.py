"""Multi-GPU plumbing for the tile-sharded frame (one process per GPU, torch.distributed).

The reference is single-GPU (PT_sv5_/SimplePathtracer.cpp:331-340); sharding follows the unused
SDK scheme of sutil/WorkDistribution.h:47-84 (small interleaved tiles) at launch-index granularity.
Each rank renders only the launch indices it owns (fovpt_config.rank/world) into a full-size frame in
which every other pixel is zero; the owned pixel sets are disjoint and cover the frame, so ONE
sum-reduce onto rank 0 over RCCL/xGMI *is* the gather of the final framebuffer.  No collective is
needed while rendering: seeds depend only on the launch index (deviceProgram.cu:411).
"""
import torch
import torch.distributed as dist


def launch_owner(pass_index, lx, ly, world, tile_w=8, tile_h=4):
    """Rank that owns launch index (lx, ly) of a pass -- mirrors launch_owned() in csrc/wavefront.hip."""
    if world <= 1:
        return 0
    return ((lx // tile_w) + 3 * (ly // tile_h) + pass_index) % world


def gather_frame(frame: torch.Tensor, dst: int = 0, group=None, async_op: bool = False):
    """Sum-reduce the per-rank frames (int32 rgba8 words, or float accum) onto rank `dst`.
    With async_op=True returns the work handle (None when there is nothing to do) so the caller can
    overlap the gather of frame k with the rendering of frame k+1."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None if async_op else frame
    work = dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else frame
