"""Multi-GPU plumbing for the tile-sharded frame (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-GPU (PT_sv5_/SimplePathtracer.cpp:331-340); sharding follows the unused SDK scheme of
sutil/WorkDistribution.h:47-84 (small interleaved tiles, dealt round-robin) at launch-index granularity.  Each rank
renders only the launch indices it owns (fovpt_config.rank/world); no collective is needed while rendering, because
seeds depend only on the launch index (deviceProgram.cu:411).  The pixels of the finished frame partition by the owner
of their last writer, so the gather of the final framebuffer is

    pack    (HIP, fovpt_gather_pack)     this rank's owned rgba8 words -> one contiguous buffer, ~1/N of the frame
    gather  (RCCL)                       N packed buffers -> rank 0 (grouped send / recv over xGMI)
    unpack  (HIP, fovpt_gather_unpack)   rank 0 scatters them into the frame

`gather_frame` (one full-frame sum-reduce, N times the bytes) remains as the simple alternative and as the check.
"""
import numpy as np
import torch
import torch.distributed as dist


def launch_owner(pass_index, lx, ly, world, tile_w=8, tile_h=4):
    """Rank that owns launch index (lx, ly) of a pass -- mirrors launch_owned() in csrc/wavefront.hip."""
    if world <= 1:
        return 0
    return ((lx // tile_w) + 3 * (ly // tile_h) + pass_index) % world


def gather_frame(frame: torch.Tensor, dst: int = 0, group=None, async_op: bool = False, force_collective: bool = False):
    """Sum-reduce the per-rank frames (int32 rgba8 words, or float accum) onto rank `dst`.
    With async_op=True returns the work handle (None when there is nothing to do) so the caller can
    overlap the gather of frame k with the rendering of frame k+1.  force_collective: issue the collective even
    in a group of one (tests: the RCCL code path on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return None if async_op else frame
    work = dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else frame


# ---- the plan in numpy: what fovpt_gather_plan computes on the device (tests, and the CPU rehearsal of the collective) ----
def plan_from_owner_map(owner: np.ndarray, world: int):
    """owner: per-pixel owning rank (255 = nobody), any shape -> list of ascending flat pixel indices per rank."""
    flat = np.asarray(owner).reshape(-1)
    return [np.nonzero(flat == r)[0].astype(np.int64) for r in range(world)]


def gather_packed(packed: torch.Tensor, stride: int, dst: int = 0, group=None, out: torch.Tensor = None):
    """Gather every rank's packed buffer (padded to `stride` words) onto rank `dst`; returns the (world, stride) tensor
    there, None elsewhere.  NCCL/RCCL implements gather as grouped send / recv; gloo (CPU rehearsal) needs CPU tensors."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    buf = packed if packed.numel() == stride else torch.nn.functional.pad(packed, (0, stride - packed.numel()))
    staged = buf.cpu() if (backend == "gloo" and buf.is_cuda) else buf
    if rank == dst:
        g = out if (out is not None and out.device == staged.device) else torch.empty((world, stride), dtype=staged.dtype, device=staged.device)
        dist.gather(staged, gather_list=list(g.unbind(0)), dst=dst, group=group)
        if out is not None and g is not out:
            out.copy_(g)
            return out
        return g.to(packed.device) if g.device != packed.device else g
    dist.gather(staged, gather_list=None, dst=dst, group=group)
    return None


class PackedGather:
    """Per-frame gather of the owned pixels of a SampleRenderer's frame onto rank `dst`.

    r: renderer.SampleRenderer with config.rank/world set; frame tensors are int32 [H*W] on the renderer's device.
    All device work (pack, unpack) runs on the library's frame-completion stream (fovpt_stream()); issue `gather`
    under `torch.cuda.stream(ExternalStream(r.stream))` so the collective is ordered behind the pack on the device."""

    def __init__(self, r, device, dst=0, group=None, nbuffers=2, force_collective=False):
        self.r, self.device, self.dst, self.group = r, device, dst, group
        self.force_collective = force_collective and dist.is_initialized()      # tests: a group of one still calls RCCL
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nbuffers = nbuffers
        self.counts, self.stride = None, 0
        self.packed, self.gathered = [], []

    def plan(self):
        """(Re)build the plan for the renderer's current config / frame size / gaze; cheap when nothing changed."""
        counts = self.r.gather_plan()
        if counts != self.counts:
            self.counts = counts
            self.stride = (max(counts) + 63) // 64 * 64 or 64
            self.packed = [torch.zeros(self.stride, dtype=torch.int32, device=self.device) for _ in range(self.nbuffers)]
            self.gathered = [torch.zeros((self.world, self.stride), dtype=torch.int32, device=self.device)
                             for _ in range(self.nbuffers if self.rank == self.dst else 0)]
        return counts

    def bytes_per_rank(self):
        return self.stride * 4

    def gather(self, frame: torch.Tensor, k: int = 0, async_op: bool = False):
        """pack + gather of buffer set k.  Returns the work handle (async) or None; call finish() afterwards on dst."""
        self.r.gather_pack(frame.data_ptr(), self.packed[k].data_ptr())
        if self.world == 1 and not self.force_collective:
            if self.rank == self.dst:
                self.gathered[k][0].copy_(self.packed[k])
            return None
        if dist.get_backend(self.group) == "gloo":                       # rehearsal on one GPU: host-staged, synchronous
            self.r.synchronize()
            gather_packed(self.packed[k], self.stride, self.dst, self.group, out=self.gathered[k] if self.rank == self.dst else None)
            return None
        gl = list(self.gathered[k].unbind(0)) if self.rank == self.dst else None
        return dist.gather(self.packed[k], gather_list=gl, dst=self.dst, group=self.group, async_op=async_op)

    def finish(self, frame: torch.Tensor, k: int = 0):
        """On dst: scatter the gathered buffers into the frame (asynchronous on the library's stream)."""
        if self.rank == self.dst:
            self.r.gather_unpack(self.gathered[k].data_ptr(), self.stride, frame.data_ptr())
