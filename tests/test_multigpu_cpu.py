"""The N>1 plumbing on CPU: two gloo ranks each hold the pixels of the launch tiles they own
(zero elsewhere); one sum-reduce onto rank 0 must reproduce the unsharded frame exactly."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fovpathtracing_optixcodelatest_amd import multigpu


def _owner_image(w, h, world):
    """Owner rank per pixel for a uniform pass (factor 1): the launch index is the pixel."""
    ys, xs = np.mgrid[0:h, 0:w]
    return multigpu.launch_owner(0, xs, ys, world)


def _worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(path)
    own = _owner_image(full.shape[1], full.shape[0], world)
    mine = np.where(own == rank, full, 0).astype(np.int32)
    t = torch.from_numpy(mine.copy()).reshape(-1)
    multigpu.gather_frame(t, dst=0)
    if rank == 0:
        np.save(path + ".out.npy", t.numpy().reshape(full.shape))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_ownership_partitions_the_launch_grid():
    for world in (2, 3, 4, 8):
        own = _owner_image(96, 64, world)
        assert set(np.unique(own)) == set(range(world))
        counts = np.bincount(own.ravel(), minlength=world)
        assert counts.min() > 0.6 * counts.max()          # interleaved 8x4 tiles balance the load
    # pass index rotates the pattern so the fovea is not owned by one rank in every pass
    assert multigpu.launch_owner(0, 0, 0, 4) != multigpu.launch_owner(1, 0, 0, 4)


def test_gloo_sum_gather_reproduces_frame(tmp_path):
    rng = np.random.default_rng(0)
    full = rng.integers(0, 2 ** 31 - 1, size=(64, 96), dtype=np.int32)
    path = str(tmp_path / "full.npy")
    np.save(path, full)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, path), nprocs=2, join=True)
    out = np.load(path + ".out.npy")
    assert np.array_equal(out, full)
