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


def _worker_packed(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(path)
    own = _owner_image(full.shape[1], full.shape[0], world)
    own[3:5, 7:19] = 255                                          # a few pixels nobody writes (holes between the rings)
    plan = multigpu.plan_from_owner_map(own, world)
    counts = [len(p) for p in plan]
    stride = (max(counts) + 63) // 64 * 64
    mine = np.where(own == rank, full, 0).astype(np.int32)        # what this rank rendered: zero outside its pixels
    packed = torch.from_numpy(mine.reshape(-1)[plan[rank]].copy())
    g = multigpu.gather_packed(packed, stride, dst=0)
    if rank == 0:
        out = np.full(full.size, -7, np.int32)                    # holes keep what the root's buffer held
        for r in range(world):
            out[plan[r]] = g[r, :counts[r]].numpy()
        np.save(path + ".packed.npy", out.reshape(full.shape))
        np.save(path + ".bytes.npy", np.array([stride * 4, full.size * 4]))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_packed_gather_reproduces_frame(tmp_path):
    """The packed gather (pack owned pixels -> gather -> scatter on the root) over gloo with two ranks: the root ends up
    with every rank's pixels, holes untouched, and each rank sends about half a frame instead of a whole one."""
    rng = np.random.default_rng(1)
    full = rng.integers(1, 2 ** 31 - 1, size=(64, 96), dtype=np.int32)
    path = str(tmp_path / "full.npy")
    np.save(path, full)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_packed, args=(2, port, path), nprocs=2, join=True)
    out = np.load(path + ".packed.npy")
    own = _owner_image(96, 64, 2)
    hole = np.zeros_like(own, bool)
    hole[3:5, 7:19] = True
    assert np.array_equal(out[~hole], full[~hole]) and (out[hole] == -7).all()
    sent, whole = np.load(path + ".bytes.npy")
    assert sent < 0.6 * whole
