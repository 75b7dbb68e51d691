"""Lifecycle / API-usage parity: what PT_sv5_/main.cpp does around render() -- repeated frames,
moving gaze and camera, subframe bookkeeping, resize, scene and probe replacement, depth and spp
settings -- each frame checked bit-exactly against the oracle driven the same way."""
import os

import numpy as np
import pytest

from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes

from common import cfg_foveated, cfg_uniform, make_gpu, make_oracle

pytestmark = pytest.mark.gpu


def _eq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_frame_sequence_with_moving_gaze_and_camera(oracle):
    """Five frames like the app's loop (main.cpp:347-481): the caller pokes frame.c and the camera every
    frame and bumps subframe_index after render() (:479); render() itself bumps it too (:210-211)."""
    size = (160, 96)
    model, probe = scenes.cornell_box(), scenes.sky_probe()
    cfg = cfg_foveated(10, 28, (1, 2, 4))
    r = make_gpu(model, probe, scenes.CORNELL_CAMERA, size, cfg)
    S, F = make_oracle(oracle, model, probe, scenes.CORNELL_CAMERA, size)
    for k in range(5):
        gaze = (40 + 20 * k, 30 + 9 * k)
        eye = (278.0 + 15 * k, 273.0, -800.0 + 30 * k)
        cam = dict(scenes.CORNELL_CAMERA, eye=eye)
        r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], size[0] / size[1]))
        U, V, W = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], size[0] / float(size[1]))
        F.lp.camera.eye.set(eye); F.lp.camera.U.set(U); F.lp.camera.V.set(V); F.lp.camera.W.set(W)
        for lp in (r.launchParams, F.lp):
            lp.frame.c.x, lp.frame.c.y = gaze
        r.render()
        oracle.render(S, F, cfg)
        assert r.launchParams.frame.subframe_index == F.lp.frame.subframe_index == 2 * k + 1
        assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame), k
        for lp in (r.launchParams, F.lp):
            lp.frame.subframe_index += 1                       # main.cpp:479
    r.close()


def test_progressive_accumulation_over_frames(oracle):
    """accumulate = 1 with a running subframe_index: the periphery pass blends into its history
    (PT_sv4_vmv2/deviceProgram.cu:545-553), the redraw passes do not."""
    size = (128, 80)
    model, probe = scenes.cornell_box(), scenes.sky_probe()
    cfg = cfg_foveated(8, 24, (1, 2, 4))
    cfg.accumulate = 1
    r = make_gpu(model, probe, scenes.CORNELL_CAMERA, size, cfg)
    S, F = make_oracle(oracle, model, probe, scenes.CORNELL_CAMERA, size)
    for k in range(4):
        r.render()
        oracle.render(S, F, cfg)
        assert _eq(r.downloadAccum(), F.accum), k
    r.close()


@pytest.mark.parametrize("depth,spp", [(1, (1, 1, 1)), (2, (2, 3, 5)), (8, (1, 2, 4)), (4, (8, 16, 32))])
def test_depth_and_spp_settings(oracle, depth, spp):
    size = (128, 72)
    cfg = cfg_foveated(9, 26, spp, max_depth=depth)
    model = scenes.atrium(5000)
    r = make_gpu(model, scenes.sky_probe(), scenes.ATRIUM_CAMERA, size, cfg)
    r.render()
    S, F = make_oracle(oracle, model, scenes.sky_probe(), scenes.ATRIUM_CAMERA, size)
    cnt = oracle.render(S, F, cfg)
    assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame)
    assert r.stats().paths == cnt[2]
    r.close()


def test_resize_and_probe_replacement(oracle):
    model = scenes.cornell_box()
    cfg = cfg_uniform(2, 4)
    r = make_gpu(model, scenes.sky_probe(), scenes.CORNELL_CAMERA, (64, 48), cfg)
    r.render()
    small = r.downloadAccum()
    # window resized: new buffers, aspect ratio recomputed by setCamera (SimplePathtracer.cpp:286)
    r.resize((200, 120))
    r.setCamera(renderer.Camera(**{"eye": scenes.CORNELL_CAMERA["eye"], "lookat": scenes.CORNELL_CAMERA["lookat"],
                                   "up": scenes.CORNELL_CAMERA["up"], "fovY": scenes.CORNELL_CAMERA["fovy"]}))
    r.setProbe(renderer.ProbeData(scenes.ambient_probe(200, 120, 1.5)).BuildCDF())
    r.launchParams.frame.subframe_index = 0
    r.render()
    S, F = make_oracle(oracle, model, scenes.ambient_probe(200, 120, 1.5), scenes.CORNELL_CAMERA, (200, 120))
    oracle.render(S, F, cfg)
    assert small.shape == (48, 64, 4)
    assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame)
    r.close()


def test_launch_equals_render_for_the_uniform_branch(oracle):
    """fovpt_launch with the FOV_OFF parameters == fovpt_render with config.uniform (SimplePathtracer.cpp:85-131)."""
    size = (96, 64)
    model, probe = scenes.cornell_box(), scenes.sky_probe()
    a = make_gpu(model, probe, scenes.CORNELL_CAMERA, size, cfg_uniform(3, 4))
    a.render()
    b = make_gpu(model, probe, scenes.CORNELL_CAMERA, size, abi.Config.reference_default())
    f = b.launchParams.frame
    f.factor.x = f.factor.y = f.factor.z = 1
    f.fillSize, f.r_inner, f.r_outer, f.redraw = 1, 0.0, 1e9, 0
    f.offset.x = f.offset.y = 0
    b.launchParams.samples_per_launch = 3
    b.launch(size[0], size[1])
    b.synchronize()
    assert _eq(a.downloadAccum(), b.downloadAccum()) and np.array_equal(a.downloadPixels(), b.downloadPixels())
    a.close(); b.close()


def test_hdr_probe_and_non_monotone_cdf_fallback(oracle):
    """A probe with negative texels makes the row CDFs non-monotone: the library must then use the plain
    binary search (Probe.cuh:119-136) and still agree with the oracle."""
    size = (96, 64)
    data = scenes.sky_probe(48, 24)
    data[5:9, 10:20, :3] *= -0.5
    model = scenes.cornell_box()
    cfg = cfg_uniform(2, 3)
    r = make_gpu(model, data, scenes.CORNELL_CAMERA, size, cfg)
    r.render()
    S, F = make_oracle(oracle, model, data, scenes.CORNELL_CAMERA, size)
    oracle.render(S, F, cfg)
    assert _eq(r.downloadAccum(), F.accum)
    r.close()


def test_build_cdf_on_device_matches_host_and_oracle(oracle):
    """fovpt_set_probe_data: ProbeData::BuildCDF on the GPU, rows in parallel, left-to-right within a row.  The kernel works in
    tiles of 8 rows x 256 columns: sizes below, at and across a tile, widths that are not a multiple of 4, a single row."""
    rng = np.random.default_rng(17)
    ragged = [rng.random((h, w, 4), dtype=np.float32) * np.float32(3.0) for (w, h) in ((1, 1), (7, 3), (255, 9), (256, 8), (257, 17), (773, 31), (1030, 5), (5, 4100))]
    for data in [scenes.sky_probe(96, 40, seed=3), scenes.ambient_probe(320, 180, 2.5)] + ragged:
        h, w = data.shape[:2]
        r = renderer.SampleRenderer(scenes.cornell_box())
        p = r.setProbeData(data)
        got = [r.download(ptr, np.empty(shape, np.float32)) for ptr, shape in
               ((p.pdfValuesX, (h, w)), (p.cdfValuesX, (h, w)), (p.pdfValuesY, (h,)), (p.cdfValuesY, (h,)))]
        r.close()
        host = renderer.ProbeData(data).BuildCDF()
        want = (host.pdfValuesX, host.cdfValuesX, host.pdfValuesY, host.cdfValuesY)
        ora = oracle.build_cdf(data)
        for g, hst, o in zip(got, want, ora):
            assert _eq(g, hst) and _eq(g, o)


def test_torch_external_stream_orders_after_the_frame():
    """bench.py hands the frame to torch.distributed under torch.cuda.ExternalStream(library stream): work
    queued on that stream must see the finished frame without any host synchronisation."""
    import torch
    size = (256, 144)
    r = make_gpu(scenes.atrium(8000), scenes.ambient_probe(64, 36, 2.5), scenes.ATRIUM_CAMERA, size, cfg_foveated(20, 64, (1, 2, 8)))
    frame = torch.zeros(size[0] * size[1], dtype=torch.int32, device="cuda")
    ext = torch.cuda.ExternalStream(r.stream)
    copies = []
    for _ in range(3):
        r.launchParams.frame.subframe_index = 0
        r.launchParams.frame.frame_buffer = frame.data_ptr()
        with torch.cuda.stream(ext):
            r.render_async()
            copies.append(frame.clone())           # enqueued right behind the frame, on the same stream
            frame.zero_()                          # and the buffer is recycled at once
    ext.synchronize()
    r.launchParams.frame.subframe_index = 0
    r.render()
    want = torch.from_numpy(r.download(frame.data_ptr(), np.empty(size[0] * size[1], np.int32)))
    for c in copies:
        assert torch.equal(c.cpu(), want)
    assert int((want != 0).sum()) > 0.9 * want.numel()
    r.close()


@pytest.mark.parametrize("accumulate", [0, 1])
def test_frames_in_flight_do_not_change_the_frames(accumulate):
    """fovpt_config.frames_in_flight / chains_per_frame: with two frames in flight the main chains of consecutive frames run
    beside each other on two stream pairs, with two chains per frame the halves of ONE frame do; every frame must still come
    out as with one -- resolves in issue order (progressive accumulation
    makes frame k depend on frame k-1), and work the caller queues on fovpt_stream() between two frames ordered between them
    (each frame is copied out and the ONE frame buffer cleared right behind it, six frames back to back, no host sync)."""
    import os
    import torch
    size = (320, 180)
    n = 10                                                                           # (every state set of the rotation is used again)
    got = {}
    for fif in (1, 2, 0, 23):
        cfg = cfg_foveated(24, 80, (1, 2, 4))
        cfg.accumulate = accumulate
        cfg.frames_in_flight = 2 if fif == 23 else fif if fif > 0 else 1
        cfg.chains_per_frame = 2 if fif == 0 else 1                                  # "0": one frame at a time, as two chains
        # "23": two frames in flight over THREE state sets instead of the default four (round 4: a lane's next job no longer waits for
        # the resolve of its previous one; FOVPT_SETS is read when the context is created)
        if fif == 23:
            os.environ["FOVPT_SETS"] = "3"
        try:
            r = make_gpu(scenes.atrium(9000), scenes.sky_probe(), scenes.ATRIUM_CAMERA, size, cfg, gaze=(200, 70))
        finally:
            os.environ.pop("FOVPT_SETS", None)
        frame = torch.zeros(size[0] * size[1], dtype=torch.int32, device="cuda")
        ext = torch.cuda.ExternalStream(r.stream)
        copies = []
        cam = scenes.ATRIUM_CAMERA
        for k in range(n):
            eye = list(cam["eye"]); eye[0] += 3.0 * k * (1 - accumulate)         # a moving camera unless the frames accumulate
            r.setCamera(renderer.Camera(eye, cam["lookat"], cam["up"], cam["fovy"], size[0] / size[1]))
            r.launchParams.frame.subframe_index = k if accumulate else 0
            r.launchParams.frame.frame_buffer = frame.data_ptr()
            with torch.cuda.stream(ext):
                r.render_async()
                copies.append(frame.clone())
                frame.zero_()
        ext.synchronize()
        got[fif] = ([c.cpu() for c in copies], r.downloadAccum().copy())
        r.close()
    for other in (2, 0, 23):
        for a, b in zip(got[1][0], got[other][0]):
            assert torch.equal(a, b) and int((a != 0).sum()) > 0.9 * a.numel()
        assert _eq(got[1][1], got[other][1])
    if not accumulate:
        assert not torch.equal(got[1][0][0], got[1][0][-1])                       # (the camera did move)


def test_bench_two_rank_rehearsal_gathers_the_full_frame():
    """bench.py's N > 1 path end to end on ONE GPU: two ranks share device 0 and talk over gloo
    (FOVPT_BENCH_REHEARSAL; RCCL refuses two ranks per device).  Rank 0's gathered frame must equal the
    unsharded frame bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    env = dict(os.environ, FOVPT_BENCH_REHEARSAL="1", FOVPT_BENCH_CHECK="1", MASTER_ADDR="127.0.0.1")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))                               # a free port, not a fixed one (no EADDRINUSE on a busy box)
    port = sock.getsockname()[1]
    sock.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "6", "--warmup", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["gather_mismatch_vs_single_gpu"] == 0
    assert out["config"]["rays_per_frame"] > 3.0e6        # both ranks' rays are counted


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["FOVPT_ROOT"], "tests"))
from fovpathtracing_optixcodelatest_amd import multigpu, scenes
from common import cfg_foveated, make_gpu
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
size = (320, 200)
cfg = cfg_foveated(24, 80, (1, 2, 6))
r = make_gpu(scenes.atrium(8000), scenes.sky_probe(), scenes.ATRIUM_CAMERA, size, cfg, gaze=(250, 60))
r.render()
want = torch.from_numpy(r.downloadPixels().reshape(-1).view(np.int32).copy())
pg = multigpu.PackedGather(r, dev, dst=0, nbuffers=2, force_collective=True)
counts = pg.plan()
frames = [torch.zeros(size[0] * size[1], dtype=torch.int32, device="cuda") for _ in range(2)]
ext = torch.cuda.ExternalStream(r.stream, device=dev)
works, done = [], []
with torch.cuda.stream(ext):
    for k in range(4):                                   # bench.py's loop: render, pack + RCCL gather, retire two frames later
        b = k % 2
        if k >= 2:
            works[k - 2].wait()
            pg.finish(frames[b], b)
            done.append(frames[b].clone())               # enqueued behind the unpack; compared after the loop
            frames[b].zero_()
        r.launchParams.frame.subframe_index = 0
        r.launchParams.frame.frame_buffer = frames[b].data_ptr()
        r.render_async()
        w = pg.gather(frames[b], b, async_op=True)
        assert w is not None                             # a real collective, not the group-of-one shortcut
        works.append(w)
        frames[b].zero_()                                # only the gathered buffers carry the pixels from here on
    for k in (2, 3):
        works[k].wait()
        pg.finish(frames[k % 2], k % 2)
    # the older full-frame reduce through RCCL as well
    f = want.to("cuda").clone()
    w = multigpu.gather_frame(f, dst=0, async_op=True, force_collective=True)
    w.wait()
ext.synchronize()
torch.cuda.synchronize()
for k, d in enumerate(done + frames):
    assert torch.equal(d.cpu(), want), "frame %d" % k
assert torch.equal(f.cpu(), want)
r.close()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK", counts)
"""


def test_rccl_group_of_one_runs_the_packed_gather_on_the_library_stream():
    """The collective half of the multi-GPU gather on the real backend: RCCL is initialised (a group of one -- this box has
    one GPU, and RCCL refuses two ranks per device), and bench.py's loop shape -- render, HIP pack, `dist.gather` under
    torch.cuda.ExternalStream(fovpt_stream()), wait, HIP unpack two frames later -- reproduces the frame bit for bit with no host
    synchronisation in between."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FOVPT_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], capture_output=True, text=True, env=env, timeout=300, cwd=root)
    assert res.returncode == 0 and "RCCL-ONE-RANK-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize("uniform", [False, True])
def test_chunked_jobs_for_large_launches(oracle, monkeypatch, uniform):
    """Launches whose sample slots exceed the per-job budget (the reference's own 3840x2160 x 32 spp FOV_OFF
    benchmark has 265 M) are cut into row chunks run in launch order.  Forced here with a tiny budget."""
    monkeypatch.setenv("FOVPT_SLOT_BUDGET", "3000")
    size = (160, 96)
    cfg = cfg_uniform(3, 4) if uniform else cfg_foveated(10, 30, (1, 2, 8))
    model = scenes.atrium(5000)
    r = make_gpu(model, scenes.sky_probe(), scenes.ATRIUM_CAMERA, size, cfg)
    r.render()
    S, F = make_oracle(oracle, model, scenes.sky_probe(), scenes.ATRIUM_CAMERA, size)
    cnt = oracle.render(S, F, cfg)
    assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame)
    st = r.stats()
    assert st.paths == cnt[2] and st.frames == 1
    r.close()
    # sharded + chunked: the shards still add up
    monkeypatch.setenv("FOVPT_SLOT_BUDGET", "5000")
    total = np.zeros((size[1], size[0]), np.uint64)
    for rank in range(2):
        c = cfg.copy()
        c.rank, c.world = rank, 2
        rr = make_gpu(model, scenes.sky_probe(), scenes.ATRIUM_CAMERA, size, c)
        rr.render()
        total += rr.downloadPixels()
        rr.close()
    assert np.array_equal(total.astype(np.uint32), F.frame)


@pytest.mark.parametrize("budget", [None, "4000"])
def test_accumulating_tile_shards_sum_to_the_oracle(oracle, monkeypatch, budget):
    """accumulate = 1 + world > 1, unchunked and cut into chunks by a small slot budget: every rank blends
    its owned periphery pixels with ITS history of them (static gaze: the owner of a pixel does not change), foreign
    pixels and -- on ranks other than 0 -- holes stay zero, so the shards add up to the oracle's frame after each of
    several progressive frames."""
    if budget:
        monkeypatch.setenv("FOVPT_SLOT_BUDGET", budget)
    size = (160, 96)
    model, probe = scenes.atrium(5000), scenes.sky_probe()
    cfg = cfg_foveated(10, 30, (1, 2, 4))
    cfg.accumulate = 1
    world = 3
    ranks = []
    for rank in range(world):
        c = cfg.copy()
        c.rank, c.world = rank, world
        ranks.append(make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, c))
    S, F = make_oracle(oracle, model, probe, scenes.ATRIUM_CAMERA, size)
    for k in range(3):
        oracle.render(S, F, cfg)
        sa = np.zeros_like(F.accum)
        sf = np.zeros(F.frame.shape, np.uint64)
        for r in ranks:
            r.render()
            sa += r.downloadAccum()
            sf += r.downloadPixels()
        assert _eq(sa, F.accum), k
        assert np.array_equal(sf.astype(np.uint32), F.frame), k
    for r in ranks:
        r.close()


def test_three_separate_launches_per_frame_with_tile_shards():
    """A caller that issues the P, M and F passes as three fovpt_launch calls (= three optixLaunch, SimplePathtracer.cpp:
    148-209) with world > 1: a later launch must not wipe what this rank wrote in an earlier one; the
    per-rank frames still sum to the single-GPU frame."""
    size = (192, 108)
    W, H = size
    model, probe = scenes.atrium(8000), scenes.ambient_probe(96, 54, 2.5)
    ri, ro = 15, 48
    cfg = cfg_foveated(ri, ro, (1, 2, 8))
    full = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, cfg)
    full.render()
    fa, ff = full.downloadAccum(), full.downloadPixels()
    full.close()

    def three_launches(r):
        lp, f = r.launchParams, r.launchParams.frame
        f.subframe_index = 0
        f.factor.x, f.factor.y, f.factor.z, f.fillSize = 4, 4, 1, 4
        f.r_inner, f.r_outer, f.offset.x, f.offset.y, f.redraw = float(ro), 1e9, 0, 0, 0
        lp.samples_per_launch = 1
        r.launch(W // 4, H // 4)
        f.factor.x, f.factor.y, f.fillSize = 2, 2, 2
        f.r_inner, f.r_outer, f.offset.x, f.offset.y, f.redraw = float(ri), float(ro + 2), W // 2 - (ro + 2), H // 2 - (ro + 2), 1
        lp.samples_per_launch = 2
        r.launch(ro + 2, ro + 2)
        f.factor.x, f.factor.y, f.fillSize = 1, 1, 1
        f.r_inner, f.r_outer, f.offset.x, f.offset.y = 0.0, float(ri + 1), W // 2 - (ri + 1), H // 2 - (ri + 1)
        lp.samples_per_launch = 8
        r.launch(2 * (ri + 1), 2 * (ri + 1))
        r.synchronize()

    for world in (1, 2, 4):
        sa = np.zeros_like(fa)
        sf = np.zeros(ff.shape, np.uint64)
        for rank in range(world):
            c = cfg.copy()
            c.rank, c.world = rank, world
            r = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, c)
            three_launches(r)
            sa += r.downloadAccum()
            sf += r.downloadPixels()
            r.close()
        assert _eq(sa, fa), world
        assert np.array_equal(sf.astype(np.uint32), ff), world


@pytest.mark.parametrize("world", [2, 3, 8])
def test_packed_gather_plan_pack_unpack(world):
    """fovpt_gather_plan / _pack / _unpack (the HIP side of the multi-GPU gather): every rank computes the same plan, the
    plan covers exactly the pixels some launch index writes, and packing every rank's frame and scattering the buffers on
    the root reproduces the unsharded frame bit for bit -- with gaze off-centre so that the rings hang over the border."""
    import torch
    size = (200, 120)
    W, H = size
    model, probe = scenes.atrium(6000), scenes.sky_probe()
    cfg = cfg_foveated(14, 44, (1, 2, 4))
    gaze = (150, 40)
    full = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, cfg, gaze=gaze)
    full.render()
    want, want_acc = full.downloadPixels(), full.downloadAccum()
    full.close()
    plans, packed, root = [], [], None
    for rank in range(world):
        c = cfg.copy()
        c.rank, c.world = rank, world
        r = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, c, gaze=gaze)
        counts = r.gather_plan()
        assert counts == r.gather_plan()                              # cached, stable
        plans.append(counts)
        r.render()
        stride = (max(counts) + 63) // 64 * 64
        buf = torch.full((stride,), -1, dtype=torch.int32, device="cuda")
        r.gather_pack(r.launchParams.frame.frame_buffer, buf.data_ptr())
        r.synchronize()
        packed.append(buf.cpu())
        if rank == 0:
            root = r
        else:
            r.close()
    assert all(p == plans[0] for p in plans)
    assert sum(plans[0]) == int((want_acc[..., 3] == 1).sum())          # exactly the pixels that have a writer
    assert max(plans[0]) - min(plans[0]) < 0.25 * max(plans[0])           # interleaved tiles balance the pixel counts too
    stride = packed[0].numel()
    gathered = torch.stack(packed).to("cuda")
    target = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    root.gather_unpack(gathered.data_ptr(), stride, target.data_ptr())
    root.synchronize()
    got = target.cpu().numpy().view(np.uint32).reshape(H, W)
    assert np.array_equal(got, want)
    # uniform frames (FOV_OFF) partition too
    cu = cfg_uniform(1, 2)
    cu.rank, cu.world = 0, world
    root.config = cu
    counts = root.gather_plan()
    assert sum(counts) == W * H and max(counts) - min(counts) <= 0.15 * max(counts)
    root.close()


@pytest.mark.parametrize("world", [2, 5])
def test_packed_gather_of_a_chunked_frame(world, monkeypatch):
    """above the slot budget every pass of a frame runs as jobs of its own, and the tile ownership of
    the middle and fovea passes must still rotate with the pass's place in the FRAME (the gather plan is made for the
    frame): pack / unpack of chunked, sharded frames is the unsharded, unchunked frame."""
    import torch
    size = (200, 120)
    W, H = size
    model, probe = scenes.atrium(6000), scenes.sky_probe()
    cfg = cfg_foveated(14, 44, (1, 2, 4))
    gaze = (90, 70)
    full = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, cfg, gaze=gaze)
    full.render()
    want = full.downloadPixels()
    full.close()
    monkeypatch.setenv("FOVPT_SLOT_BUDGET", "700")            # >= one launch row of every pass (50, 92, 120 slots), << a pass
    packed, root, plans = [], None, []
    for rank in range(world):
        c = cfg.copy()
        c.rank, c.world = rank, world
        r = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, c, gaze=gaze)
        counts = r.gather_plan()
        plans.append(counts)
        r.render()
        stride = (max(counts) + 63) // 64 * 64
        buf = torch.full((stride,), -1, dtype=torch.int32, device="cuda")
        r.gather_pack(r.launchParams.frame.frame_buffer, buf.data_ptr())
        r.synchronize()
        packed.append(buf.cpu())
        if rank == 0:
            root = r
        else:
            r.close()
    assert all(p == plans[0] for p in plans)
    gathered = torch.stack(packed).to("cuda")
    target = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    root.gather_unpack(gathered.data_ptr(), packed[0].numel(), target.data_ptr())
    root.synchronize()
    root.close()
    got = target.cpu().numpy().view(np.uint32).reshape(H, W)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("options", [abi.OPT_SKY_MISS, abi.OPT_RUSSIAN_ROULETTE, abi.OPT_SKY_MISS | abi.OPT_RUSSIAN_ROULETTE])
def test_opt_in_extensions_match_the_oracle(oracle, options):
    """fovpt_config.options: sky radiance for escaped secondary rays (the MIS counterpart PT_sv5_ carries commented out,
    deviceProgram.cu:259-269) and Russian roulette (its //!TODO at :518-520).  Neither exists in the reference; the oracle
    implements them the same way and the GPU must agree with it bit for bit, ray counts included."""
    size = (192, 108)
    model, probe = scenes.atrium(12000), scenes.sky_probe(96, 48, seed=3)
    cfg = cfg_foveated(15, 48, (1, 2, 8), max_depth=6)
    cfg.options = options
    r = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, cfg)
    r.render()
    S, F = make_oracle(oracle, model, probe, scenes.ATRIUM_CAMERA, size)
    cnt = oracle.render(S, F, cfg)
    assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame)
    st = r.stats()
    assert (st.radiance_rays, st.shadow_rays, st.paths) == (cnt.lib_radiance, cnt.lib_shadow, cnt[2])
    # and they do something: the plain frame differs
    plain = cfg.copy()
    plain.options = 0
    r.config = plain
    r.launchParams.frame.subframe_index = 0
    r.render()
    assert not _eq(r.downloadAccum(), F.accum)
    if options & abi.OPT_RUSSIAN_ROULETTE:
        assert r.stats().radiance_rays - st.radiance_rays > st.radiance_rays          # (stats accumulate: the plain frame traced more)
    r.close()


def test_stereo_asymmetric_frusta(oracle):
    """Two eyes = two cameras and two render() calls per frame (BASELINE.json configs[4]); per-eye off-centre
    frusta enter only through camera U, V, W, so the oracle sees the same LaunchParams."""
    size = (128, 128)
    model, probe = scenes.atrium(5000), scenes.sky_probe()
    cfg = cfg_foveated(10, 30, (1, 2, 4), max_depth=8)
    r = make_gpu(model, probe, scenes.ATRIUM_CAMERA, size, cfg)
    S, F = make_oracle(oracle, model, probe, scenes.ATRIUM_CAMERA, size)
    fwd = np.array(scenes.ATRIUM_CAMERA["lookat"], np.float64) - np.array(scenes.ATRIUM_CAMERA["eye"], np.float64)
    for eye_dx, (al, ar) in ((-3.2, (-0.9, 0.7)), (3.2, (-0.7, 0.9))):
        eye = (scenes.ATRIUM_CAMERA["eye"][0], scenes.ATRIUM_CAMERA["eye"][1], scenes.ATRIUM_CAMERA["eye"][2] + eye_dx)
        r.setCameraFov(eye, fwd, (0, 1, 0), al, ar, 0.8, -0.8)
        F.lp.camera = r.launchParams.camera
        for lp in (r.launchParams, F.lp):
            lp.frame.subframe_index = 0
        r.render()
        oracle.render(S, F, cfg)
        assert _eq(r.downloadAccum(), F.accum)
    # a symmetric XrFovf equals sutil::Camera::UVWFrame up to rounding
    r.setCameraFov(scenes.ATRIUM_CAMERA["eye"], fwd, (0, 1, 0), -0.5, 0.5, 0.5, -0.5)
    a = np.array(r.launchParams.camera.U.tolist() + r.launchParams.camera.V.tolist())
    r.setCamera(renderer.Camera(scenes.ATRIUM_CAMERA["eye"], scenes.ATRIUM_CAMERA["lookat"], (0, 1, 0), np.degrees(1.0), 1.0))
    wlen = np.linalg.norm(fwd)
    b = np.array(r.launchParams.camera.U.tolist() + r.launchParams.camera.V.tolist()) / wlen
    assert np.allclose(a, b, rtol=1e-4, atol=1e-5)
    r.close()


_FUZZ_LC = range(int(os.environ.get("FOVPT_FUZZLC_FROM", "0")), int(os.environ.get("FOVPT_FUZZLC_TO", "24")))


@pytest.mark.parametrize("seed", _FUZZ_LC)
def test_random_lifecycle(oracle, seed):
    """One renderer through a random sequence of what an application does between frames: resize, new probe,
    new camera, new gaze point, other radii / sample counts / depth / mode, subframe index set or left running,
    asynchronous frames back to back (the two state sets alternate) -- after every frame the image is the oracle's."""
    rng = np.random.default_rng(12000 + seed)
    model = scenes.cornell_box() if seed % 2 else scenes.atrium(int(rng.integers(500, 4000)), seed=3 + seed)
    base_cam = scenes.CORNELL_CAMERA if seed % 2 else scenes.ATRIUM_CAMERA
    S = oracle.OracleScene(model)
    size = (int(rng.integers(30, 120)), int(rng.integers(20, 90)))
    probe_data = scenes.sky_probe()
    cfg = cfg_foveated(8, 24, (1, 2, 4), max_depth=3)
    cam = dict(base_cam)
    r = make_gpu(model, probe_data, cam, size, cfg)
    F = oracle.OracleFrame(size[0], size[1], oracle.HostProbe(probe_data), cam)

    def set_camera():
        r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], size[0] / float(size[1])))
        U, V, W = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], size[0] / float(size[1]))
        F.lp.camera.eye.set(cam["eye"]); F.lp.camera.U.set(U); F.lp.camera.V.set(V); F.lp.camera.W.set(W)

    for step in range(10):
        op = int(rng.integers(0, 7))
        if op == 0:                                                   # window resized: buffers start from zero
            size = (int(rng.integers(30, 120)), int(rng.integers(20, 90)))
            sub, gaze = int(F.lp.frame.subframe_index), (int(F.lp.frame.c.x), int(F.lp.frame.c.y))
            r.resize(size)
            F = oracle.OracleFrame(size[0], size[1], oracle.HostProbe(probe_data), cam, gaze=gaze, subframe_index=sub)
            r.launchParams.frame.subframe_index = sub
            set_camera()
        elif op == 1:                                                 # other environment
            probe_data = (scenes.sky_probe(), scenes.ambient_probe(size[0], size[1], 2.5), scenes.ambient_probe(24, 12, 0.8))[int(rng.integers(0, 3))]
            if rng.random() < 0.5:
                r.setProbe(renderer.ProbeData(probe_data).BuildCDF())            # BuildCDF on the host (as the reference)
            else:
                r.setProbeData(probe_data)                                        # BuildCDF on the device
            hp = oracle.HostProbe(probe_data)
            F.probe = hp; F.lp.probe = hp.struct
        elif op == 2:                                                 # camera moved
            e = np.asarray(base_cam["eye"], np.float64)
            cam = dict(base_cam, eye=tuple(float(x) for x in e + rng.uniform(-40, 40, 3)), fovy=float(rng.uniform(30, 60)))
            set_camera()
        elif op == 3:                                                 # other settings
            if rng.random() < 0.3:
                cfg = cfg_uniform(int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 5)))
            else:
                r_i = int(rng.integers(1, 25))
                cfg = cfg_foveated(r_i, r_i + int(rng.integers(1, 40)), tuple(int(x) for x in rng.integers(1, 5, 3)), max_depth=int(rng.integers(1, 5)))
            cfg.accumulate = int(rng.random() < 0.4)
            cfg.frames_in_flight = int(seed * 7 + step) % 3              # 0 (= the default, 2), 1 or 2: never changes a frame
            cfg.chains_per_frame = int(seed * 5 + step) % 3              # 0 / 1: one chain per frame, 2: two (frames >= 16384 slots)
            r.config = cfg
        elif op == 4:                                                 # the application resets the subframe counter
            sub = int(rng.integers(0, 3))
            for lp in (r.launchParams, F.lp):
                lp.frame.subframe_index = sub
        gaze = (int(rng.integers(0, size[0])), int(rng.integers(0, size[1])))
        for lp in (r.launchParams, F.lp):
            lp.frame.c.x, lp.frame.c.y = gaze
        frames = int(rng.integers(1, 4))                              # several frames in flight before anyone looks
        for _ in range(frames):
            r.render_async()
            oracle.render(S, F, cfg)
        r.synchronize()
        assert r.launchParams.frame.subframe_index == F.lp.frame.subframe_index, (seed, step, op, frames, bool(cfg.uniform))
        assert _eq(r.downloadAccum(), F.accum) and np.array_equal(r.downloadPixels(), F.frame), (seed, step, op, size)
    r.close()


@pytest.mark.gpu
def test_library_and_torch_share_one_hip_runtime_in_either_import_order():
    """libfovpt.so first and torch afterwards used to leave torch with "No HIP GPUs are available" (two HIP
    runtimes in one process, see lib._share_torch_hip_runtime); both orders must work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = ("from fovpathtracing_optixcodelatest_amd import renderer, scenes\n"
            "def lib_part():\n"
            "    r = renderer.SampleRenderer(scenes.cornell_box()); r.resize((64, 64)); cam = scenes.CORNELL_CAMERA\n"
            "    r.setCamera(renderer.Camera(cam['eye'], cam['lookat'], cam['up'], cam['fovy'], 1.0))\n"
            "    r.setProbe(renderer.ProbeData(scenes.ambient_probe(64, 32, 0.2)).BuildCDF()); r.render(); return int(r.downloadPixels().sum())\n"
            "def torch_part():\n"
            "    import torch; return int(torch.arange(10, device='cuda').sum().item())\n")
    for order in ("a = lib_part(); b = torch_part()", "b = torch_part(); a = lib_part()"):
        out = subprocess.run([sys.executable, "-c", body + order + "\nprint('ok', a > 0, b)"], cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ok True 45" in out.stdout, (order, out.stdout[-500:], out.stderr[-1500:])
