#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on N MI355X GPUs of one node.

Metric : Mray/s (radiance + occlusion rays actually traced per second, whole job) and ms/frame
Step   : one SampleRenderer::render() frame (three foveation passes) of workload C3:
         Sponza-class procedural atrium (~262 k triangles), 1920x1080, foveated 8/2/1 spp
         (fovea / middle ring / periphery), radii 148/482, gaze at the frame centre, full Disney
         BSDF + probe NEE, depth cap 4, ambient probe 2.5 at frame resolution.  All inputs are
         synthetic and resident in HBM before the timed region.
N > 1  : ONE frame is sharded by interleaved launch-index tiles over the ranks (strong scaling);
         each rank renders its tiles into a full-size zeroed frame and the frames are summed onto
         rank 0 with one RCCL reduce over xGMI (the "gather": owned pixel sets are disjoint).

Prints ONE JSON line on rank 0 (see the contract in the task description), carrying `roofline`
(dominant kernel, algorithmic bytes / HIP-event time) and, at N=1, `cpu_baseline` (the CPU
oracle timed on the host cores on the same frame).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
R_INNER, R_OUTER = 148, 482          # the reference's own 2x radii (SimplePathtracer.cpp:20-21 comments)
SPP = (1, 2, 8)                      # periphery, middle, fovea
TARGET_TRIS = 262144
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def algorithmic_bytes_per_ray(num_tris, kind):
    """B_queue + B_trav + leaf triangles (+ hit shading for radiance hits is charged
    to the shade kernel, not to the traversal kernels measured here)."""
    b_queue = 128                                             # 64-B ray record written once + read once
    levels = max(1, math.ceil(math.log2(max(2.0, num_tris / 4.0))))
    b_trav = 64 * levels                                      # one root-to-leaf descent, 64-B nodes
    b_leaf = 4 * 48                                           # one leaf of 4 x 48-B triangle records
    b_out = 16 if kind == "closest" else 32                   # hit record write / accumulator read+write
    return b_queue + b_trav + b_leaf + b_out


def host_cpu_share():
    """CPU threads this job may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--advance-subframe", action="store_true",
                    help="let render() advance subframe_index from frame to frame (new P-pass seeds every frame) instead of "
                         "resetting it to 0 as the shipped application does (main.cpp:402-407)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch                     # before libfovpt: both then share one HIP runtime (same soname)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    # FOVPT_BENCH_REHEARSAL=1: all ranks share device 0 and talk over gloo -- a logic rehearsal of the
    # N > 1 path on a one-GPU box (RCCL refuses two ranks on one device); numbers are meaningless
    rehearsal = os.environ.get("FOVPT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
    from fovpathtracing_optixcodelatest_amd import multigpu

    model = scenes.atrium(TARGET_TRIS, seed=1234, material="app")
    probe_data = scenes.ambient_probe(W, H, 2.5)               # loadColor at frame resolution, main.cpp:175-187,229
    r = renderer.SampleRenderer(model, device=local_rank)
    r.resize((W, H))
    cam = scenes.ATRIUM_CAMERA
    r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / float(H)))
    probe = renderer.ProbeData(probe_data).BuildCDF()
    r.setProbe(probe)
    cfg = abi.Config.reference_default()
    cfg.r_inner, cfg.r_outer = R_INNER, R_OUTER
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = SPP
    cfg.max_depth = 4
    cfg.rank, cfg.world = rank, world
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2

    # the frame the ranks render into and gather: torch tensors, handed to the library as the
    # caller-owned render target (render(CUDAOutputBuffer&), SimplePathtracer.cpp:216-226).  Two of
    # them at N > 1: the RCCL reduce-gather of frame k runs beside the rendering of frame k+1.
    frames = [torch.zeros(H * W, dtype=torch.int32, device="cuda") for _ in range(2 if world > 1 else 1)]
    pending = [None, None]
    sync_gather = rehearsal
    step_no = [0]

    # torch's view of the stream on which the library's frames complete (fovpt_stream()): collectives issued
    # under it are ordered after the frame on the device, so the loop needs no host synchronisation
    lib_stream = torch.cuda.ExternalStream(r.stream, device=torch.device("cuda", local_rank))

    def step():
        k = step_no[0] % len(frames)
        step_no[0] += 1
        with torch.cuda.stream(lib_stream):
            if pending[k] is not None:                         # the gather that last used this buffer:
                pending[k].wait()                              # the library's stream waits for it (device side)
                pending[k] = None
            # the shipped app resets subframe_index to 0 before every render() (main.cpp:402-407)
            if not args.advance_subframe:
                r.launchParams.frame.subframe_index = 0
            r.launchParams.frame.frame_buffer = frames[k].data_ptr()
            r.render_async()
            if world > 1 and not sync_gather:                  # RCCL waits for the frame, then reduces beside frame k+1
                pending[k] = multigpu.gather_frame(frames[k], dst=0, async_op=True)
        if world > 1 and sync_gather:                          # gloo (rehearsal) knows nothing about HIP streams
            r.synchronize()
            multigpu.gather_frame(frames[k], dst=0)

    def fence():
        r.synchronize()
        with torch.cuda.stream(lib_stream):
            for k in range(len(pending)):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
        lib_stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather_mode = "none" if world == 1 else ("host-synchronised (rehearsal)" if sync_gather else "async, overlapped with the next frame")
    try:
        for _ in range(args.warmup):
            step()
        fence()
    except Exception as e:                                     # keep the run alive and say so in the JSON line
        if world == 1 or sync_gather:
            raise
        sys.stderr.write("bench.py: stream-ordered gather failed (%s); falling back to a host-synchronised gather\n" % (e,))
        sync_gather = True
        gather_mode = "host-synchronised (fallback: %s)" % type(e).__name__
        pending[0] = pending[1] = None
        for _ in range(max(1, args.warmup)):
            step()
        fence()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    last_frame_host = frames[(step_no[0] - 1) % len(frames)].cpu()      # rank 0: the gathered frame
    st = r.stats()
    rays_local = float(st.radiance_rays + st.shadow_rays)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total = float(rr.item())
    else:
        rays_total = rays_local
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_total / elapsed / 1e6

    # ---- latency of ONE frame (render() + synchronise, nothing else in flight): the timed loop above keeps
    # frames back to back, so the tail of frame k overlaps the head of frame k+1 (two state sets)
    lat_frames = max(5, min(50, args.steps))
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(lat_frames):
        r.launchParams.frame.subframe_index = 0
        r.render_async()
        r.synchronize()
    frame_latency_ms = (time.perf_counter() - t0) / lat_frames * 1e3

    # ---- roofline of the dominant kernel: separate profiled frames (hipEvents on the library's
    # own stream around every kernel; fovpt_stats accumulates them)
    cfg.profile = 1
    r.config = cfg
    prof_frames = max(5, min(50, args.steps))
    for _ in range(3):
        r.launchParams.frame.subframe_index = 0
        r.render_async()
    r.synchronize()
    r.reset_stats()
    for _ in range(prof_frames):
        r.launchParams.frame.subframe_index = 0
        r.render_async()
    r.synchronize()
    ps = r.stats()
    cfg.profile = 0
    r.config = cfg
    # k_traverse is the one traversal kernel (4 lanes per ray): closest-hit launches run on the main
    # stream (fovpt_stats books them under ms_trace), occlusion launches on the shadow stream
    # (ms_shadow); rocprofv3 reports both under the one kernel name
    dom = "k_traverse"
    ms_k = ps.ms_trace + ps.ms_shadow
    n_launch = ps.n_trace_launches + ps.n_shadow_launches
    n_rays = ps.radiance_rays + ps.shadow_rays
    b_closest = algorithmic_bytes_per_ray(int(ps.num_triangles), "closest")
    b_any = algorithmic_bytes_per_ray(int(ps.num_triangles), "any")
    b_ray = (b_closest * ps.radiance_rays + b_any * ps.shadow_rays) / max(1, n_rays)
    avg_ms = ms_k / max(1, n_launch)
    bytes_per_launch = b_ray * (n_rays / max(1, n_launch))
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "avg_launch_ms": round(avg_ms, 5), "launches_per_frame": n_launch / prof_frames,
        "rays_per_launch": n_rays / max(1, n_launch), "algorithmic_bytes_per_ray": round(b_ray, 1),
        "per_frame_ms": {"generate": ps.ms_generate / prof_frames, "traverse_closest": ps.ms_trace / prof_frames,
                         "traverse_occlusion_async": ps.ms_shadow / prof_frames,
                         "shade": ps.ms_shade / prof_frames, "resolve": ps.ms_resolve / prof_frames},
    }

    out = {
        "metric": "Mray/s", "value": round(mrays, 2), "unit": "Mray/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "frame_latency_ms": round(frame_latency_ms, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "C3: Sponza-class procedural atrium, %d tris, %dx%d, foveated spp fovea/mid/periphery %d/%d/%d, "
                        "radii %d/%d, gaze centre, Disney BSDF + probe NEE, depth 4, ambient probe 2.5 @ frame res"
                        % (model.num_triangles, W, H, SPP[2], SPP[1], SPP[0], R_INNER, R_OUTER),
            "triangles": model.num_triangles, "width": W, "height": H,
            "paths_per_frame": int(st.paths // max(1, args.steps)) if world == 1 else None,
            "rays_per_frame": rays_total / args.steps,
            "parallelism": "tile-shard x%d + RCCL reduce-gather" % world if world > 1 else "single GPU",
            "gather": gather_mode,
        },
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle (a scalar C++ port of the path) on the host cores, rank 0, N=1
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as orc
        orc.build()
        orc.set_math_mode(True)
        S = orc.OracleScene(model)
        hp = orc.HostProbe(probe_data, cdf=(probe.pdfValuesX, probe.cdfValuesX, probe.pdfValuesY, probe.cdfValuesY))
        F = orc.OracleFrame(W, H, hp, cam)
        cores = host_cpu_share()
        ocfg = cfg.copy()
        ocfg.rank, ocfg.world = 0, 1
        frames_done, rays_done = 0, 0
        t0 = time.perf_counter()
        while True:
            F.lp.frame.subframe_index = 0
            c = orc.render(S, F, ocfg, nthreads=cores)
            frames_done += 1
            rays_done += c[0] + c[1]
            if time.perf_counter() - t0 >= args.cpu_seconds or frames_done >= 50:
                break
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(rays_done / cpu_s / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "%d full C3 frame(s) (%.0f rays each, incl. the reference's discarded last segment) in %.1f s, "
                      "oracle/libfovpt_oracle.so, std::thread over %d threads" % (frames_done, rays_done / frames_done, cpu_s, cores),
            "ms_per_frame": round(cpu_s / frames_done * 1e3, 2),
        }
        # the timed frames are parity frames too: the oracle just rendered the same frame
        import numpy as np
        gpu_px = last_frame_host.numpy().view(np.uint32).reshape(H, W)
        if not args.advance_subframe:                        # (with advancing seeds the last timed frame is not frame 0)
            out["parity_vs_oracle_rgba8_mismatch"] = int((gpu_px != F.frame).sum())

    if world > 1 and rank == 0 and os.environ.get("FOVPT_BENCH_CHECK") == "1":
        # the gathered frame on rank 0 must be the unsharded frame, bit for bit
        gathered = last_frame_host
        c1 = cfg.copy()
        c1.rank, c1.world, c1.profile = 0, 1, 0
        r.config = c1
        solo = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        r.launchParams.frame.subframe_index = 0
        r.launchParams.frame.frame_buffer = solo.data_ptr()
        r.render()
        out["gather_mismatch_vs_single_gpu"] = int((gathered != solo.cpu()).sum())
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
