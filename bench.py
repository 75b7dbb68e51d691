#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on N MI355X GPUs of one node.

Metric : Mray/s (radiance + occlusion rays actually traced per second, whole job) and ms/frame
Step   : one SampleRenderer::render() frame (three foveation passes) of workload C3 (BASELINE.json configs[2]):
         Sponza-class procedural atrium (~262 k triangles), 1920x1080, foveated 8/2/1 spp (fovea / middle ring /
         periphery), radii 148/482, gaze at the frame centre, full Disney BSDF + probe NEE, depth cap 4, ambient probe 2.5
         at frame resolution, subframe_index reset to 0 before every frame as the shipped application does
         (PT_sv5_/main.cpp:402-407).  All inputs are synthetic and resident in HBM before the timed region.
N > 1  : ONE frame is sharded by interleaved launch-index tiles over the ranks (strong scaling); each rank renders its
         tiles, packs the pixels it owns (HIP) and RCCL gathers the packed buffers onto rank 0 over xGMI, which scatters
         them into the frame (fovpt_gather_*; --gather reduce = the older full-frame sum-reduce; --gather lib = the library's OWN RCCL
         transport, fovpt_comm_init / fovpt_gather_frame, what a C++ host uses).

Prints ONE JSON line on rank 0 (contract in the task description) carrying `value` (frames back to back, two in flight) and
`value_sync_per_frame` (one frame at a time, render() + synchronise: SURVEY 8(d)'s own definition of the metric), and
  roofline      dominant kernel k_traverse: algorithmic bytes / HIP-event time AS THE TIMED REGION RUNS (the contract's figure), `bound` =
                what the evidence says binds (the CUs' L1 address / data path, with the node step's dependent chain on top: neither HBM
                nor the vector ALU), flat keys `l1_frac` (texture-addresser busy time of a frame's loads / frame interval), `lane_use`,
                `hbm_frac_measured`; `traffic`, `valu` and `l1_path` measured IN THIS RUN
                by rocprofv3 --pmc child passes of this same script (or, failing that, imported from profiles/ and said so),
                per-kernel times both overlapped (as the frame runs) and serialised (every kernel alone)
  cpu_baseline  the CPU oracle timed on the host cores on the same frame (N = 1)
  variants      the same measurement without the benchmark-shaped shortcuts: a non-constant HDR probe at frame
                resolution, advancing subframe + moving camera and gaze, and 1 M / 3.8 M-triangle scenes (N = 1)
"""
import argparse
import csv
import glob
import json
import math
import os
import re
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
R_INNER, R_OUTER = 148, 482          # the reference's own 2x radii (SimplePathtracer.cpp:20-21 comments)
SPP = (1, 2, 8)                      # periphery, middle, fovea
TARGET_TRIS = 262144
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
SIMDS, CLOCK_HZ = 1024, 2.4e9        # 256 CUs x 4 SIMDs, 2.4 GHz
# A SIMD issues one wave64 vector instruction per 2 cycles (32 lanes per cycle: 157.3 TFLOP/s fp32 = 1024 SIMDs x 32 x 2 x 2.4 GHz,
# MI355X_MICROARCH.md); one wave alone gets one per 4.  Measured here with independent fmas at 8 waves per SIMD: 0.98 per ns
# = 2.45 cycles (tools/micro/valu_rate.hip).  Round 1 and most of round 2 priced issue time at 4 cycles -- twice too much.
VALU_CYCLES_PER_INST = 2.0
VALU_MEASURED_INST_PER_NS_PER_SIMD = 0.98


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


# ---- hardware counters, measured in this run: rocprofv3 --pmc passes over a short child run of this script ----------
PMC_PASSES = {
    "fetch": ["FETCH_SIZE"],                                  # TCC: 3 of the 4 slots -> its own pass (MI355X_MICROARCH.md, PMC slots)
    "write": ["WRITE_SIZE"],
    "valu": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VMEM_RD"],
    "ta": ["TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum"],      # two TA counters per pass fit the block's slots; four do not (error 38)
}
CUS = 256


def run_counter_passes(timeout_s=90):
    """-> ({kernel: {counter: mean per dispatch, "launches": n, "dur_us": mean serialised duration}}, note)"""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    per_kernel = {}
    for tag, counters in PMC_PASSES.items():
        d = tempfile.mkdtemp(prefix="fovpt_pmc_%s_" % tag, dir=os.environ.get("TMPDIR", "/tmp"))
        cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                                           sys.executable, os.path.abspath(__file__), "--child-frames", "3"]
        # The pass runs in a process group of its own: on a timeout the WHOLE group is killed -- rocprofv3 and the profiled
        # python child -- so that nothing of it still holds the GPU when the timed region starts.
        try:
            proc = subprocess.Popen(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                                    env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")))
        except Exception as e:                                  # the run goes on without counters and says so
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass '%s' failed: %s" % (tag, type(e).__name__)
        try:
            proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)                    # the exact group this function started, nothing else
                except ProcessLookupError:
                    break
                try:
                    proc.communicate(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass '%s' timed out after %d s (its process group was killed)" % (tag, timeout_s)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if proc.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass '%s' gave no counters (rc %d)" % (tag, proc.returncode)
        disp = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                m = re.search(r"\b(k_[a-z_0-9]+)(?:<[^>]*>)?\s*\(", row["Kernel_Name"])
                if not m:
                    continue
                dd = disp.setdefault((f, row["Dispatch_Id"]), {"k": m.group(1)})
                dd[row["Counter_Name"]] = float(row["Counter_Value"])
                if "Start_Timestamp" in row and row.get("End_Timestamp"):
                    dd["dur_us"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        for dd in disp.values():
            k = per_kernel.setdefault(dd["k"], {})
            for name, v in dd.items():
                if name == "k":
                    continue
                acc = k.setdefault(name, [0.0, 0])
                acc[0] += v
                acc[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, cs in per_kernel.items():
        out[k] = {name: acc[0] / max(1, acc[1]) for name, acc in cs.items()}
        out[k]["launches_seen"] = max(acc[1] for acc in cs.values())
    return out, "rocprofv3 --pmc child passes of this run (kernels serialised by the profiler)"


FRAMES_IN_FLIGHT = [0]          # --frames-in-flight (0 = the library's default)


def build_renderer(renderer, scenes, abi, tris, local_rank, rank, world, probe_kind="constant", seed=1234, scene="atrium"):
    model = scenes.atrium(tris, seed=seed, material="app") if scene == "atrium" else scenes.street(tris, material="app")
    # loadColor at frame resolution (main.cpp:175-187,229), or a seeded non-constant HDR sky of the same size
    probe_data = scenes.ambient_probe(W, H, 2.5) if probe_kind == "constant" else scenes.sky_probe(W, H, seed=11)
    r = renderer.SampleRenderer(model, device=local_rank)
    r.resize((W, H))
    cam = scenes.ATRIUM_CAMERA if scene == "atrium" else scenes.STREET_CAMERA
    r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / float(H)))
    probe = renderer.ProbeData(probe_data).BuildCDF()
    r.setProbe(probe)
    cfg = abi.Config.reference_default()
    cfg.r_inner, cfg.r_outer = R_INNER, R_OUTER
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = SPP
    cfg.max_depth = 4
    cfg.rank, cfg.world = rank, world
    cfg.frames_in_flight = FRAMES_IN_FLIGHT[0]
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
    return r, cfg, model, probe_data, probe


def time_frames(r, frames, before_frame=None):
    for k in range(3):
        if before_frame:
            before_frame(k)
        r.render_async()
    r.synchronize()
    r.reset_stats()
    t0 = time.perf_counter()
    for k in range(frames):
        if before_frame:
            before_frame(3 + k)
        r.render_async()
    r.synchronize()
    dt = (time.perf_counter() - t0) / frames
    st = r.stats()
    rays = (st.radiance_rays + st.shadow_rays) / frames
    return {"ms_per_frame": round(dt * 1e3, 4), "rays_per_frame": rays, "mray_per_s": round(rays / dt / 1e6, 1),
            "paths_per_frame": int(st.paths // frames)}


def child_main(frames):
    """What the rocprofv3 counter passes run: the C3 frame a few times, nothing else."""
    import torch                                               # noqa: F401  (one HIP runtime for torch and libfovpt)
    from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
    r, cfg, _, _, _ = build_renderer(renderer, scenes, abi, TARGET_TRIS, 0, 0, 1)
    for _ in range(frames):
        r.launchParams.frame.subframe_index = 0
        r.render()
    r.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 --pmc child passes (traffic / valu then come from profiles/, labelled imported)")
    ap.add_argument("--no-variants", action="store_true", help="skip the HDR-probe / moving-camera / 1 M / 3.8 M-triangle variants")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=(0, 1, 2),
                    help="fovpt_config.frames_in_flight for the timed region: 0 = the library's default (2), 1 = one frame at a time")
    ap.add_argument("--gather", choices=("packed", "reduce", "lib"), default="packed",
                    help="N > 1: packed owned-pixel gather through torch.distributed (default), full-frame sum-reduce, or `lib`: the library's OWN "
                         "RCCL transport (fovpt_comm_init / fovpt_gather_frame, what a C++ host uses; also runs at N = 1 as a group of one)")
    ap.add_argument("--advance-subframe", action="store_true",
                    help="let render() advance subframe_index from frame to frame (new P-pass seeds every frame) instead of "
                         "resetting it to 0 as the shipped application does (main.cpp:402-407)")
    ap.add_argument("--child-frames", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    FRAMES_IN_FLIGHT[0] = args.frames_in_flight
    if args.child_frames:
        return child_main(args.child_frames)

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks fails with
    # hipIpcGetMemHandle: invalid argument (already exported on the GPU boxes; kept for any other launcher)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
        args.gpus = world

    # counters first, before this process touches the GPU (the passes are separate processes under the profiler)
    counters, counters_note = None, "not collected (N > 1 or --no-counters)"
    if world == 1 and not args.no_counters:
        counters, counters_note = run_counter_passes()

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
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
    from fovpathtracing_optixcodelatest_amd import multigpu

    r, cfg, model, probe_data, probe = build_renderer(renderer, scenes, abi, TARGET_TRIS, local_rank, rank, world)
    cam = scenes.ATRIUM_CAMERA

    # the frame the ranks render into and gather: torch tensors, handed to the library as the
    # caller-owned render target (render(CUDAOutputBuffer&), SimplePathtracer.cpp:216-226).  Two of
    # them at N > 1: the RCCL gather of frame k runs beside the rendering of frame k+1.
    nbuf = 2 if world > 1 else 1
    frames = [torch.zeros(H * W, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    pending = [None] * nbuf
    step_no = [0]
    packed = args.gather == "packed" and world > 1
    lib_gather = args.gather == "lib"
    if lib_gather:
        # the communicator of the library itself: rank 0 makes the unique id, the existing process group carries it to the others
        if rehearsal and world > 1:
            raise SystemExit("--gather lib needs one GPU per rank: RCCL refuses two ranks of one communicator on one device")
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(renderer.SampleRenderer.comm_unique_id()), dtype=torch.uint8, device="cuda")
        if world > 1:
            dist.broadcast(uid, src=0)
        r.comm_init(bytes(uid.cpu().tolist()), rank, world)
    pg = None
    if packed:
        pg = multigpu.PackedGather(r, dev, dst=0, nbuffers=nbuf)
        pg.plan()

    # torch's view of the stream on which the library's frames complete (fovpt_stream()): collectives issued
    # under it are ordered after the frame on the device, so the loop needs no host synchronisation
    lib_stream = torch.cuda.ExternalStream(r.stream, device=dev)

    def retire(k):
        """device-side completion of the gather that last used buffer set k (and, packed, the root's scatter)"""
        if pending[k] is None:
            return
        if pending[k] != "sync":
            pending[k].wait()                                  # the library's stream waits for the collective
        if packed:
            pg.finish(frames[k], k)
        pending[k] = None

    def step():
        k = step_no[0] % nbuf
        step_no[0] += 1
        with torch.cuda.stream(lib_stream):
            retire(k)
            # the shipped app resets subframe_index to 0 before every render() (main.cpp:402-407)
            if not args.advance_subframe:
                r.launchParams.frame.subframe_index = 0
            r.launchParams.frame.frame_buffer = frames[k].data_ptr()
            r.render_async()
            if lib_gather:
                # plan -> pack -> ncclSend / ncclRecv -> unpack, all queued by the library on its own completion stream: nothing to wait
                # for here; rank 0 gathers into the frame it rendered into
                r.gather_frame(0, frames[k].data_ptr(), frames[k].data_ptr() if rank == 0 else None)
            elif world > 1:
                if packed:
                    pending[k] = pg.gather(frames[k], k, async_op=True) or "sync"
                elif rehearsal:                                # gloo knows nothing about HIP streams
                    r.synchronize()
                    multigpu.gather_frame(frames[k], dst=0)
                else:                                          # RCCL waits for the frame, then reduces beside frame k+1
                    pending[k] = multigpu.gather_frame(frames[k], dst=0, async_op=True)

    def fence():
        r.synchronize()
        with torch.cuda.stream(lib_stream):
            for k in range(nbuf):
                retire(k)
        lib_stream.synchronize()
        r.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if lib_gather:
        gather_mode = "the library's own transport: fovpt_gather_frame = HIP pack -> ncclSend / ncclRecv (librccl, group of %d) -> HIP unpack on rank 0, on fovpt_stream()" % world
    elif world == 1:
        gather_mode = "none"
    elif packed:
        gather_mode = "packed owned pixels: HIP pack -> %s gather -> HIP unpack on rank 0, %d B per rank (full frame %d B)" % (
            "gloo (rehearsal, host-staged)" if rehearsal else "RCCL", pg.bytes_per_rank(), W * H * 4)
    else:
        gather_mode = "full-frame sum-reduce (%s)" % ("gloo rehearsal, host-synchronised" if rehearsal else "RCCL, overlapped with the next frame")
    # (a collective that fails raises on every rank: the run ends non-zero instead of printing a mixed-mode number)
    for _ in range(args.warmup):
        step()
    fence()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    last_frame_host = frames[(step_no[0] - 1) % nbuf].cpu()      # rank 0: the gathered frame
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
    # ... and the same two figures with fovpt_config.chains_per_frame = 2: every frame rendered as two independent chains over
    # halves of its sample slots, frames issued one at a time -- the throughput of two frames in flight at the latency of one
    two_chains = None
    if world == 1:
        cfg.chains_per_frame = 2
        r.config = cfg
        for _ in range(5):
            r.launchParams.frame.subframe_index = 0
            r.render_async()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(lat_frames):
            r.launchParams.frame.subframe_index = 0
            r.render_async()
            r.synchronize()
        lat2 = (time.perf_counter() - t0) / lat_frames * 1e3
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r.launchParams.frame.subframe_index = 0
            r.render_async()
        r.synchronize()
        two_chains = {"ms_per_step": round((time.perf_counter() - t0) / args.steps * 1e3, 4), "frame_latency_ms": round(lat2, 4),
                      "note": "fovpt_config.chains_per_frame = 2 (never `value`): frames back to back / one frame alone"}
        cfg.chains_per_frame = 0
        r.config = cfg

    # ---- per-kernel device time, HIP events on the library's own streams around every kernel (fovpt_stats): once as the
    # frame really runs (two streams, frames back to back: kernels overlap) and once with every kernel ALONE (profile 2)
    prof_frames = max(5, min(50, args.steps))
    per_frame_ms = {}
    ps = None
    timed_fif = cfg.frames_in_flight
    for mode, fif, label in ((2, timed_fif, "serialised"), (1, 1, "one_frame_in_flight"), (1, timed_fif, "overlapped")):
        cfg.profile = mode
        cfg.frames_in_flight = fif
        r.config = cfg
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
        per_frame_ms[label] = {
            "generate": round(ps.ms_generate / prof_frames, 5), "traverse_closest": round(ps.ms_trace / prof_frames, 5),
            "traverse_occlusion": round(ps.ms_shadow / prof_frames, 5), "shade": round(ps.ms_shade / prof_frames, 5),
            "resolve": round(ps.ms_resolve / prof_frames, 5)}
    cfg.profile = 0
    cfg.frames_in_flight = timed_fif
    r.config = cfg
    # k_traverse is the one traversal kernel (4 lanes per ray): closest-hit launches run on the main
    # stream (fovpt_stats books them under ms_trace), occlusion launches on the shadow stream
    # (ms_shadow); rocprofv3 reports both under the one kernel name.  `ps` holds the overlapped run.
    dom = "k_traverse"
    ms_k = ps.ms_trace + ps.ms_shadow
    n_launch = ps.n_trace_launches + ps.n_shadow_launches
    n_rays = ps.radiance_rays + ps.shadow_rays
    b_closest = algorithmic_bytes_per_ray(int(ps.num_triangles), "closest")
    b_any = algorithmic_bytes_per_ray(int(ps.num_triangles), "any")
    b_ray = (b_closest * ps.radiance_rays + b_any * ps.shadow_rays) / max(1, n_rays)
    avg_ms = ms_k / max(1, n_launch)
    bytes_per_launch = b_ray * (n_rays / max(1, n_launch))
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0             # as timed: two frames in flight
    ser = per_frame_ms["serialised"]
    one = per_frame_ms["one_frame_in_flight"]
    launches_per_frame = n_launch / prof_frames
    avg_ms_serialised = (ser["traverse_closest"] + ser["traverse_occlusion"]) / max(1e-9, launches_per_frame)
    avg_ms_one = (one["traverse_closest"] + one["traverse_occlusion"]) / max(1e-9, launches_per_frame)
    in_flight = ms_k / prof_frames / max(1e-9, ms_per_step)           # k_traverse launches running at once, on average
    achieved_one = bytes_per_launch / (avg_ms_one * 1e-3) / 1e9 if avg_ms_one > 0 else 0.0  # one frame in flight: the line's `achieved`

    # counters: measured in this run, else imported from the committed profile (and labelled so)
    traffic, traffic_source, valu = None, counters_note, None
    if counters and dom in counters:
        c = counters[dom]
        # units and gfx950 correction as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE
        # tallies 128-B requests at 64 B -> doubled; WRITE_SIZE is exact for wide stores
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            traffic = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
        tot_inst = sum(counters[k].get("SQ_INSTS_VALU", 0.0) * counters[k].get("launches_seen", 0) for k in counters)
        frames_seen = max(1.0, counters.get("k_resolve", {}).get("launches_seen", 3))
        if "SQ_INSTS_VALU" in c:
            lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None
            issue_ms = c["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (SIMDS * CLOCK_HZ) * 1e3
            valu = {
                "SQ_INSTS_VALU_per_launch": round(c["SQ_INSTS_VALU"]), "lane_use": round(lane, 4) if lane else None,
                "issue_time_ms_per_launch": round(issue_ms, 5),
                "issue_frac_of_serialised_launch": round(issue_ms / max(1e-9, c.get("dur_us", 0.0) / 1e3), 4) if c.get("dur_us") else None,
                "serialised_launch_ms_under_profiler": round(c.get("dur_us", 0.0) / 1e3, 5),
                "SQ_INSTS_VALU_per_frame_all_kernels": round(tot_inst / frames_seen),
                "issue_time_ms_per_frame_all_kernels": round(tot_inst / frames_seen * VALU_CYCLES_PER_INST / (SIMDS * CLOCK_HZ) * 1e3, 4),
                "issue_time_ms_per_frame_at_the_measured_rate": round(tot_inst / frames_seen / (SIMDS * VALU_MEASURED_INST_PER_NS_PER_SIMD) * 1e-6, 4),
                "issue_rate_note": "one wave64 instruction per SIMD per 2 cycles (spec) / 0.98 per ns (measured, tools/micro/valu_rate.hip)",
                "per_kernel_lane_use": {k: round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)
                                        for k, v in counters.items() if v.get("SQ_ACTIVE_INST_VALU")},
            }
    # The CU's address / data path (texture addresser, 64 bytes per clock): what the closest-hit launches' occupancy law has as
    # its constant term (profiles/r03c_step_probes.txt).  TA cycles per wave-level load instruction and the
    # load instructions per launch are measured in this run when the counter passes ran.
    l1_path = None
    dev_props = torch.cuda.get_device_properties(local_rank)
    cus = int(getattr(dev_props, "multi_processor_count", CUS) or CUS)                       # 256 on MI355X: from the device, not assumed
    clock_hz = float(getattr(dev_props, "clock_rate", 0) or 0) * 1e3 or CLOCK_HZ            # the device's engine clock (kHz -> Hz), else 2.4 GHz
    l1_frac = None
    if counters and dom in counters and counters[dom].get("TA_FLAT_READ_WAVEFRONTS_sum"):
        c = counters[dom]
        per_inst = c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"]
        ta_ms = c["TA_TA_BUSY_sum"] / cus / clock_hz * 1e3
        frames_seen = max(1.0, counters.get("k_resolve", {}).get("launches_seen", 3))
        tot_busy = sum(counters[k].get("TA_TA_BUSY_sum", 0.0) * counters[k].get("launches_seen", 0) for k in counters
                       if k in ("k_generate", "k_traverse", "k_shade", "k_resolve"))
        l1_path = {
            "load_instructions_per_launch": round(c.get("SQ_INSTS_VMEM_RD", c["TA_FLAT_READ_WAVEFRONTS_sum"])),
            "ta_cycles_per_load_instruction": round(per_inst, 2),
            "ta_time_ms_per_launch": round(ta_ms, 5),
            "ta_share_of_serialised_launch": round(ta_ms / max(1e-9, c.get("dur_us", 0.0) / 1e3), 4) if c.get("dur_us") else None,
            "ta_time_ms_per_frame_all_kernels": round(tot_busy / frames_seen / cus / clock_hz * 1e3, 4),
            "ta_share_of_the_frame_interval": round(tot_busy / frames_seen / cus / clock_hz * 1e3 / ms_per_step, 4),
            "cus": cus, "clock_ghz": round(clock_hz / 1e9, 3),
            "note": "TA_TA_BUSY / (the device's CU count) at the device's engine clock (a lower bound on the time: under this load the shader clock "
                    "runs ~8 % below it); a 128-byte "
                    "node is two 16-byte loads per lane = 2 x ~17 cycles of the CU's 64-byte-per-clock address path per wave step.  "
                    "`ta_share_of_the_frame_interval` is the roof this design runs against: the address paths' time for ALL of a frame's "
                    "loads over the frame interval of the timed region",
        }
        l1_frac = l1_path["ta_share_of_the_frame_interval"]
    if traffic is None:
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            with open(pmc_path) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
            traffic_source = "IMPORTED from profiles/pmc_traffic.json (not measured in this run: %s)" % counters_note
        except Exception:
            traffic_source = "unavailable: %s" % counters_note
    # Latency model of the dominant kernel: what a launch should take if a wave's life is its chain of
    # steps -- steps per wave x measured cycles per step (s_memtime stamps in a sample of the waves of a diagnostic build,
    # profiles/r03_step_cycles.txt) -- and the launch lasts (wave life) / (wave slots busy over the launch) (every wave's
    # start and end, profiles/r03_wave_timeline_c3.txt).  The step anatomy is IMPORTED from those committed profiles
    # (tools/step_model.py); the launch times beside it are measured in this run.
    latency_model = None
    try:
        with open(os.path.join(ROOT, "profiles", "r03_step_model.json")) as f:
            sm = json.load(f)["launches"]
        c1 = sm["closest_1"]
        model_closest = sum(sm[k]["model"]["launch_us"] for k in sm if k.startswith("closest") and "model" in sm[k]) / 1e3
        model_anyhit = sum(sm[k]["model"]["launch_us"] for k in sm if k.startswith("any-hit") and "model" in sm[k]) / 1e3
        latency_model = {
            "node_step_cycles": c1["node_step_cycles"], "leaf_step_cycles": c1["leaf_step_cycles"],
            "load_wait_share_of_node_step": c1["model"]["load_wait_share_of_node_step"],
            "node_steps_per_wave_per_launch": {k: sm[k]["node_steps_per_wave"] for k in sorted(sm) if "node_steps_per_wave" in sm[k]},
            "wave_slots_busy_over_a_launch": {k: sm[k]["unstamped"]["wave_slots_busy"] for k in sorted(sm) if "unstamped" in sm[k]},
            "model_ms_per_frame": {"traverse_closest": round(model_closest, 4), "traverse_occlusion": round(model_anyhit, 4)},
            "measured_ms_per_frame_serialised": {"traverse_closest": ser["traverse_closest"], "traverse_occlusion": ser["traverse_occlusion"]},
            "reading": "a node step of a wave (16 rays in lockstep) is ~1000 cycles, ~62 % of it the wait for the two 16-byte node loads of "
                       "its slowest quad, 16 % box test + rank, 8 % LDS push / pop, 14 % loop; a closest-hit launch is three one-wave "
                       "workgroups per wave slot, started by the dispatcher as slots fall free, and lasts (waves per slot) x (a wave's "
                       "life) / (share of the launch the slots are busy)",
            "source": "IMPORTED from profiles/r03_step_model.json (step anatomy, steps per wave, busy share: diagnostic builds of ROUND 3's "
                      "kernels -- the step anatomy still holds, the steps per wave are 12-40 % fewer since round 4's reinsertion and vote); "
                      "measured_ms_per_frame_serialised from this run",
        }
    except Exception as e:                                      # the line is still valid without the model
        latency_model = {"unavailable": "%s: %s" % (type(e).__name__, e)}
    lanes_env = os.environ.get("FOVPT_LANES")
    lanes_avail = int(lanes_env) if lanes_env and lanes_env.isdigit() and 1 <= int(lanes_env) <= 4 else 2
    fif_effective = min(timed_fif if timed_fif > 0 else lanes_avail, lanes_avail)
    roofline = {
        # flat keys first (the driver's parser keeps these): what binds the kernel, how close it is to THAT roof, and the two
        # figures that say why the contract's roofs do not bind
        "bound": "l1_address_path",
        "l1_frac": l1_frac,                                       # TA busy time of all of a frame's loads / frame interval (measured in this run, else null)
        "lane_use": (valu or {}).get("lane_use"),                 # k_traverse: active lanes per issued vector instruction
        "hbm_frac_measured": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if (traffic and avg_ms > 0) else None,
        "bound_note": "neither of the contract's roofs binds k_traverse: HBM traffic is a sixth of the algorithmic bytes (the scene stays in "
                      "L2 / Infinity Cache).  A node step sits at the knee of the CU's own address / data path (`l1_path`: ~17 texture-addresser "
                      "cycles per wave-level load, two per node step; a third costs +19 %, one fewer returns nothing) and is paced by its "
                      "dependent chain (load -> box test -> stack -> pop -> load) at the hardware's 8 waves per SIMD -- closest-hit time = "
                      "0.32 + 1.53 / (waves per SIMD) ms per frame -- with the vector ALUs about half busy (`valu`: half-empty waves, 16 "
                      "rays in lockstep, since round 4 with a vote on when a node phase ends).  DESIGN.md section 4 and EXPERIMENTS.md (round 3) have the "
                      "occupancy sweep and the three probes; `achieved`/`peak`/`frac` are the contract's algorithmic-bytes figure against the 8 TB/s HBM roof",
        # the contract's figure, for the configuration `value` is timed in: algorithmic bytes per launch / the launch's
        # HIP-event duration as the timed region runs (frames_in_flight as in `config`)
        "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
        "avg_launch_ms": round(avg_ms, 5), "avg_launch_ms_serialised": round(avg_ms_serialised, 5),
        "frac_serialised": round(bytes_per_launch / (avg_ms_serialised * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if avg_ms_serialised > 0 else None,
        "avg_launch_ms_one_frame_in_flight": round(avg_ms_one, 5),
        "frac_one_frame_in_flight": round(achieved_one / HBM_PEAK_GBS, 5),
        "launches_in_flight": round(in_flight, 3),
        "frac_aggregate": round(bytes_per_launch * launches_per_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "frac_note": "`achieved` / `frac` divide one launch's algorithmic bytes by its HIP-event duration AS THE TIMED REGION RUNS (with two "
                     "frames in flight the traversal launches of two frames share the chip -- `launches_in_flight` at any time -- so each "
                     "lasts longer although more gets done per second; how much longer depends on how the two chains interleave, and the "
                     "figure has come out as 0.36-0.38 or 0.46-0.47 at the same frame interval).  `frac_one_frame_in_flight` is the same with "
                     "one frame in flight (rounds 1-3's `frac`, stable), `frac_serialised` the kernel alone on the chip, `frac_aggregate` all "
                     "traversal bytes of a frame over the frame interval of the timed region",
        "launches_per_frame": launches_per_frame,
        "rays_per_launch": n_rays / max(1, n_launch), "algorithmic_bytes_per_ray": round(b_ray, 1),
        "per_frame_ms": per_frame_ms["overlapped"], "per_frame_ms_serialised": per_frame_ms["serialised"],
        "per_frame_ms_one_frame_in_flight": per_frame_ms["one_frame_in_flight"],
        "valu": valu,
        "l1_path": l1_path,
        "latency_model": latency_model,
    }

    out = {
        "metric": "Mray/s", "value": round(mrays, 2), "unit": "Mray/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "frame_latency_ms": round(frame_latency_ms, 4),
        # SURVEY 8(d)'s own definition of the metric -- wall time of render() INCLUDING its final synchronisation, one frame at a
        # time as the reference's main loop runs -- beside the pipelined `value`
        "value_sync_per_frame": round(rays_total / args.steps / (frame_latency_ms * 1e-3) / 1e6, 2) if world == 1 else None,
        "two_chains_per_frame": two_chains,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "C3: Sponza-class procedural atrium, %d tris, %dx%d, foveated spp fovea/mid/periphery %d/%d/%d, "
                        "radii %d/%d, gaze centre, Disney BSDF + probe NEE, depth 4, ambient probe 2.5 @ frame res"
                        % (model.num_triangles, W, H, SPP[2], SPP[1], SPP[0], R_INNER, R_OUTER),
            "triangles": model.num_triangles, "width": W, "height": H,
            "paths_per_frame": int(st.paths // max(1, args.steps)) if world == 1 else None,
            "rays_per_frame": rays_total / args.steps,
            "subframe": "advancing" if args.advance_subframe else "reset to 0 every frame (as the shipped app)",
            "parallelism": "tile-shard x%d + %s gather" % (world, args.gather) if world > 1 else "single GPU",
            "frames_in_flight": fif_effective,
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
        F.lp.frame.subframe_index = 0
        c = orc.render(S, F, ocfg, nthreads=cores)                 # untimed: the frame for the parity check and the ray counts
        lib_rays = c.lib_radiance + c.lib_shadow
        orc.set_lib_counting(False)                                # the timed frames do exactly the reference's work
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
        orc.set_lib_counting(True)
        out["cpu_baseline"] = {
            "value": round(rays_done / cpu_s / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "%d full C3 frame(s) (%.0f rays each as the REFERENCE traces them, incl. its discarded last segment and shadow "
                      "rays without effect; the library traces %d of them) in %.1f s, oracle/libfovpt_oracle.so, std::thread over %d threads"
                      % (frames_done, rays_done / frames_done, lib_rays, cpu_s, cores),
            "ms_per_frame": round(cpu_s / frames_done * 1e3, 2),
        }
        # comparable rates: both sides on the REFERENCE's ray count per frame
        out["config"]["rays_per_frame_reference"] = rays_done / frames_done
        out["config"]["value_on_reference_ray_count_mray_s"] = round(rays_done / frames_done / (ms_per_step * 1e-3) / 1e6, 1)
        out["config"]["rays_per_frame_equal_oracle_lib_count"] = bool(int(rays_total / args.steps) == lib_rays)
        # the timed frames are parity frames too: the oracle just rendered the same frame
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
    r.close()

    # ---- variants without the benchmark-shaped shortcuts (N = 1): extra keys, never `value`
    if world == 1 and not args.no_variants:
        variants = {}
        vf = max(10, min(40, args.steps))
        # (a) a seeded NON-constant HDR probe at frame resolution: every probe lookup goes to its own row (33 MB of texels,
        #     2 x 8.3 MB of CDF tables) instead of the one L1-resident row of the constant ambient probe
        rv, _, _, _, _ = build_renderer(renderer, scenes, abi, TARGET_TRIS, local_rank, 0, 1, probe_kind="hdr")

        def reset(_k, rr=rv):
            rr.launchParams.frame.subframe_index = 0
        variants["hdr_probe_1920x1080"] = time_frames(rv, vf, reset)

        # (b) the frame changes every time: subframe_index advances (new periphery seeds), camera and gaze move
        def move(k, rr=rv):
            eye = (cam["eye"][0] + 2.0 * k, cam["eye"][1] + 0.25 * math.sin(0.3 * k), cam["eye"][2] + 1.5 * math.cos(0.2 * k))
            rr.setCamera(renderer.Camera(eye, cam["lookat"], cam["up"], cam["fovy"], W / float(H)))
            rr.launchParams.frame.c.x = W // 2 + int(120 * math.sin(0.37 * k))
            rr.launchParams.frame.c.y = H // 2 + int(60 * math.cos(0.23 * k))
        variants["hdr_probe_moving_camera_and_gaze_advancing_subframe"] = time_frames(rv, vf, move)
        rv.close()
        # (c) north_star's "~1 M-triangle scene" and the Bistro-class 3.8 M one, same frame settings
        # and the open street of facade modules and foliage cards (depth complexity, long rays) under the HDR sky
        for name, tris, scn, pk in (("scene_1m_tris", 1000000, "atrium", "constant"), ("scene_3p8m_tris", 3800000, "atrium", "constant"),
                                    ("street_3p8m_tris_hdr_probe", 3800000, "street", "hdr")):
            rv, _, mv, _, _ = build_renderer(renderer, scenes, abi, tris, local_rank, 0, 1, probe_kind=pk, scene=scn)

            def reset2(_k, rr=rv):
                rr.launchParams.frame.subframe_index = 0
            v = time_frames(rv, vf, reset2)
            v["triangles"] = mv.num_triangles
            variants[name] = v
            rv.close()
        out["variants"] = variants

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
