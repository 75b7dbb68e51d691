"""Runs the BASELINE.json configurations C2..C5 on ONE GPU (functional + timing check; C4/C5 are
multi-GPU configs in BASELINE.json, here every rank's work is done by one device)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes

def run(name, tris, size, cfg, material="app", frames=40, eyes=1, gaze=None, scene="atrium"):      # (two frames are in flight: few frames would time the pipeline filling up)
    W, H = size
    t = time.time(); model = scenes.atrium(tris, material=material) if scene == "atrium" else scenes.street(tris, material=material)
    r = renderer.SampleRenderer(model)
    r.resize(size)
    cam = dict(scenes.ATRIUM_CAMERA if scene == "atrium" else scenes.STREET_CAMERA)
    r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5) if scene == "atrium" else scenes.sky_probe(512, 256, seed=5)).BuildCDF())
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = gaze or (W // 2, H // 2)
    setup = time.time() - t
    cams = []
    for e in range(eyes):                       # stereo: two cameras +-32 mm (scene units: cm-ish) apart
        eye = list(cam["eye"]); eye[2] += (e - (eyes - 1) / 2.0) * 6.4
        cams.append(renderer.Camera(eye, cam["lookat"], cam["up"], cam["fovy"], W / H))
    def frame():
        for c_ in cams:
            r.setCamera(c_)
            r.launchParams.frame.subframe_index = 0
            r.render_async()
    for _ in range(2): frame()
    r.synchronize(); r.reset_stats()
    t = time.time()
    for _ in range(frames): frame()
    r.synchronize()
    dt = (time.time() - t) / frames
    s = r.stats()
    rays = (s.radiance_rays + s.shadow_rays) / frames
    acc = r.downloadAccum()
    print("%s: %d tris, %dx%d x%d eye(s): %.3f ms/frame, %.2f Mrays/frame, %.0f Mray/s, paths %d, bvh depth %d nodes %d build %.1f ms, setup %.1f s, finite %s"
          % (name, model.num_triangles, W, H, eyes, dt * 1e3, rays / 1e6, rays / dt / 1e6, s.paths // frames, s.bvh_max_depth, s.num_bvh_nodes, s.ms_bvh_build, setup, np.isfinite(acc).all()))
    r.close()

def fov(ri, ro, depth=4):
    c = abi.Config.reference_default(); c.r_inner, c.r_outer = ri, ro
    c.spp_periphery, c.spp_middle, c.spp_fovea = 1, 2, 8; c.max_depth = depth
    return c
uni = abi.Config.reference_default(); uni.uniform, uni.spp_uniform = 1, 1
which = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
if "C2" in which: run("C2", 262144, (1920, 1080), uni, material="diffuse")
if "C3" in which: run("C3", 262144, (1920, 1080), fov(148, 482))
if "C4" in which: run("C4", 3800000, (2560, 1440), fov(197, 643))
if "C5" in which: run("C5", 3800000, (2160, 2160), fov(296, 964, depth=8), eyes=2, frames=10)
if "C4S" in which: run("C4-street", 3800000, (2560, 1440), fov(197, 643), scene="street")
if "C5S" in which: run("C5-street", 3800000, (2160, 2160), fov(296, 964, depth=8), eyes=2, frames=10, scene="street")
if "REF" in which:   # the reference's own shipped settings: 74/241, 8/16/32 spp
    c = abi.Config.reference_default(); run("REF-shipped", 262144, (1920, 1080), c)
if "U4" in which:    # FOV_OFF as shipped: uniform 4 spp
    c = abi.Config.reference_default(); c.uniform = 1; run("FOV_OFF-4spp", 262144, (1920, 1080), c, frames=20)
if "PUBF" in which:  # the reference's published foveated benchmark: 3840x2160, radii 74/241, spp 32/16/8 (133.7 ms on its RTX GPU, Sponza)
    c = abi.Config.reference_default(); c.spp_periphery, c.spp_middle, c.spp_fovea = 8, 16, 32
    run("published-fov-4K-32/16/8", 262144, (3840, 2160), c, frames=20)
if "PUBU" in which:  # the reference's published uniform benchmark: 3840x2160, 32 spp (3405 ms on its RTX GPU, Sponza)
    c = abi.Config.reference_default(); c.uniform, c.spp_uniform = 1, 32
    run("published-uniform-4K-32spp", 262144, (3840, 2160), c, frames=2)
