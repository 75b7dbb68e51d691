import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
model = scenes.atrium(262144)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
for world in (8,):
    cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    cfg.rank, cfg.world = 0, world
    cfg.profile = 1
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
    for _ in range(3):
        r.launchParams.frame.subframe_index = 0; r.render()
    r.reset_stats(); n = 30
    for _ in range(n):
        r.launchParams.frame.subframe_index = 0; r.render()      # synchronous: one frame at a time
    s = r.stats()
    print("world %d (sync frames): gen %.3f closest %.3f occl %.3f shade %.3f resolve %.3f ms/frame; launches closest %d occl %d" % (
        world, s.ms_generate / n, s.ms_trace / n, s.ms_shadow / n, s.ms_shade / n, s.ms_resolve / n, s.n_trace_launches // n, s.n_shadow_launches // n))
