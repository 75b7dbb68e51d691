"""tools/quick_perf.py with a non-constant HDR probe at frame resolution (per-kernel times, profile 1 and 2)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
size = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (W, H)
model = scenes.atrium(262144)
for kind in ("constant", "hdr"):
    r = renderer.SampleRenderer(model); r.resize((W, H))
    cam = scenes.ATRIUM_CAMERA
    r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
    data = scenes.ambient_probe(size[0], size[1], 2.5) if kind == "constant" else scenes.sky_probe(size[0], size[1], seed=11)
    r.setProbe(renderer.ProbeData(data).BuildCDF())
    cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    for profile in (0, 2):
        cfg.profile = profile; r.config = cfg
        r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
        for _ in range(3):
            r.launchParams.frame.subframe_index = 0; r.render()
        r.reset_stats(); t = time.time()
        for _ in range(frames):
            r.launchParams.frame.subframe_index = 0; r.render_async()
        r.synchronize(); dt = (time.time() - t) / frames; s = r.stats()
        if profile == 0:
            print(kind, "probe %dx%d: ms/frame %.3f rays %.0f" % (size[0], size[1], dt * 1e3, (s.radiance_rays + s.shadow_rays) / frames))
        else:
            print("   serialised ms: gen %.3f closest %.3f occlusion %.3f shade %.3f resolve %.3f" % tuple(
                x / frames for x in (s.ms_generate, s.ms_trace, s.ms_shadow, s.ms_shade, s.ms_resolve)))
    r.close()
