"""Per-rank frame time of the tile-sharded C3 frame on ONE GPU (rank 0 of world N), without the gather."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
model = scenes.atrium(262144)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
for world in (1, 2, 4, 8):
    for rank in sorted(set((0, world - 1))):
        cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
        cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
        cfg.rank, cfg.world = rank, world
        r.config = cfg
        r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
        for _ in range(3):
            r.launchParams.frame.subframe_index = 0; r.render()
        r.reset_stats(); n = 30
        t = time.time()
        for _ in range(n):
            r.launchParams.frame.subframe_index = 0; r.render_async()
        r.synchronize(); dt = (time.time() - t) / n
        s = r.stats()
        print("world %d rank %d: %.3f ms/frame, paths %d, rays %d" % (world, rank, dt * 1e3, s.paths // n, (s.radiance_rays + s.shadow_rays) // n))
